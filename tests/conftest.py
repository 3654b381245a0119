import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')
    # The CLI writes its derived-image cache beside the model file: not for the models under tests/golden (a cache left
    # by one run would change what the next run exercises).  The cache tests switch it on for a copy of a model.
    os.environ.setdefault('JPPGPU_NO_IMAGE_CACHE', '1')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(scope='session')
def emu_lib():
    """TEST ONLY: kernel sources built against the CPU fiber emulator."""
    import __graft_entry__ as ge
    # (tools/emu_asan.sh: the emulator library built with -fsanitize=address)
    if os.environ.get('JPPEMU_TEST_LIB'):
        return os.environ['JPPEMU_TEST_LIB']
    return ge.build_emu()


@pytest.fixture(scope='session')
def gpu_lib():
    import __graft_entry__ as ge
    # developer builds of the same sources
    if os.environ.get('JPPGPU_TEST_LIB'):
        return os.environ['JPPGPU_TEST_LIB']
    return ge.build_native()


@pytest.fixture(scope='session')
def ref_tools():
    """oracle/_ref binaries (real reference); None if they were not built."""
    d = os.path.join(ROOT, 'oracle', '_ref')
    need = ['ref_dump', 'jpp_jumandic_bootstrap', 'jumanpp_v2']
    if all(os.path.exists(os.path.join(d, n)) for n in need):
        return d
    return None
