"""Parity over the REFERENCE'S OWN test fixtures (BASELINE.md section 3 names jumanpp_minimal.mdic as the parity
dictionary): the four dictionaries of /root/reference/test/jumandic (jumanpp_minimal, codegen, bug-28-lattice,
bug950111-003), the sentences the reference's tests analyse over them (jumandic_codegen_test.cc:52-63,
bug_28_lattice.cc, bug_950111-003_test.cc, train_mini_01.txt, unk_ex.data, partial_01.data), a random-weight model
per dictionary and a model trained by jumanpp_v2_train on train_mini_01.txt.

tests/golden/ref/ holds the models and the reference CLI's stdout per mode, written by tests/golden/make_ref_fixtures.sh
in the build container (the reference tree does not exist on the GPU box).  The emulator tests run the same kernel
sources on the CPU; the `gpu` tests run the shipped library / binary on the MI355X.  Everything is byte / bit exact."""
import os
import subprocess

import pytest

import golden_io as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, 'tests', 'golden', 'ref')
MODELS = ['minimal', 'minimal_trained', 'codegen', 'bug28', 'bug950111']
MODES = [
    ('juman', []),
    ('s5', ['-s', '5']),
    ('gbeam0', ['--global-beam=0']),
    ('b32', ['--beam=32', '--global-beam=32', '--right-beam=32', '-s', '32']),
    ('b3g10', ['--beam=3', '--global-beam=10', '--right-check=2', '--right-beam=4']),
    ('morph', ['-M']),
    ('segment', ['--segment']),
]


@pytest.fixture(scope='session')
def cli_emu(emu_lib):
    import __graft_entry__ as ge
    return ge.build_host_emu()


@pytest.fixture(scope='session')
def cli_gpu(gpu_lib):
    import __graft_entry__ as ge
    return ge.build_host()


def _cli(cli, model, flags, inputs):
    p = subprocess.run([cli, '--model=' + os.path.join(FIX, model + '.jppmdl')] + flags + inputs, capture_output=True)
    return p.returncode, p.stdout, p.stderr


def _check_modes(cli, model):
    bad = []
    for mode, flags in MODES:
        ref = open(os.path.join(FIX, '%s.%s.out' % (model, mode)), 'rb').read()
        assert len(ref) > 100, (model, mode)
        rc, out, err = _cli(cli, model, flags, [os.path.join(FIX, model + '.txt')])
        if rc != 0 or out != ref:
            bad.append((model, mode, rc, err[-200:]))
    assert not bad, bad


def _check_partial(cli, model):
    ref = open(os.path.join(FIX, model + '.partial.out'), 'rb').read()
    assert len(ref) > 100
    rc, out, err = _cli(cli, model, ['--partial-input'], [os.path.join(FIX, 'partial_01.data')])
    assert rc == 0 and out == ref, err[-300:]


def _check_lattice(J, lib, image, gold_name, **beams):
    ctx = J.Context(os.path.join(FIX, image), lib_path=lib, **beams)
    lines = [l.rstrip('\n') for l in open(os.path.join(FIX, 'minimal.txt'), encoding='utf-8')]
    meta, gold = G.read_gold(os.path.join(FIX, gold_name))
    assert meta['nsent'] == len(lines)
    res = ctx.analyze(lines).fetch(full=True)
    errs = []
    for s in range(len(lines)):
        errs += G.compare_sentence(res, s, gold[s], meta)
    assert not errs, errs[:10]


LATTICES = [('minimal.img', 'minimal.gold', {}), ('minimal_trained.img', 'minimal_trained.gold', {}),
            ('minimal.img', 'minimal_b32.gold', dict(beam=32, global_beam=32, right_check=1, right_beam=32))]


@pytest.mark.parametrize('model', MODELS)
def test_reference_fixture_dictionaries_cli_byte_identical(cli_emu, model):
    """Juman, -s 5 lattice, --global-beam=0, beam 32 lattice, beam 3 / global beam 10 / right-check 2, -M, --segment"""
    _check_modes(cli_emu, model)


@pytest.mark.parametrize('model', ['minimal', 'minimal_trained'])
def test_reference_partial_fixture(cli_emu, model):
    """test/jumandic/partial_01.data through --partial-input (the ScorePlugin path, partial_example.cc)"""
    _check_partial(cli_emu, model)


@pytest.mark.parametrize('image,gold_name,beams', LATTICES)
def test_reference_minimal_dictionary_full_lattice(emu_lib, image, gold_name, beams):
    """jumanpp_minimal.mdic: nodes, UNK records, entry rows, 14 patterns, T0 bits, global beams, beams, cells, top-1 path
    against `ref_dump dump` of the reference"""
    import jumanpp_amd as J
    _check_lattice(J, emu_lib, image, gold_name, **beams)


@pytest.mark.gpu
@pytest.mark.parametrize('model', MODELS)
def test_gpu_reference_fixture_dictionaries_cli_byte_identical(cli_gpu, model):
    _check_modes(cli_gpu, model)


@pytest.mark.gpu
@pytest.mark.parametrize('model', ['minimal', 'minimal_trained'])
def test_gpu_reference_partial_fixture(cli_gpu, model):
    _check_partial(cli_gpu, model)


@pytest.mark.gpu
@pytest.mark.parametrize('image,gold_name,beams', LATTICES)
def test_gpu_reference_minimal_dictionary_full_lattice(gpu_lib, image, gold_name, beams):
    import jumanpp_amd as J
    _check_lattice(J, gpu_lib, image, gold_name, **beams)
