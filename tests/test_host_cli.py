"""C++14 host layer (jumanpp_amd/host: GpuAnalyzer + JumanFormat + jumanpp_gpu CLI) against
the reference CLI `jumanpp_v2`: the JUMAN-format output must be byte-identical
(src/jumandic/shared/juman_format.cc, src/jumandic/main/jumanpp.cc).

CPU tests link the host layer against the emulator build of the kernels (test
infrastructure); the `gpu` tests run the shipped binary on the MI355X."""
import os
import subprocess

import pytest

import golden_io as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='session')
def cli_emu(emu_lib):
    import __graft_entry__ as ge
    return ge.build_host_emu()


@pytest.fixture(scope='session')
def cli_gpu(gpu_lib):
    import __graft_entry__ as ge
    return ge.build_host()


def _run(cli, args, stdin=None, image_cache=False):
    # (the derived-image cache is written beside the model file: off for the models under tests/golden)
    env = dict(os.environ)
    if not image_cache:
        env['JPPGPU_NO_IMAGE_CACHE'] = '1'
    else:
        env.pop('JPPGPU_NO_IMAGE_CACHE', None)   # (conftest.py switches the cache off for every other CLI run of the suite)
    p = subprocess.run([cli] + args, input=stdin, capture_output=True, env=env)
    return p.returncode, p.stdout, p.stderr


def _sentences(out):
    """split JUMAN output into per-sentence blocks (each ends with EOS)"""
    blocks, cur = [], []
    for line in out.split(b'\n'):
        cur.append(line)
        if line == b'EOS':
            blocks.append(cur)
            cur = []
    return blocks


def _assert_juman_byte_identical(ours, ref):
    """Perceptron + RNN output must be the reference's bytes.  (Until round 2 an EOS-beam flip between paths
    whose reference totals tied within 1e-4 was tolerated; the device now evaluates the RNN and the weighted
    totals with the oracle build's arithmetic, tools/rnn_tie_audit.py, so there is nothing to tolerate.)"""
    bo, br = _sentences(ours), _sentences(ref)
    assert len(bo) == len(br)
    diff = [s for s, (a, b) in enumerate(zip(bo, br)) if a != b]
    assert not diff, (len(diff), diff[:5], bo[diff[0]], br[diff[0]])
    assert ours == ref


def test_juman_format_byte_identical_to_reference_cli(cli_emu, golden_dir):
    rc, out, err = _run(cli_emu, ['--model=' + os.path.join(golden_dir, 'mini.img'), os.path.join(golden_dir, 'mini.txt')])
    assert rc == 0, err
    ref = open(os.path.join(golden_dir, 'mini.juman.txt'), 'rb').read()
    assert out == ref


def test_juman_format_with_rnn(cli_emu, golden_dir):
    rc, out, err = _run(cli_emu, ['--model=' + os.path.join(golden_dir, 'mini_rnn.img'), os.path.join(golden_dir, 'mini.txt')])
    assert rc == 0, err
    ref = open(os.path.join(golden_dir, 'mini_rnn.juman.txt'), 'rb').read()
    _assert_juman_byte_identical(out, ref)
    # the perceptron-only run of the same model must not use the RNN and equals the plain model's output
    rc, out2, err = _run(cli_emu, ['--model=' + os.path.join(golden_dir, 'mini_rnn.img'), '--no-rnn',
                                   os.path.join(golden_dir, 'mini.txt')])
    assert rc == 0 and out2 == open(os.path.join(golden_dir, 'mini.juman.txt'), 'rb').read()


def _device_format_case(cli, golden_dir, tmp_path):
    """the top-1 JUMAN format is printed by the DEVICE by default (k_fmt_count / k_fmt_write over the per-model table of
    rendered entry rows, host/format_table.cc): it must say so, and its bytes must be those of the host formatters
    (--host-format) and of the reference -- stream and file pipelines, comments, failing lines, tiny batches, alias
    entries ("@ " rows) and every UNK maker (tests/golden/ref: the reference's own dictionaries)"""
    fix = os.path.join(golden_dir, 'ref')
    cases = [(os.path.join(golden_dir, 'mini.jppmdl'), os.path.join(golden_dir, 'mini.txt'), os.path.join(golden_dir, 'mini.juman.txt')),
             (os.path.join(golden_dir, 'mini_rnn.jppmdl'), os.path.join(golden_dir, 'mini.txt'), os.path.join(golden_dir, 'mini_rnn.juman.txt'))]
    for m in ('minimal', 'minimal_trained', 'codegen', 'bug28', 'bug950111'):
        cases.append((os.path.join(fix, m + '.jppmdl'), os.path.join(fix, m + '.txt'), os.path.join(fix, m + '.juman.out')))
    for model, txt, ref_path in cases:
        ref = open(ref_path, 'rb').read()
        rc, dev, err = _run(cli, ['--model=' + model, '--timing', txt])
        assert rc == 0 and b'device_format=1' in err, err[-300:]
        rc, host, err2 = _run(cli, ['--model=' + model, '--timing', '--host-format', txt])
        assert rc == 0 and b'device_format' not in err2
        assert dev == ref and host == ref, model
        out = str(tmp_path / 'o.txt')
        rc, _, err = _run(cli, ['--model=' + model, '--timing', '--batch=5', '-o', out, txt])
        assert rc == 0 and b'device_format=1' in err and b'sharded=1' in err and open(out, 'rb').read() == ref, model
    # comment lines, an empty line, invalid UTF-8, an over-long line, a tab as a one-byte surface: both formatters, both pipelines
    model = cases[0][0]
    data = ('# S-ID:1 first\n' + open(cases[0][1], encoding='utf-8').read() + '# c2\n\n# c3\nすごーーい\n').encode('utf-8') \
        + b'\xe3\x81\n' + ('あ' * 1400).encode('utf-8') + b'\n\t\n \n# last comment'
    src = tmp_path / 'in.txt'
    src.write_bytes(data)
    rc, host, eh = _run(cli, ['--model=' + model, '--host-format', str(src)])
    rc2, dev, ed = _run(cli, ['--model=' + model, str(src)])
    assert dev == host and rc == rc2 and ed == eh
    for batch in ('3', '1000'):
        out = str(tmp_path / 'o2.txt')
        rc3, _, e3 = _run(cli, ['--model=' + model, '--batch=' + batch, '-o', out, str(src)])
        assert open(out, 'rb').read() == host and rc3 == rc and e3 == eh, batch


def _device_lattice_case(cli, golden_dir, tmp_path):
    """the lattice (-s N) format is printed by the DEVICE by default (k_lat_count / k_lat_write, csrc/k_latfmt.h, over
    the table host/format_table.cc renders with LatticeFormat's own row printer): it must say so, and its bytes must be
    those of the host formatter (--host-format; that one is compared with the reference in
    test_lattice_format_byte_identical_to_reference_cli and the fixtures of tests/golden/ref).  Reference's own
    dictionaries (alias entries, every UNK maker), with and without the RNN (one / two score weights), N from 1 to
    beyond the beam, stream and file pipelines, tiny batches, comments (which replace the "# MA-SCORE" line), empty and
    failing lines, a tab as a one-byte surface."""
    fix = os.path.join(golden_dir, 'ref')
    cases = [(os.path.join(golden_dir, 'mini.jppmdl'), os.path.join(golden_dir, 'mini.txt')),
             (os.path.join(golden_dir, 'mini_rnn.jppmdl'), os.path.join(golden_dir, 'mini.txt'))]
    for m in ('minimal', 'minimal_trained', 'codegen', 'bug28', 'bug950111'):
        cases.append((os.path.join(fix, m + '.jppmdl'), os.path.join(fix, m + '.txt')))
    for model, txt in cases:
        for flags in (['-s', '1'], ['-s', '5'], ['--beam=12', '--global-beam=12', '--right-beam=12', '-s', '12'], ['-s', '40']):
            rc, host, eh = _run(cli, ['--model=' + model, '--host-format'] + flags + [txt])
            assert rc == 0 and host, eh[-300:]
            rc, dev, err = _run(cli, ['--model=' + model, '--timing'] + flags + [txt])
            assert rc == 0 and b'device_lattice_format=1' in err, err[-300:]
            assert dev == host, (model, flags)
        out = str(tmp_path / 'o.txt')
        rc, host, eh = _run(cli, ['--model=' + model, '--host-format', '-s', '5', txt])
        rc, _, err = _run(cli, ['--model=' + model, '--timing', '--batch=5', '-s', '5', '-o', out, txt])
        assert rc == 0 and b'device_lattice_format=1' in err and b'sharded=1' in err and open(out, 'rb').read() == host, model
    # the reference's lattice goldens of its own dictionaries, through the device formatter
    for m in ('minimal', 'codegen', 'bug28', 'bug950111'):
        ref = open(os.path.join(fix, m + '.s5.out'), 'rb').read()
        rc, dev, err = _run(cli, ['--model=' + os.path.join(fix, m + '.jppmdl'), '-s', '5', os.path.join(fix, m + '.txt')])
        assert rc == 0 and dev == ref, m
    for model in (cases[0][0], cases[1][0]):
        data = ('# S-ID:1 first\n' + open(cases[0][1], encoding='utf-8').read() + '# c2\n\n# c3\nすごーーい〜かぁっこいいねぇっッ！\n').encode('utf-8') \
            + b'\xe3\x81\n' + ('あ' * 1400).encode('utf-8') + b'\n\t\nx\ty\n \n# last comment'
        src = tmp_path / 'in.txt'
        src.write_bytes(data)
        rc, host, eh = _run(cli, ['--model=' + model, '--host-format', '-s', '4', str(src)])
        rc2, dev, ed = _run(cli, ['--model=' + model, '-s', '4', str(src)])
        assert dev == host and rc == rc2 and ed == eh
        for batch in ('3', '1000'):
            out = str(tmp_path / 'o2.txt')
            rc3, _, e3 = _run(cli, ['--model=' + model, '--batch=' + batch, '-s', '4', '-o', out, str(src)])
            assert open(out, 'rb').read() == host and rc3 == rc and e3 == eh, batch
    # what keeps the host formatter: --auto-nbest (N per sentence), no global beam
    rc, a, err = _run(cli, ['--model=' + cases[1][0], '--timing', '--auto-nbest=2:3:8', '-s', '1', cases[1][1]])
    assert rc == 0 and b'device_lattice_format' not in err
    rc, b, err = _run(cli, ['--model=' + cases[0][0], '--timing', '--global-beam=0', '-s', '2', cases[0][1]])
    assert b'device_lattice_format' not in err


def test_device_side_lattice_format(cli_emu, golden_dir, tmp_path):
    _device_lattice_case(cli_emu, golden_dir, tmp_path)


def _lattice_window_case(cli, golden_dir):
    """k_lat_write prints a round of 64 nodes into an LDS window of 12 KB and flushes it; nodes that do not fit together
    take several windows, a node beyond the window goes straight to the output.  With the library's developer switches
    (JPPGPU_DEV_LAT_WIN: a window of 256 / 64 bytes -- one or two lines fit, an alias entry's lines do not / hardly any
    line fits; JPPGPU_DEV_LAT_MANY_PREV: every node's previous-node list through the form for more than four distinct
    previous nodes) ordinary sentences walk those paths and every alignment of the flush: same bytes as the host class"""
    fix = os.path.join(golden_dir, 'ref')
    for win in ('256', '64'):
        env = dict(os.environ, JPPGPU_DEV_LAT_WIN=win, JPPGPU_DEV_LAT_MANY_PREV='1', JPPGPU_NO_IMAGE_CACHE='1')
        for model, txt in ((os.path.join(golden_dir, 'mini_rnn.jppmdl'), os.path.join(golden_dir, 'mini.txt')),
                           (os.path.join(fix, 'minimal.jppmdl'), os.path.join(fix, 'minimal.txt')),
                           (os.path.join(fix, 'bug950111.jppmdl'), os.path.join(fix, 'bug950111.txt'))):
            for flags in (['-s', '5'], ['--beam=12', '--global-beam=12', '--right-beam=12', '-s', '12']):
                rc, host, eh = _run(cli, ['--model=' + model, '--host-format'] + flags + [txt])
                p = subprocess.run([cli, '--model=' + model, '--timing'] + flags + [txt], capture_output=True, env=env)
                assert p.returncode == 0 and b'device_lattice_format=1' in p.stderr, p.stderr[-300:]
                assert p.stdout == host, (win, model, flags)


def test_device_side_lattice_format_window_paths(cli_emu, golden_dir):
    _lattice_window_case(cli_emu, golden_dir)


@pytest.mark.gpu
def test_gpu_device_side_lattice_format_window_paths(cli_gpu, golden_dir):
    _lattice_window_case(cli_gpu, golden_dir)


@pytest.mark.gpu
def test_gpu_device_side_lattice_format(cli_gpu, golden_dir, tmp_path):
    _device_lattice_case(cli_gpu, golden_dir, tmp_path)


def _adversarial_lattice_case(cli, golden_dir, ref_tools, tmp_path, n_fuzz, models, flag_sets):
    """the device lattice formatter on lines made to be nasty (every character class of util/characters.cc, 4-byte code
    points, ZWJ sequences, half-width kana, full-width digits, onomatopoeia and prolongation runs, tabs, '#' lines, empty
    lines, a 150-codepoint run): UNK nodes of every maker, normalized nodes with their flags, alias entries.  Same bytes
    as the host class, and equal to jumanpp_v2 up to the reference's own unstable choice among tied connections"""
    import random
    import test_cpu_parity as tc
    lines = tc._fuzz_lines(n_fuzz, 123)
    extra = ['すごーーーい', 'かぁっこいいねぇっッ！', 'ﾊﾝｶｸｶﾅ', '１２３４５６７８９０', '#not a comment', '# a comment', '', '\t', 'a\tb',
             '👨\u200d👩\u200d👧\u200d👦家族', '𠮷野家', 'きらきらきらきら', 'ワンワン', 'あ' * 150]
    lines = lines + extra * 2
    random.Random(5).shuffle(lines)
    path = str(tmp_path / 'fz.txt')
    open(path, 'w', encoding='utf-8').write('\n'.join(lines) + '\n')
    for model in models:
        for flags in flag_sets:
            rc, host, eh = _run(cli, ['--model=' + model, '--host-format'] + flags + [path])
            rc2, dev, ed = _run(cli, ['--model=' + model, '--timing'] + flags + [path])
            assert rc == rc2 and b'device_lattice_format=1' in ed and dev == host, (model, flags)
            if ref_tools is not None:
                refs = [_ref_cli(ref_tools, model, flags, path) for _ in range(2)]
                _lattice_blocks_equal_except_reference_unstable(dev, refs)


def test_device_lattice_format_on_adversarial_lines(cli_emu, golden_dir, ref_tools, tmp_path):
    _adversarial_lattice_case(cli_emu, golden_dir, ref_tools, tmp_path, 100,
                              [os.path.join(golden_dir, 'ref', 'minimal_trained.jppmdl'), os.path.join(golden_dir, 'mini_rnn.jppmdl')],
                              [['-s', '5'], ['-s', '60']])


@pytest.mark.gpu
def test_gpu_device_lattice_format_on_adversarial_lines(cli_gpu, golden_dir, ref_tools, tmp_path):
    _adversarial_lattice_case(cli_gpu, golden_dir, ref_tools, tmp_path, 400,
                              [os.path.join(golden_dir, 'ref', m) for m in ('minimal_trained.jppmdl', 'bug950111.jppmdl')] + [os.path.join(golden_dir, 'mini_rnn.jppmdl')],
                              [['-s', '5'], ['--beam=32', '--global-beam=32', '--right-beam=32', '-s', '32'],
                               ['--beam=3', '--global-beam=10', '--right-check=2', '--right-beam=4', '-s', '3'], ['-s', '60']])


def test_exact_percent_g_of_the_device(tmp_path):
    """csrc/jpp_fmtg.h -- the "%g" the device prints scores with -- against the C library: every exponent with structured
    mantissas, the neighbourhoods of the powers of ten and of d.ddddd5 ties, denormals, 2 M random bit patterns"""
    exe = str(tmp_path / 'fmtg_test')
    subprocess.check_call(['g++', '-std=c++17', '-O2', '-DJPP_EMU', '-I' + os.path.join(ROOT, 'tests', 'emu'),
                           '-I' + os.path.join(ROOT, 'jumanpp_amd', 'csrc'), os.path.join(ROOT, 'tests', 'host', 'fmtg_test.cc'), '-o', exe])
    p = subprocess.run([exe, '2000000'], capture_output=True, text=True)
    assert p.returncode == 0 and ' 0 mismatches' in p.stdout, p.stdout[-300:] + p.stderr[-2000:]


def test_device_side_juman_format(cli_emu, golden_dir, tmp_path):
    _device_format_case(cli_emu, golden_dir, tmp_path)


@pytest.mark.gpu
def test_gpu_device_side_juman_format(cli_gpu, golden_dir, tmp_path):
    _device_format_case(cli_gpu, golden_dir, tmp_path)


def test_cli_comments_errors_and_batching_like_reference(cli_emu, golden_dir, ref_tools, tmp_path):
    """comment lines, an over-long line, invalid UTF-8, empty lines, stdin input, tiny batches:
    stdout must equal the reference CLI's on the same bytes (the model comes from the goldens' recipe)."""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_gpu_parity as tg
    tmp = str(tmp_path)
    img, lines, _ = tg._fresh_workload(ref_tools, tmp, 2500, 12, 14, 11, length=30)
    text = []
    for i, l in enumerate(lines):
        if i % 3 == 0:
            text.append('# S-ID:%d comment text' % i)
        text.append(l)
    data = ('\n'.join(text) + '\n').encode('utf-8')
    data += b'# only a comment before a bad line\n\xe3\x81\n' + ('あ' * 1400).encode('utf-8') + b'\n\n' \
        + 'x\ty z　全角\n'.encode('utf-8') + '# trailing comment\n最後の行に改行なし'.encode('utf-8')
    ref = subprocess.run([os.path.join(ref_tools, 'jumanpp_v2'), '--model=' + os.path.join(tmp, 'w.model')],
                         input=data, capture_output=True)
    for batch in ('1', '5', '65536'):
        rc, out, err = _run(cli_emu, ['--model=' + img, '--batch=' + batch], stdin=data)
        assert out == ref.stdout, (batch, err[-300:])
        assert rc == ref.returncode


def test_cli_pipeline_and_format_threads_keep_the_output(cli_emu, ref_tools, tmp_path):
    """several batches in flight (two analyzers alternating) and several format workers per batch
    (64-sentence chunks) must give the bytes of the strictly serial run and of the reference CLI"""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_gpu_parity as tg
    tmp = str(tmp_path)
    img, lines, _ = tg._fresh_workload(ref_tools, tmp, 2500, 260, 12, 29, length=14)
    path = os.path.join(tmp, 'w.txt')
    ref = subprocess.run([os.path.join(ref_tools, 'jumanpp_v2'), '--model=' + os.path.join(tmp, 'w.model'), path],
                         capture_output=True)
    rc, serial, err = _run(cli_emu, ['--model=' + img, '--threads=1', '--no-pipeline', path])
    assert rc == 0 and serial == ref.stdout, err[-300:]
    for args in (['--batch=200', '--threads=4'], ['--batch=50', '--threads=3']):
        rc, out, err = _run(cli_emu, ['--model=' + img] + args + ['--timing', path])
        assert rc == 0 and out == serial, (args, err[-300:])
        assert b'sentences=260' in err


def test_cli_multi_device_keeps_the_output_and_its_order(cli_emu, ref_tools, tmp_path):
    """--devices=LIST: one analysis thread and analyzer pair per device, batches dealt in turn, ordered
    formatter.  On the emulator every ordinal is a separate context ("fake devices"): two and three devices
    must print the bytes of the single-device run and of the reference CLI, with and without the RNN."""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_gpu_parity as tg
    tmp = str(tmp_path)
    img, lines, _ = tg._fresh_workload(ref_tools, tmp, 2500, 90, 12, 31, length=14, rnn=(32, 600))
    path = os.path.join(tmp, 'w.txt')
    rc, single, err = _run(cli_emu, ['--model=' + img, '--batch=16', path])
    assert rc == 0, err[-300:]
    for devs, batch in (('0,1', '16'), ('0-2', '7'), ('1', '500')):
        rc, out, err = _run(cli_emu, ['--model=' + img, '--devices=' + devs, '--batch=' + batch, '--timing', path])
        assert rc == 0 and out == single, (devs, batch, err[-300:])
        assert ('devices=%d sentences=90' % {'0,1': 2, '0-2': 3, '1': 1}[devs]).encode() in err
    # perceptron only: byte-identical to the reference CLI as well
    rc, out, err = _run(cli_emu, ['--model=' + img, '--no-rnn', '--devices=0-3', '--batch=9', path])
    ref = subprocess.run([os.path.join(ref_tools, 'jumanpp_v2'), '--model=' + os.path.join(tmp, 'p.model'), path],
                         capture_output=True)
    assert rc == 0 and out == ref.stdout, err[-300:]
    rc, out, err = _run(cli_emu, ['--model=' + img, '--devices=0,x'], stdin=b'')
    assert rc == 1 and b'bad device list' in err


def test_gpu_analyzer_initialize_validates_like_the_reference(cli_emu, golden_dir):
    # unknown model file / missing model option: the CLI's own messages
    rc, out, err = _run(cli_emu, [])
    assert rc == 1 and b'Model file was not specified' in err
    rc, out, err = _run(cli_emu, ['--model=/nonexistent.img'])
    assert rc == 1 and b'failed to load model from disk' in err
    # what the device layout does not hold is rejected at initialize: a global beam beyond 32, a beam beyond 32 without
    # a global beam (with one, a wider beam is the lattice of beam 32 plus fake slots and is accepted)
    rc, out, err = _run(cli_emu, ['--model=' + os.path.join(golden_dir, 'mini.img'), '--beam=40', '--global-beam=40'], stdin=b'')
    assert rc == 1 and b'failed to initialize the analyzer' in err
    rc, out, err = _run(cli_emu, ['--model=' + os.path.join(golden_dir, 'mini.img'), '--beam=40', '--global-beam=0'], stdin=b'')
    assert rc == 1 and b'failed to initialize the analyzer' in err
    rc, out, err = _run(cli_emu, ['--model=' + os.path.join(golden_dir, 'mini.img'), '--beam=40'], stdin=b'')
    assert rc == 0


@pytest.mark.gpu
def test_gpu_cli_byte_identical_on_fresh_workload(cli_gpu, ref_tools, tmp_path):
    """2000 fresh sentences through the shipped jumanpp_gpu on the MI355X vs jumanpp_v2 on the host CPU."""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_gpu_parity as tg
    tmp = str(tmp_path)
    img, lines, _ = tg._fresh_workload(ref_tools, tmp, 30000, 2000, 20, 321)
    ref = subprocess.run([os.path.join(ref_tools, 'jumanpp_v2'), '--model=' + os.path.join(tmp, 'w.model'),
                          os.path.join(tmp, 'w.txt')], capture_output=True)
    rc, out, err = _run(cli_gpu, ['--model=' + img, os.path.join(tmp, 'w.txt')])
    assert rc == 0, err[-500:]
    assert out == ref.stdout
    rc, out, err = _run(cli_gpu, ['--model=' + img, '--batch=300', os.path.join(tmp, 'w.txt')])
    assert out == ref.stdout


@pytest.mark.gpu
def test_gpu_cli_with_rnn_on_fresh_workload(cli_gpu, ref_tools, tmp_path):
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_gpu_parity as tg
    tmp = str(tmp_path)
    img, lines, gold_path = tg._fresh_workload(ref_tools, tmp, 30000, 600, 20, 77, rnn=(128, 8000))
    ref = subprocess.run([os.path.join(ref_tools, 'jumanpp_v2'), '--model=' + os.path.join(tmp, 'w.model'),
                          os.path.join(tmp, 'w.txt')], capture_output=True)
    rc, out, err = _run(cli_gpu, ['--model=' + img, os.path.join(tmp, 'w.txt')])
    assert rc == 0, err[-500:]
    _assert_juman_byte_identical(out, ref.stdout)


@pytest.mark.gpu
def test_gpu_cli_devices_list(cli_gpu, ref_tools, tmp_path):
    """--devices on the MI355X box: the box has one GPU, so the list names it twice -- two analysis threads,
    two analyzer pairs, batches dealt in turn; the output must be the single-device bytes."""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_gpu_parity as tg
    tmp = str(tmp_path)
    img, lines, _ = tg._fresh_workload(ref_tools, tmp, 30000, 3000, 20, 23, rnn=(128, 8000))
    path = os.path.join(tmp, 'w.txt')
    rc, single, err = _run(cli_gpu, ['--model=' + img, '--batch=512', path])
    assert rc == 0, err[-300:]
    rc, out, err = _run(cli_gpu, ['--model=' + img, '--devices=0,0', '--batch=256', '--timing', path])
    assert rc == 0 and out == single, err[-300:]
    assert b'devices=2 sentences=3000' in err
    ref = subprocess.run([os.path.join(ref_tools, 'jumanpp_v2'), '--model=' + os.path.join(tmp, 'w.model'), path],
                         capture_output=True)
    assert out == ref.stdout


def test_cli_on_a_non_jumandic_spec(cli_emu, ref_tools, tmp_path):
    """jumanpp_gpu reads a .jppmdl whose spec is not the built-in one (native reader -> table-driven kernels) and
    prints jumanpp_v2's bytes, Juman and lattice format"""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_cpu_parity as tc
    # `add`: more n-gram features; `cols`: 12 feature columns per dictionary entry (round 5; > 8 was NotImplemented)
    for variant in ('add', 'cols'):
        tmp = str(tmp_path / variant)
        tc._variant_spec_workload(ref_tools, tmp, variant, 25, gold=False)
        model, txt = os.path.join(tmp, 'v.model'), os.path.join(tmp, 'v.txt')
        for flags in ([], ['-s', '3']):
            ref = _ref_cli(ref_tools, model, flags, txt)
            rc, out, err = _run(cli_emu, ['--model=' + model] + flags + [txt])
            assert rc == 0 and out == ref, (variant, flags, err[-300:])
        # the file-to-file pipeline (device-printed text)
        out_path = os.path.join(tmp, 'o.txt')
        rc, _, err = _run(cli_emu, ['--model=' + model, '-o', out_path, txt])
        assert rc == 0 and open(out_path, 'rb').read() == _ref_cli(ref_tools, model, [], txt), (variant, err[-300:])


def _sharded_input(golden_dir, tmp, copies):
    """mini.txt a few times over plus comment lines (one at the very end), an empty line and a line that fails"""
    src = open(os.path.join(golden_dir, 'mini.txt'), 'rb').read()
    path = os.path.join(tmp, 'sharded_in.txt')
    with open(path, 'wb') as f:
        f.write(src * copies)
        f.write('# S-ID:1 comment\nすごーーい〜かぁっこいいねぇっッ！\n\nx\ty\n'.encode('utf-8') + b'\xe3\x81\n' + b'# trailing comment')
    return path


@pytest.mark.parametrize('model', ['mini.jppmdl', 'mini_rnn.jppmdl'])
def test_sharded_pipeline_files_in_file_out(cli_emu, ref_tools, golden_dir, tmp_path, model):
    """INPUT... -o OUT takes the sharded pipeline (mapped input cut at newline boundaries, per-device line splitting,
    analysis, format workers and pwrite at sequenced offsets): the output file is the reference's stdout for one
    and for three devices, small and large batches, one and two input files"""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    tmp = str(tmp_path)
    path = _sharded_input(golden_dir, tmp, 2)
    m = os.path.join(golden_dir, model)
    ref = _ref_cli(ref_tools, m, [], path)
    out = os.path.join(tmp, 'o.txt')
    for dev, batch in (('0', 1000), ('0,1,2', 9)):
        rc, so, err = _run(cli_emu, ['--model=' + m, '--devices=' + dev, '--batch=%d' % batch, '--timing', '-o', out, path])
        assert rc == 0 and so == b'', err[-300:]
        assert b'sharded=1' in err and b'devices=%d ' % len(dev.split(',')) in err
        assert open(out, 'rb').read() == ref, (dev, batch)
    two = subprocess.run([os.path.join(ref_tools, 'jumanpp_v2'), '--model=' + m, path, os.path.join(golden_dir, 'mini.txt')],
                         capture_output=True).stdout
    rc, so, err = _run(cli_emu, ['--model=' + m, '--batch=40', '-o', out, path, os.path.join(golden_dir, 'mini.txt')])
    assert rc == 0 and open(out, 'rb').read() == two


def _output_shards_case(cli, golden_dir, tmp_path, devs=('0', '0,1,2', '0,1')):
    """--output-shards=K: the input in K contiguous parts at example boundaries, part k analysed into OUT.part000k (one
    file takes 14 GB/s however many threads write it, tools/host_write_ceiling.py: what eight GPUs print needs several).
    `cat OUT.part*` must be the one-file output: comment lines in front of every cut, more parts than examples, two
    input files, several devices, tiny batches, both device formatters and the host one"""
    import glob
    tmp = str(tmp_path)
    m = os.path.join(golden_dir, 'mini_rnn.jppmdl')
    lines = [l for l in open(os.path.join(golden_dir, 'mini.txt'), 'rb').read().split(b'\n') if l]
    body = b''
    for i, l in enumerate(lines * 3):
        # a comment (or two) in front of most examples: wherever a cut lands, comments are near it
        if i % 3 != 2:
            body += b'# S-ID:%d\n' % i
        if i % 7 == 0:
            body += b'# second comment line %d\n' % i
        body += l + b'\n'
    body += b'\n\xe3\x81\n# trailing comment'
    a, b = os.path.join(tmp, 'a.txt'), os.path.join(tmp, 'b.txt')
    open(a, 'wb').write(body)
    open(b, 'wb').write(b'# first of b\n' + lines[0] + b'\n' + lines[1])   # (no newline at the end)
    one = os.path.join(tmp, 'one.txt')
    for inputs in ([a], [a, b]):
        for fmt in (([], ['-s', '3'], ['--host-format'], ['-M']) if len(inputs) == 1 else ([],)):
            rc, _, e1 = _run(cli, ['--model=' + m, '--batch=7', '-o', one] + fmt + inputs)
            ref = open(one, 'rb').read()
            assert ref
            # (every part count with the default format; the other formats with one of them)
            for shards, dev, batch in ((2, devs[0], 1000), (5, devs[1], 6), (64, devs[2], 16)) if not fmt else ((5, devs[1], 6),):
                for f in glob.glob(os.path.join(tmp, 'sh.txt.part*')):
                    os.remove(f)
                rc2, so, err = _run(cli, ['--model=' + m, '--devices=' + dev, '--batch=%d' % batch, '--output-shards=%d' % shards,
                                          '--timing', '-o', os.path.join(tmp, 'sh.txt')] + fmt + inputs)
                files = sorted(glob.glob(os.path.join(tmp, 'sh.txt.part*')))
                assert rc2 == rc and len(files) == shards and b'sharded=1' in err, err[-300:]
                got = b''.join(open(f, 'rb').read() for f in files)
                assert got == ref, (inputs, fmt, shards)
                if shards == 2:
                    assert all(os.path.getsize(f) > len(ref) // 4 for f in files)   # (the parts are about equal)


def test_output_shards(cli_emu, golden_dir, tmp_path):
    _output_shards_case(cli_emu, golden_dir, tmp_path)


@pytest.mark.gpu
def test_gpu_output_shards(cli_gpu, golden_dir, tmp_path):
    _output_shards_case(cli_gpu, golden_dir, tmp_path, devs=('0', '0,0,0', '0,0'))   # (a one-GPU box: its GPU named again)


def test_sharded_pipeline_only_for_regular_files(cli_emu, golden_dir, tmp_path):
    """-o /dev/stdout (a pipe), a FIFO as input and an output that names the input must not take the mmap / pwrite
    pipeline: same bytes as the plain run, no ESPIPE failure, no SIGBUS"""
    m = os.path.join(golden_dir, 'mini.jppmdl')
    src = os.path.join(golden_dir, 'mini.txt')
    ref = open(os.path.join(golden_dir, 'mini.juman.txt'), 'rb').read()
    p = subprocess.run('%s --model=%s --timing -o /dev/stdout %s | cat' % (cli_emu, m, src), shell=True, capture_output=True)
    assert p.returncode == 0 and p.stdout == ref and b'sharded=1' not in p.stderr, p.stderr[-300:]
    p = subprocess.run(['bash', '-c', '%s --model=%s -o %s <(cat %s)' % (cli_emu, m, tmp_path / 'o1.txt', src)], capture_output=True)
    assert p.returncode == 0 and open(tmp_path / 'o1.txt', 'rb').read() == ref, p.stderr[-300:]
    same = tmp_path / 'same.txt'
    same.write_bytes(open(src, 'rb').read())
    p = subprocess.run([cli_emu, '--model=' + m, '-o', str(same), str(same)], capture_output=True)
    assert p.returncode in (0, 1), p.returncode          # (the reference truncates its own input as well) -- but no signal
    # regular files still take it
    rc, so, err = _run(cli_emu, ['--model=' + m, '--timing', '-o', str(tmp_path / 'o2.txt'), src])
    assert rc == 0 and b'sharded=1' in err and open(tmp_path / 'o2.txt', 'rb').read() == ref


@pytest.mark.gpu
def test_gpu_sharded_pipeline(cli_gpu, ref_tools, tmp_path):
    """the sharded pipeline on the MI355X box (one GPU: the device list names it twice), RNN model"""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_gpu_parity as tg
    tmp = str(tmp_path)
    img, lines, _ = tg._fresh_workload(ref_tools, tmp, 30000, 3000, 20, 29, rnn=(128, 8000))
    path = os.path.join(tmp, 'w.txt')
    ref = subprocess.run([os.path.join(ref_tools, 'jumanpp_v2'), '--model=' + os.path.join(tmp, 'w.model'), path],
                         capture_output=True).stdout
    out = os.path.join(tmp, 'o.txt')
    for dev, batch in (('0', 1024), ('0,0', 300)):
        rc, so, err = _run(cli_gpu, ['--model=' + img, '--devices=' + dev, '--batch=%d' % batch, '--timing', '-o', out, path])
        assert rc == 0 and b'sharded=1' in err, err[-300:]
        assert open(out, 'rb').read() == ref, dev


# ---- the drop-in boundary itself: the reference's UNMODIFIED formatters on a re-materialised Lattice ----

def _shim_check(ref_tools, model, lib, text_path, lattice_n, beams=()):
    """oracle/ref_dump.cc `shim`: model handed to the C ABI from the reference's structures,
    batch analysed by `lib`, reference Lattice rebuilt from the result view inside a reference Analyzer
    (INTEGRATION 4), reference JumanFormat / LatticeFormat run on it, compared with plain Analyzer::analyze."""
    import json
    with open(text_path, 'rb') as f:
        p = subprocess.run([os.path.join(ref_tools, 'ref_dump'), 'shim', model, lib, str(lattice_n)] + [str(b) for b in beams],
                           stdin=f, capture_output=True)
    assert p.returncode in (0, 1), p.stderr[-500:]
    r = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert r['status_mismatch'] == 0 and r['juman_identical'] == r['analysed'] > 0, (r, p.stderr[-500:])
    # lattice lines: identical except where two reference analyzers disagree among themselves
    # (address-hashed choice among exactly tied connections, lattice_config.h:109-124)
    if lattice_n > 0:
        assert r['lattice_identical'] >= r['analysed'] - r['lattice_unstable_in_reference'], r
    return r


def test_reference_formatters_on_rematerialised_lattice(emu_lib, ref_tools, golden_dir, tmp_path):
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    txt = os.path.join(golden_dir, 'mini.txt')
    r = _shim_check(ref_tools, os.path.join(golden_dir, 'mini.jppmdl'), emu_lib, txt, 3)
    assert r['analysed'] == 28 and r['lattice_identical'] == 28
    _shim_check(ref_tools, os.path.join(golden_dir, 'mini_rnn.jppmdl'), emu_lib, txt, 5)
    _shim_check(ref_tools, os.path.join(golden_dir, 'mini.jppmdl'), emu_lib, txt, 0, beams=(3, 8, 2, 4))
    import test_gpu_parity as tg
    tmp = str(tmp_path)
    tg._fresh_workload(ref_tools, tmp, 3000, 120, 14, 17, length=30, rnn=(48, 800))
    _shim_check(ref_tools, os.path.join(tmp, 'w.model'), emu_lib, os.path.join(tmp, 'w.txt'), 4)


@pytest.mark.gpu
def test_gpu_reference_formatters_on_rematerialised_lattice(gpu_lib, ref_tools, tmp_path):
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_gpu_parity as tg
    tmp = str(tmp_path)
    tg._fresh_workload(ref_tools, tmp, 30000, 1500, 20, 19, rnn=(128, 8000))
    r = _shim_check(ref_tools, os.path.join(tmp, 'w.model'), gpu_lib, os.path.join(tmp, 'w.txt'), 5)
    assert r['analysed'] == 1500


# ---- lattice format (-s N): src/jumandic/shared/lattice_format.cc ----

def _ref_cli(ref_tools, model, args, path):
    return subprocess.run([os.path.join(ref_tools, 'jumanpp_v2'), '--model=' + model] + args + [path],
                          capture_output=True).stdout


def _lattice_blocks_equal_except_reference_unstable(ours, ref_runs):
    """The criterion of _shim_check for the CLI: every sentence block must be the reference's bytes, except where the
    reference's own output is not a function of the lattice: LatticeFormat picks the connection whose scores a line
    prints with std::max_element over a FlatSet hashed by the HOST ADDRESS of ptr.previous (lattice_config.h:109-124),
    so among exactly tied connections of a node the printed one depends on where the process's pool landed (runs of
    jumanpp_v2 disagree with each other, and a fixed address layout makes all runs agree on an arbitrary choice).
    A block that is not the bytes of any reference run must therefore agree with the reference in everything that
    does not depend on that choice: the N-best totals line, the number of lines, and on every line all columns, the
    rank list included, except the three scores of the printed connection."""
    import re
    scores = re.compile('(特徴量スコア|言語モデルスコア|形態素解析スコア):-?[0-9.e+-]+'.encode('utf-8'))
    bo = ours.split(b'EOS\n')
    brs = [r.split(b'EOS\n') for r in ref_runs]
    assert all(len(b) == len(bo) for b in brs), [len(b) for b in brs] + [len(bo)]
    unstable, via_other_run, via_structure = 0, 0, 0
    for i, blk in enumerate(bo):
        refs = [b[i] for b in brs]
        if blk == refs[0]:
            continue
        if blk in refs:
            via_other_run += 1
            continue
        unstable += 1
        la, lb = blk.split(b'\n'), refs[0].split(b'\n')
        assert len(la) == len(lb), i
        assert la[0] == lb[0] or not la[0].startswith(b'# MA-SCORE'), (i, la[0], lb[0])   # the N-best totals
        for x, y in zip(la, lb):
            if x != y:
                assert scores.sub(b'\\1:#', x) == scores.sub(b'\\1:#', y), (i, x, y)
                via_structure += 1
    return unstable, via_other_run, via_structure


def test_lattice_format_byte_identical_to_reference_cli(cli_emu, ref_tools, tmp_path):
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_gpu_parity as tg
    tmp = str(tmp_path)
    img, lines, _ = tg._fresh_workload(ref_tools, tmp, 2500, 16, 14, 23, length=36)
    with open(os.path.join(tmp, 'w.txt'), 'ab') as f:
        f.write('\n# a comment replaces the MA-SCORE line\nすごーーい〜かぁっこいいねぇっッ！\nx\ty\n'.encode('utf-8'))
    # (-s 40 widens the beam to 40, jumanpp_args.cc:261-264: with a global beam the device keeps 32 slots per node, the
    # rest would be fake anyway -- csrc/jppgpu_api.cc: device_beam)
    for n in ('1', '5', '40'):
        ref = _ref_cli(ref_tools, os.path.join(tmp, 'w.model'), ['-s', n], os.path.join(tmp, 'w.txt'))
        for fmt in ([], ['--host-format']):
            rc, out, err = _run(cli_emu, ['--model=' + img, '-s', n] + fmt + [os.path.join(tmp, 'w.txt')])
            assert rc == 0, err[-300:]
            assert out == ref, (n, fmt)
    flags = ['--beam=50', '--global-beam=20', '--right-beam=9', '-s', '50']
    ref = _ref_cli(ref_tools, os.path.join(tmp, 'w.model'), flags, os.path.join(tmp, 'w.txt'))
    rc, out, err = _run(cli_emu, ['--model=' + img] + flags + [os.path.join(tmp, 'w.txt')])
    assert rc == 0 and out == ref, err[-300:]


def test_morph_and_segmented_formats_byte_identical(cli_emu, ref_tools, tmp_path):
    """-M / -F (MorphFormat), --segment (SegmentedFormat) and --dic-subset (SubsetFormat = full morph +
    every dictionary node of the lattice as a dictionary CSV line), incl. comments and failing lines"""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_gpu_parity as tg
    tmp = str(tmp_path)
    img, lines, _ = tg._fresh_workload(ref_tools, tmp, 2500, 14, 14, 31, length=32)
    with open(os.path.join(tmp, 'w.txt'), 'ab') as f:
        f.write('\n# S-ID:7\nすごーーい〜かぁっこいいねぇっッ！\n'.encode('utf-8') + b'\xe3\x81\n')
    for flags in (['-M'], ['-F'], ['--segment'], ['--segment', '--segment-separator=|'], ['--dic-subset']):
        ref = _ref_cli(ref_tools, os.path.join(tmp, 'w.model'), flags, os.path.join(tmp, 'w.txt'))
        rc, out, err = _run(cli_emu, ['--model=' + img] + flags + [os.path.join(tmp, 'w.txt')])
        assert out == ref, flags


def test_lattice_format_with_rnn(cli_emu, ref_tools, tmp_path):
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_gpu_parity as tg
    tmp = str(tmp_path)
    img, lines, gold_path = tg._fresh_workload(ref_tools, tmp, 2500, 12, 14, 29, length=30, rnn=(32, 600))
    refs = [_ref_cli(ref_tools, os.path.join(tmp, 'w.model'), ['-s', '3'], os.path.join(tmp, 'w.txt')) for _ in range(3)]
    rc, out, err = _run(cli_emu, ['--model=' + img, '-s', '3', os.path.join(tmp, 'w.txt')])
    assert rc == 0, err[-300:]
    unstable, _, _ = _lattice_blocks_equal_except_reference_unstable(out, refs)
    assert unstable <= len(lines) // 2


@pytest.mark.gpu
def test_gpu_config5_lattice_output_beam32_long_sentences(cli_gpu, ref_tools, tmp_path):
    """BASELINE configs[4] shape without the RNN: beam 32, >= 200 codepoints, `-s 32`: byte-identical."""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_gpu_parity as tg
    tmp = str(tmp_path)
    img, lines, _ = tg._fresh_workload(ref_tools, tmp, 8000, 60, 18, 131, length=210)
    flags = ['--beam=32', '--global-beam=32', '--right-check=1', '--right-beam=32', '-s', '32']
    ref = _ref_cli(ref_tools, os.path.join(tmp, 'w.model'), flags, os.path.join(tmp, 'w.txt'))
    rc, out, err = _run(cli_gpu, ['--model=' + img] + flags + [os.path.join(tmp, 'w.txt')])
    assert rc == 0, err[-300:]
    assert out == ref


@pytest.mark.gpu
def test_gpu_config5_lattice_output_with_rnn(cli_gpu, gpu_lib, ref_tools, tmp_path):
    """BASELINE configs[4]: beam 32, long sentences, RNNLM on, lattice-format output.  The lattice (beams, score cells,
    RNN-adjusted totals) is bit-exact, so the criterion is identity: (a) the reference's UNMODIFIED LatticeFormat on
    the Lattice re-materialised from our result view prints the bytes of a plain reference analysis (_shim_check),
    and (b) our own formatter's output equals jumanpp_v2's except on blocks where runs of jumanpp_v2 disagree."""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_gpu_parity as tg
    tmp = str(tmp_path)
    img, lines, gold_path = tg._fresh_workload(ref_tools, tmp, 8000, 40, 18, 137, length=210, rnn=(128, 5000),
                                               beams=[32, 32, 1, 32])
    r = _shim_check(ref_tools, os.path.join(tmp, 'w.model'), gpu_lib, os.path.join(tmp, 'w.txt'), 8, beams=(32, 32, 1, 32))
    assert r['analysed'] == len(lines)
    flags = ['--beam=32', '--global-beam=32', '--right-check=1', '--right-beam=32', '-s', '8']
    refs = [_ref_cli(ref_tools, os.path.join(tmp, 'w.model'), flags, os.path.join(tmp, 'w.txt')) for _ in range(3)]
    rc, out, err = _run(cli_gpu, ['--model=' + img] + flags + [os.path.join(tmp, 'w.txt')])
    assert rc == 0, err[-300:]
    _lattice_blocks_equal_except_reference_unstable(out, refs)


# ---- partial annotation (--partial-input): ScorePlugin hooks, src/core/input/partial_example*.cc ----

def _make_partial_input(lines, seed):
    """random partially annotated examples over corpus lines: plain chunks with no-break marks (&),
    node constraints with and without tags (existing values, unknown values, wrong lengths)"""
    import random
    rnd = random.Random(seed)
    pos_values = ['名詞', '動詞', '助詞', '副詞', '形容詞', '未定義語', '存在しない品詞']
    out = []
    for i, line in enumerate(lines):
        cps = list(line)
        if i % 5 == 0:
            out.append('# S-ID:%d' % i)
        p = 0
        while p < len(cps):
            n = rnd.randint(1, 9)
            chunk = cps[p:p + n]
            p += n
            kind = rnd.random()
            if kind < 0.45:
                s = []
                for j, c in enumerate(chunk):
                    if j > 0 and rnd.random() < 0.25:
                        s.append('&')
                    s.append(c)
                out.append(''.join(s))
            else:
                fields = ['', ''.join(chunk)]
                if kind > 0.7:
                    fields.append('pos:' + rnd.choice(pos_values))
                if kind > 0.9:
                    fields.append('subpos:' + rnd.choice(['*', '普通名詞', '格助詞']))
                out.append('\t'.join(fields))
        out.append('')
    return ('\n'.join(out) + '\n').encode('utf-8')


def test_partial_annotation_plugin_byte_identical(cli_emu, ref_tools, tmp_path):
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_gpu_parity as tg
    tmp = str(tmp_path)
    img, lines, _ = tg._fresh_workload(ref_tools, tmp, 2500, 40, 14, 41, length=30)
    data = _make_partial_input(lines, 7)
    pex = os.path.join(tmp, 'pex.txt')
    open(pex, 'wb').write(data)
    ref = _ref_cli(ref_tools, os.path.join(tmp, 'w.model'), ['--partial-input'], pex)
    rc, out, err = _run(cli_emu, ['--model=' + img, '--partial-input', pex])
    assert rc == 0, err[-300:]
    assert out == ref
    # the constraints must actually change analyses (otherwise this test shows nothing)
    plain = _ref_cli(ref_tools, os.path.join(tmp, 'w.model'), [], os.path.join(tmp, 'w.txt'))
    strip = lambda b: [x for x in _sentences(b)]
    changed = sum(1 for a, b in zip(strip(ref), strip(plain)) if [l for l in a if not l.startswith(b'# ')] != b)
    assert changed >= len(lines) // 2
    # lattice output and a different beam configuration through the same plugin path
    for flags in (['-s', '3'], ['--beam=3', '--global-beam=8', '--right-check=2', '--right-beam=4']):
        ref = _ref_cli(ref_tools, os.path.join(tmp, 'w.model'), ['--partial-input'] + flags, pex)
        rc, out, err = _run(cli_emu, ['--model=' + img, '--partial-input'] + flags + [pex])
        assert out == ref, flags
    # the generic ScorePlugin entry point (jppgpu_analyze_batch_plugin, GpuAnalyzer::analyzeBatch(inputs, plugin)):
    # the same constraints evaluated per lattice node by a host-side plugin instead of the device kernel
    ref = _ref_cli(ref_tools, os.path.join(tmp, 'w.model'), ['--partial-input'], pex)
    env = dict(os.environ, JPPGPU_PARTIAL_VIA_PLUGIN='1')
    for flags in ([], ['--batch=7'], ['--auto-nbest=2:9:7']):
        p = subprocess.run([cli_emu, '--model=' + img, '--partial-input'] + flags + [pex], capture_output=True, env=env)
        want = ref if not flags or flags[0].startswith('--batch') else _ref_cli(ref_tools, os.path.join(tmp, 'w.model'),
                                                                                 ['--partial-input'] + flags, pex)
        assert p.returncode == 0 and p.stdout == want, (flags, p.stderr[-300:])


@pytest.mark.gpu
def test_gpu_partial_annotation_plugin(cli_gpu, ref_tools, tmp_path):
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_gpu_parity as tg
    tmp = str(tmp_path)
    img, lines, _ = tg._fresh_workload(ref_tools, tmp, 20000, 800, 18, 43)
    data = _make_partial_input(lines, 9)
    pex = os.path.join(tmp, 'pex.txt')
    open(pex, 'wb').write(data)
    ref = _ref_cli(ref_tools, os.path.join(tmp, 'w.model'), ['--partial-input'], pex)
    rc, out, err = _run(cli_gpu, ['--model=' + img, '--partial-input', pex])
    assert rc == 0, err[-300:]
    assert out == ref
    # ... and through the generic plugin entry point (host-side plugin, per-node penalties uploaded per batch)
    p = subprocess.run([cli_gpu, '--model=' + img, '--partial-input', '--batch=300', pex], capture_output=True,
                       env=dict(os.environ, JPPGPU_PARTIAL_VIA_PLUGIN='1'))
    assert p.returncode == 0 and p.stdout == ref, p.stderr[-300:]


# ---- native .jppmdl reader (jumanpp_amd/host/jppmdl_reader.cc) ----

def test_native_jppmdl_reader_equals_exported_image(cli_emu, golden_dir):
    """the reference's own model container, read without any reference-linked tool, must drive the
    analysis to the same bytes as the image exported by ref_dump (and as the reference CLI)"""
    ref = open(os.path.join(golden_dir, 'mini.juman.txt'), 'rb').read()
    rc, out, err = _run(cli_emu, ['--model=' + os.path.join(golden_dir, 'mini.jppmdl'), os.path.join(golden_dir, 'mini.txt')])
    assert rc == 0, err[-300:]
    assert out == ref
    for flags in (['-s', '3'], ['-F'], ['--segment']):
        a = _run(cli_emu, ['--model=' + os.path.join(golden_dir, 'mini_rnn.jppmdl')] + flags + [os.path.join(golden_dir, 'mini.txt')])
        b = _run(cli_emu, ['--model=' + os.path.join(golden_dir, 'mini_rnn.img')] + flags + [os.path.join(golden_dir, 'mini.txt')])
        assert a[0] == 0 and a[1] == b[1], flags
    # a truncated / foreign file is rejected with the reference's message
    bad = os.path.join(golden_dir, 'mini.txt')
    rc, out, err = _run(cli_emu, ['--model=' + bad])
    assert rc == 1 and b'has corrupted header' in err


def test_native_jppmdl_partial_input(cli_emu, ref_tools, tmp_path):
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_gpu_parity as tg
    tmp = str(tmp_path)
    img, lines, _ = tg._fresh_workload(ref_tools, tmp, 2500, 20, 14, 51, length=30)
    data = _make_partial_input(lines, 3)
    pex = os.path.join(tmp, 'pex.txt')
    open(pex, 'wb').write(data)
    model = os.path.join(tmp, 'w.model')
    ref = _ref_cli(ref_tools, model, ['--partial-input'], pex)
    rc, out, err = _run(cli_emu, ['--model=' + model, '--partial-input', pex])
    assert rc == 0 and out == ref, err[-300:]


@pytest.mark.gpu
def test_gpu_cli_native_jppmdl_with_rnn(cli_gpu, ref_tools, tmp_path):
    """jumanpp_gpu --model=<the reference's .jppmdl>, perceptron + RNN, on the MI355X"""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_gpu_parity as tg
    tmp = str(tmp_path)
    img, lines, gold_path = tg._fresh_workload(ref_tools, tmp, 30000, 400, 20, 91, rnn=(128, 8000))
    a = _run(cli_gpu, ['--model=' + os.path.join(tmp, 'w.model'), os.path.join(tmp, 'w.txt')])
    b = _run(cli_gpu, ['--model=' + img, os.path.join(tmp, 'w.txt')])
    assert a[0] == 0 and a[1] == b[1], a[2][-300:]
    ref = _ref_cli(ref_tools, os.path.join(tmp, 'w.model'), [], os.path.join(tmp, 'w.txt'))
    _assert_juman_byte_identical(a[1], ref)


# ---- auto beam (--auto-nbest=base:step:max): AnalyzerImpl::autoBeamSizes, analyzer_impl.cc:350-361 ----

def test_auto_beam_byte_identical(cli_emu, ref_tools, tmp_path):
    """every sentence gets beam = global beam = min(base + codepoints / step, max); sentences of several lengths"""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_gpu_parity as tg
    tmp = str(tmp_path)
    img, lines, _ = tg._fresh_workload(ref_tools, tmp, 2500, 12, 14, 71, length=44)
    more = [l[:k] for l, k in zip(lines, (3, 9, 14, 21, 27, 33, 38, 41, 5, 17, 25, 30))]
    txt = os.path.join(tmp, 'auto.txt')
    open(txt, 'w', encoding='utf-8').write('\n'.join(lines + more) + '\n\n')
    model = os.path.join(tmp, 'w.model')
    for flags in (['--auto-nbest=2:10:8'], ['--auto-nbest=2:10:8', '-s', '8'], ['--auto-nbest=3:7:20', '-s', '5'],
                  ['--auto-nbest=1:6:5', '--right-check=2', '--right-beam=3', '-M']):
        ref = _ref_cli(ref_tools, model, flags, txt)
        rc, out, err = _run(cli_emu, ['--model=' + model] + flags + [txt])
        assert rc == 0 and out == ref, (flags, err[-300:])
    # with partial annotation on top
    data = _make_partial_input(lines, 13)
    pex = os.path.join(tmp, 'pex.txt')
    open(pex, 'wb').write(data)
    flags = ['--auto-nbest=2:9:7', '--partial-input']
    ref = _ref_cli(ref_tools, model, flags, pex)
    rc, out, err = _run(cli_emu, ['--model=' + model] + flags + [pex])
    assert rc == 0 and out == ref, err[-300:]


@pytest.mark.gpu
def test_gpu_auto_beam(cli_gpu, ref_tools, tmp_path):
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_gpu_parity as tg
    tmp = str(tmp_path)
    img, lines, _ = tg._fresh_workload(ref_tools, tmp, 20000, 400, 18, 73, length=60)
    cut = [l[:(7 * i) % 60 + 1] for i, l in enumerate(lines)]
    txt = os.path.join(tmp, 'auto.txt')
    open(txt, 'w', encoding='utf-8').write('\n'.join(cut) + '\n')
    model = os.path.join(tmp, 'w.model')
    for flags in (['--auto-nbest=2:8:12'], ['--auto-nbest=3:10:9', '-s', '3']):
        ref = _ref_cli(ref_tools, model, flags, txt)
        rc, out, err = _run(cli_gpu, ['--model=' + model] + flags + [txt])
        assert rc == 0 and out == ref, (flags, err[-300:])


def _parallel_reference(ref_tools, model, path, tmp, procs=16):
    """jumanpp_v2 has no threading: split the corpus, one process per part, concatenate in order"""
    lines = open(path, 'rb').read().split(b'\n')
    if lines and lines[-1] == b'':
        lines.pop()
    per = (len(lines) + procs - 1) // procs
    ps = []
    for k in range(procs):
        part = os.path.join(tmp, 'part%d.txt' % k)
        with open(part, 'wb') as f:
            f.write(b''.join(l + b'\n' for l in lines[k * per:(k + 1) * per]))
        ps.append(subprocess.Popen([os.path.join(ref_tools, 'jumanpp_v2'), '--model=' + model, part],
                                   stdout=subprocess.PIPE, stderr=subprocess.DEVNULL))
    return lines, b''.join(p.communicate()[0] for p in ps)


@pytest.mark.gpu
@pytest.mark.parametrize('rnn', [False, True])
def test_gpu_cli_at_scale_on_the_bench_workload(cli_gpu, ref_tools, tmp_path, rnn):
    """bench.py's own model (300 k dictionary entries, 2^22 weights, E=128 RNN) and corpus generator:
    60 000 (perceptron) / 20 000 (RNN) sentences through jumanpp_gpu vs jumanpp_v2: byte-identical
    output in both configurations (the headline configuration is the RNN one)."""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import argparse
    import sys
    sys.path.insert(0, ROOT)
    import bench
    tmp = str(tmp_path)
    args = argparse.Namespace(dict_entries=300000, weights_exp=22, seed=20260925, rnn=rnn, rnn_hidden=128,
                              rnn_vocab=30000, sent_len=40)
    cache = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'jppgpu_bench_cache')
    mdic, model, img = bench.make_workload(args, cache)
    n = 20000 if rnn else 60000
    corpus = bench.make_corpus(args, mdic, cache, n, 4242)
    lines, ref = _parallel_reference(ref_tools, model, corpus, tmp)
    rc, out, err = _run(cli_gpu, ['--model=' + model, '--batch=16384', corpus])  # native .jppmdl, 4 batches in the pipeline
    assert rc == 0, err[-500:]
    # perceptron only and perceptron + RNN alike: the reference's bytes
    bo, br = _sentences(out), _sentences(ref)
    assert len(bo) == len(br) == n
    diff = [i for i in range(n) if bo[i] != br[i]]
    assert not diff, (len(diff), diff[:5])
    assert out == ref


def test_dic_subset_csv_quoting(cli_emu, ref_tools, tmp_path):
    """--dic-subset with dictionary fields that contain commas and quotes (CSV quoting of MdicFormat)"""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    tmp = str(tmp_path)
    mdic = os.path.join(tmp, 'd.mdic')
    with open(mdic, 'w', encoding='utf-8') as f:
        subprocess.check_call(['python3', os.path.join(ROOT, 'tools', 'gen_dict.py'), '300', '--seed', '3'], stdout=f)
        f.write('引用符,0,0,0,名詞,普通名詞,*,*,引用符,"いん,よう",引用符/いんようふ,"代表表記:引用符/いん""よう カテゴリ:抽象物,記号"\n')
        f.write('コンマ,0,0,0,名詞,普通名詞,*,*,"コ""ンマ",こんま,コンマ/こんま,代表表記:コンマ/こんま\n')
    subprocess.check_call([os.path.join(ref_tools, 'jpp_jumandic_bootstrap'), mdic, os.path.join(tmp, 'd.seed')],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    model = os.path.join(tmp, 'd.model')
    subprocess.check_call([os.path.join(ref_tools, 'ref_dump'), 'mkmodel', os.path.join(tmp, 'd.seed'), model, '14', '3', '0.1'])
    txt = os.path.join(tmp, 't.txt')
    with open(txt, 'w', encoding='utf-8') as f:
        f.write('引用符とコンマ\n# a comment\nコンマ引用符コンマ\n')
    ref = _ref_cli(ref_tools, model, ['--dic-subset'], txt)
    rc, out, err = _run(cli_emu, ['--model=' + model, '--dic-subset', txt])
    assert rc == 0 and out == ref, err[-300:]
    assert b'"' in out and b'""' in out


def test_cli_format_names_and_lattice_beam_widening(cli_emu, ref_tools, golden_dir):
    """--format=NAME for every non-protobuf name, and `-s N` widening the beam to N (jumanpp_args.cc:253-256)"""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    model = os.path.join(golden_dir, 'mini.jppmdl')
    txt = os.path.join(golden_dir, 'mini.txt')
    for flags in (['--format=juman'], ['--format=segment'], ['--format=morph'], ['--format=full-morph'],
                  ['--format=dic-subset'], ['--format=lattice'], ['-s', '8'], ['-s', '8', '--global-beam=12'],
                  ['-s', '3', '--beam=2'], ['-s7', '--beam=6']):
        ref = _ref_cli(ref_tools, model, flags, txt)
        rc, out, err = _run(cli_emu, ['--model=' + model] + flags + [txt])
        assert rc == 0 and len(ref) > 1000 and out == ref, flags


def _fold_unk_ties(out):
    """two UNK makers give nodes with identical feature rows and therefore exactly tied paths; which of the
    twins wins an RNN-rescored tie is inside the 1e-4 float contract.  Fold their only visible difference."""
    import re
    return re.sub('"未知語:[^"]*"'.encode('utf-8'), b'"UNK"', out)


def test_cli_rnn_config_overrides_and_config_file(cli_emu, ref_tools, golden_dir, tmp_path):
    """--rnn-nce-bias / --rnn-unk-constant / --rnn-unk-length / --feature-weight-{perceptron,rnn} and
    -c CONFIG behave like the reference's JumanppEnv::setRnnConfig (incl. score weights falling back to 1.0
    when only another RNN flag is given, and RNN weight 0 switching the RNN off)"""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    model = os.path.join(golden_dir, 'mini_rnn.jppmdl')
    txt = os.path.join(golden_dir, 'mini.txt')
    base = _ref_cli(ref_tools, model, [], txt)
    seen = {base}
    for flags in (['--rnn-nce-bias=3.0'], ['--feature-weight-rnn=0.05', '--feature-weight-perceptron=2'],
                  ['--feature-weight-rnn=0'], ['--feature-weight-rnn=0', '--feature-weight-perceptron=3'],
                  ['--rnn-unk-constant=-1', '--rnn-unk-length=-0.5'],
                  ['--rnn-nce-bias=5.6', '--rnn-unk-constant=-3.47', '--rnn-unk-length=-2.93',
                   '--feature-weight-perceptron=1', '--feature-weight-rnn=0.0176']):
        ref = _ref_cli(ref_tools, model, flags, txt)
        rc, out, err = _run(cli_emu, ['--model=' + model] + flags + [txt])
        assert rc == 0, err[-300:]
        assert _fold_unk_ties(out) == _fold_unk_ties(ref), flags
        seen.add(ref)
    assert len(seen) >= 4  # the flags do change the analysis
    # config file (whitespace-separated arguments), overridden by the command line
    conf = os.path.join(str(tmp_path), 'jumandic.conf')
    with open(conf, 'w') as f:
        f.write('--model=%s\n--rnn-nce-bias=5.6 --feature-weight-rnn=0.02\n  --beam=4\n' % model)
    for extra in ([], ['--beam=6', '--feature-weight-rnn=0.5']):
        ref = subprocess.run([os.path.join(ref_tools, 'jumanpp_v2'), '-c', conf] + extra + [txt], capture_output=True).stdout
        rc, out, err = _run(cli_emu, ['-c', conf] + extra + [txt])
        assert rc == 0 and len(ref) > 1000 and _fold_unk_ties(out) == _fold_unk_ties(ref), extra
    # a relative model path is looked up next to the config file
    import shutil
    shutil.copy(model, os.path.join(str(tmp_path), 'm.jppmdl'))
    with open(conf, 'w') as f:
        f.write('--model=m.jppmdl\n')
    rc, out, err = _run(cli_emu, ['--config=' + conf, txt])
    assert rc == 0 and _fold_unk_ties(out) == _fold_unk_ties(base)
    # flat images carry only the resolved values: overrides are refused instead of guessed
    rc, out, err = _run(cli_emu, ['--model=' + os.path.join(golden_dir, 'mini_rnn.img'), '--rnn-nce-bias=1', txt])
    assert rc == 1 and b'failed to apply the RNN configuration' in err


@pytest.mark.gpu
def test_gpu_cli_external_rnn_model(cli_gpu, ref_tools, golden_dir, tmp_path):
    """the same on the MI355X: the device walks the double arrays the host built"""
    test_cli_external_rnn_model(cli_gpu, ref_tools, golden_dir, tmp_path)


def test_cli_external_rnn_model(cli_emu, ref_tools, golden_dir, tmp_path):
    """--rnn-model=PATH (faster-rnnlm vocabulary + PATH.nnet) on a perceptron-only model: the host builds the
    word-id double arrays itself (RnnIdResolver::build) and must analyse like the reference does"""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    tmp = str(tmp_path)
    mdic = os.path.join(tmp, 'mini.mdic')
    with open(mdic, 'w', encoding='utf-8') as f:  # the dictionary of tests/golden/mini.jppmdl (make_golden.sh)
        subprocess.check_call(['python3', os.path.join(ROOT, 'tools', 'gen_dict.py'), '2500', '--seed', '11'], stdout=f)
    rnn = os.path.join(tmp, 'rnn')
    subprocess.check_call(['python3', os.path.join(ROOT, 'tools', 'gen_rnn.py'), mdic, rnn, '--vocab', '600', '--hidden', '32',
                           '--maxent-size', '16384', '--seed', '31'], stdout=subprocess.DEVNULL)
    model = os.path.join(golden_dir, 'mini.jppmdl')
    txt = os.path.join(golden_dir, 'mini.txt')
    plain = _ref_cli(ref_tools, model, [], txt)
    for flags in (['--rnn-model=' + rnn, '--rnn-fields=surface,pos'],
                  ['--rnn-model=' + rnn, '--rnn-fields=surface,pos', '--rnn-nce-bias=5.6', '--rnn-unk-constant=-3.47',
                   '--rnn-unk-length=-2.93', '--feature-weight-perceptron=1', '--feature-weight-rnn=0.0176']):
        ref = _ref_cli(ref_tools, model, flags, txt)
        rc, out, err = _run(cli_emu, ['--model=' + model] + flags + [txt])
        assert rc == 0, err[-300:]
        assert ref != plain and _fold_unk_ties(out) == _fold_unk_ties(ref), flags
    # the same RNN embedded in the model by the reference's trainer gives the same analysis
    emb = _ref_cli(ref_tools, os.path.join(golden_dir, 'mini_rnn.jppmdl'), [], txt)
    assert _fold_unk_ties(out) == _fold_unk_ties(emb)
    rc, out, err = _run(cli_emu, ['--model=' + model, '--rnn-model=' + rnn + '.missing', '--rnn-fields=surface,pos', txt])
    assert rc == 1 and b'failed to load the RNN model' in err
    rc, out, err = _run(cli_emu, ['--model=' + model, '--rnn-model=' + rnn, '--rnn-fields=nosuchfield', txt])
    assert rc == 1 and b'could not find a field' in err


def test_double_array_builder_on_random_keys(emu_lib, tmp_path):
    """the host's darts-clone-layout builder (rnn_external.cc): 120 000 random keys incl. prefixes of each other,
    all found with their values, near-misses rejected, every probe inside the array"""
    host = os.path.join(ROOT, 'jumanpp_amd', 'host')
    srcs = [os.path.join(host, f) for f in sorted(os.listdir(host)) if f.endswith('.cc') and 'main' not in f]
    exe = os.path.join(str(tmp_path), 'da_builder_test')
    subprocess.check_call(['g++', '-std=c++14', '-O2', '-pthread', '-I' + os.path.join(ROOT, 'include'), '-I' + host,
                           os.path.join(ROOT, 'tests', 'host', 'da_builder_test.cc')] + srcs +
                          ['-o', exe, '-L' + os.path.dirname(emu_lib), '-l:libjppgpu_emu.so', '-Wl,-rpath,' + os.path.dirname(emu_lib)])
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-500:]


def test_cli_without_the_t0_memo_equals_the_reference(cli_emu, ref_tools, golden_dir):
    """the plain k_t0 (every node from scratch), which the per-entry memo normally replaces for the built-in spec, behind
    its developer switch: same bytes as the reference, with and without the memo"""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    m = os.path.join(golden_dir, 'mini_rnn.jppmdl')
    path = os.path.join(golden_dir, 'mini.txt')
    ref = _ref_cli(ref_tools, m, [], path)
    for memo in ('0', '1'):
        env = dict(os.environ, JPPGPU_DEV_T0_MEMO=memo)
        p = subprocess.run([cli_emu, '--model=' + m, path], capture_output=True, env=env)
        assert p.returncode == 0, p.stderr[-300:]
        assert p.stdout == ref, memo
        assert (b'T0 memo:' in p.stderr) == (memo == '1')


def _image_cache_case(cli, golden_dir, tmp_path):
    """the derived-image cache (host/derived_cache.h): the first process writes <model>.jppgpu-cache, the second maps it
    (T0 records handed to jppgpu_ctx_create, format table adopted) and prints the same bytes; a model file that changed
    (mtime) is not served from the old cache"""
    import shutil
    import time
    model = str(tmp_path / 'm.jppmdl')
    shutil.copy(os.path.join(golden_dir, 'mini_rnn.jppmdl'), model)
    txt = os.path.join(golden_dir, 'mini.txt')
    ref = open(os.path.join(golden_dir, 'mini_rnn.juman.txt'), 'rb').read()
    out1 = str(tmp_path / 'o1.txt')
    rc, _, err = _run(cli, ['--model=' + model, '--timing', '-o', out1, txt], image_cache=True)
    assert rc == 0 and b'image_cache=miss' in err and b'image cache written' in err, err[-400:]
    assert os.path.exists(model + '.jppgpu-cache')
    assert open(out1, 'rb').read() == ref
    for mode in (['-o', str(tmp_path / 'o2.txt')], []):    # the sharded and the stream pipeline
        rc, out, err = _run(cli, ['--model=' + model, '--timing'] + mode + [txt], image_cache=True)
        assert rc == 0 and b'image_cache=hit' in err and b'(image cache)' in err, err[-400:]
        got = open(mode[1], 'rb').read() if mode else out
        assert got == ref
    os.utime(model, (time.time() + 5, time.time() + 5))
    rc, out, err = _run(cli, ['--model=' + model, '--timing', txt], image_cache=True)
    assert rc == 0 and b'image_cache=miss' in err and out == ref, err[-400:]
    # a truncated cache file is ignored
    with open(model + '.jppgpu-cache', 'r+b') as f:
        f.truncate(1000)
    rc, out, err = _run(cli, ['--model=' + model, '--timing', txt], image_cache=True)
    assert rc == 0 and b'image_cache=miss' in err and out == ref, err[-400:]
    # (that run rewrote it) one flipped payload byte: the content hash of the header no longer matches -> ignored
    cache = model + '.jppgpu-cache'
    size = os.path.getsize(cache)
    with open(cache, 'r+b') as f:
        f.seek(size - 200)
        b = f.read(1)
        f.seek(size - 200)
        f.write(bytes([b[0] ^ 0x40]))
    rc, out, err = _run(cli, ['--model=' + model, '--timing', txt], image_cache=True)
    assert rc == 0 and b'image_cache=miss' in err and out == ref, err[-400:]
    # a cache that is writable by others is not trusted
    os.chmod(cache, 0o666)
    rc, out, err = _run(cli, ['--model=' + model, '--timing', txt], image_cache=True)
    assert rc == 0 and b'image_cache=miss' in err and out == ref, err[-400:]
    # a first run WITHOUT device text (lattice output) stores the records only; the next device-text run hits, builds
    # the table, and rewrites the cache with both parts; the run after that adopts the table
    os.unlink(cache)
    rc, out, err = _run(cli, ['--model=' + model, '--timing', '-s', '2', txt], image_cache=True)
    assert rc == 0 and b'image_cache=miss' in err and b'image cache written' in err, err[-400:]
    rc, out, err = _run(cli, ['--model=' + model, '--timing', txt], image_cache=True)
    assert rc == 0 and b'image_cache=hit' in err and b'(image cache)' not in err and b'image cache written' in err and out == ref, err[-400:]
    rc, out, err = _run(cli, ['--model=' + model, '--timing', txt], image_cache=True)
    assert rc == 0 and b'image_cache=hit' in err and b'(image cache)' in err and b'image cache written' not in err and out == ref, err[-400:]
    # ... and the lattice format's table is the third part: that first -s 2 run stored it, this one adopts it and prints
    # the host formatter's bytes (with its own score weights: they are not part of the cached table)
    rc, lat_host, err = _run(cli, ['--model=' + model, '--host-format', '-s', '2', txt])
    rc, lat_dev, err = _run(cli, ['--model=' + model, '--timing', '-s', '2', txt], image_cache=True)
    assert rc == 0 and b'device_lattice_format=1' in err and b'build_ms=0 (image cache)' in err and lat_dev == lat_host, err[-400:]
    rc, lat_host, err = _run(cli, ['--model=' + model, '--host-format', '--feature-weight-perceptron=0.5', '--feature-weight-rnn=2', '-s', '2', txt])
    rc, lat_dev, err = _run(cli, ['--model=' + model, '--timing', '--feature-weight-perceptron=0.5', '--feature-weight-rnn=2', '-s', '2', txt], image_cache=True)
    assert rc == 0 and b'build_ms=0 (image cache)' in err and lat_dev == lat_host, err[-400:]


def _lattice_from_fifo_case(cli, golden_dir, tmp_path):
    """-s N without --batch samples its input for the default batch size: only a regular file may be sampled (ADVICE r05:
    a FIFO lost its first megabyte, or the second open blocked for good)"""
    import threading
    model = os.path.join(golden_dir, 'mini_rnn.jppmdl')
    txt = os.path.join(golden_dir, 'mini.txt')
    rc, ref, err = _run(cli, ['--model=' + model, '-s', '3', txt])
    assert rc == 0 and ref, err[-300:]
    fifo = str(tmp_path / 'in.fifo')
    os.mkfifo(fifo)

    def feed():
        with open(fifo, 'wb') as f:
            f.write(open(txt, 'rb').read())
    t = threading.Thread(target=feed)
    t.start()
    try:
        p = subprocess.run([cli, '--model=' + model, '-s', '3', fifo], capture_output=True, timeout=120,
                           env=dict(os.environ, JPPGPU_NO_IMAGE_CACHE='1'))
    finally:
        t.join(timeout=10)
    assert p.returncode == 0 and p.stdout == ref, p.stderr[-300:]


def test_emulated_lattice_output_from_a_fifo(cli_emu, golden_dir, tmp_path):
    _lattice_from_fifo_case(cli_emu, golden_dir, tmp_path)


@pytest.mark.gpu
def test_gpu_lattice_output_from_a_fifo(cli_gpu, golden_dir, tmp_path):
    _lattice_from_fifo_case(cli_gpu, golden_dir, tmp_path)


def test_emulated_derived_image_cache(cli_emu, golden_dir, tmp_path):
    _image_cache_case(cli_emu, golden_dir, tmp_path)


@pytest.mark.gpu
def test_gpu_derived_image_cache(cli_gpu, golden_dir, tmp_path):
    _image_cache_case(cli_gpu, golden_dir, tmp_path)
