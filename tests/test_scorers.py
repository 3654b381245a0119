"""ScorerDef::others beyond the model's RNN and the per-connection ScorePlugin against the
reference built with the SAME test scorer / test plugin (oracle/ref_dump.cc `top1x`: TestScorer, TestPlugin):

* jppgpu_analyze_batch_scored: a host ScoreComputer fills its slot of the score cells, the device re-makes the beam
  totals and the EOS beam (k_adjust.h: adjustBeamScores / remakeEosBeam) -- perceptron only and behind the RNN;
* jppgpu_analyze_batch_pairs: an amount per (left node, right node) connection, applied inside the sweep where
  applyPluginToPrescores / applyPluginToGbeam act.

Compared: the packed top-1 path of every sentence (EntryPtr, start, end) and the EOS beam totals, bit for bit."""
import os
import subprocess

import numpy as np
import pytest

import jumanpp_amd as J

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ITEM_DT = np.dtype([('eptr', '<i4'), ('start', '<u2'), ('end', '<u2')])


def _ref_top1x(ref_tools, model, lines, tmp, weight, plugin, beams=None):
    out = os.path.join(tmp, 'x.bin')
    data = ('\n'.join(lines) + '\n').encode('utf-8')
    subprocess.run([os.path.join(ref_tools, 'ref_dump'), 'top1x', model, out, 'none' if weight is None else repr(weight), '1' if plugin else '0'] +
                   [str(b) for b in (beams or [])], input=data, check=True, stderr=subprocess.DEVNULL)
    raw = open(out, 'rb').read()
    magic, n = np.frombuffer(raw, dtype='<u4', count=2)
    assert magic == 0x31504f54 and n == len(lines)
    pos, paths = 8, []
    for _ in range(n):
        st, c = np.frombuffer(raw, dtype='<u4', count=2, offset=pos)
        pos += 8
        paths.append(None if st else np.frombuffer(raw, dtype=ITEM_DT, count=int(c), offset=pos).copy())
        pos += 8 * int(c)
    t = open(out + '.totals', 'rb').read()
    tn, tb = np.frombuffer(t, dtype='<u4', count=2)
    totals = np.frombuffer(t, dtype='<u4', count=int(tn) * int(tb), offset=8).reshape(int(tn), int(tb))
    return paths, totals


def _device_paths(res):
    r = res.fetch(full=True)
    paths, totals = [], []
    for s in range(r.n):
        if r.status[s] != 0:
            paths.append(None)
            totals.append(None)
            continue
        nb, pl = int(r.node_base[s]), int(r.path_len[s])
        nodes = [int(r.path_nodes[nb + k]) for k in range(pl)][::-1][:-1] if pl else []   # text order, EOS dropped
        paths.append(np.array([(r.nodes[nb + k]['eptr'], r.nodes[nb + k]['start'], r.nodes[nb + k]['end']) for k in nodes], dtype=ITEM_DT))
        eos = r.beams[nb + int(r.nnodes[s]) - 1]
        totals.append(np.array([0 if (x['left'] == 0xffff and x['beam'] == 0xffff) else np.float32(x['total']).view('<u4') for x in eos], dtype='<u4'))
    return paths, totals


def _compare(dev, ref):
    dp, dt = dev
    rp, rt = ref
    bad = []
    for s in range(len(rp)):
        if rp[s] is None:
            if dp[s] is not None:
                bad.append((s, 'status'))
            continue
        if dp[s] is None or len(dp[s]) != len(rp[s]) or not np.array_equal(dp[s], rp[s]):
            bad.append((s, 'path'))
        elif len(rp[s]) and not np.array_equal(dt[s][:rt.shape[1]], rt[s]):
            bad.append((s, 'totals', [hex(int(x)) for x in dt[s]], [hex(int(x)) for x in rt[s]]))
    return bad


def test_scorer(lat, idx, cells):
    """TestScorer of oracle/ref_dump.cc: every (right node, global-beam element) cell of every boundary"""
    G = lat['gbeam']
    for s in range(lat['n']):
        if lat['status'][s] != 0:
            continue
        nb, bb = int(lat['node_base'][s]), int(lat['bnd_base'][s])
        for b in range(2, int(lat['ncp'][s]) + 3):
            R, first = int(lat['bnd_count'][bb + b]), int(lat['bnd_first'][bb + b])
            if R == 0 or int(lat['ncp'][s]) == 0:
                continue   # (no nodes start here: the boundary is not scored and has no global beam)
            ngb = int(lat['gbeam_count'][bb + b])
            ef = int(lat['end_first'][bb + b])
            for i in range(ngb):
                ln = lat['nodes'][nb + int(lat['end_nodes'][nb + ef + int(lat['gbeam_entries'][bb + b][i]['left'])])]
                base = (-0.25 if ln['eptr'] < 0 else 0.0) - 0.0625 * float(int(ln['end']) - int(ln['start']))
                for r in range(R):
                    rn = lat['nodes'][nb + first + r]
                    cells[nb + first + r, i, idx] = np.float32(base + (0.125 if rn['eptr'] < 0 else 0.0))


def test_plugin(lat, pen):
    """TestPlugin of oracle/ref_dump.cc: one amount per (left, right) pair"""
    for s in range(lat['n']):
        if lat['status'][s] != 0:
            continue
        nb, bb = int(lat['node_base'][s]), int(lat['bnd_base'][s])
        for b in range(2, int(lat['ncp'][s]) + 3):
            R, first = int(lat['bnd_count'][bb + b]), int(lat['bnd_first'][bb + b])
            L, ef = int(lat['end_count'][bb + b]), int(lat['end_first'][bb + b])
            p0 = int(lat['pair_base'][bb + b])
            for l in range(L):
                ln = lat['nodes'][nb + int(lat['end_nodes'][nb + ef + l])]
                for r in range(R):
                    rn = lat['nodes'][nb + first + r]
                    a = (1.0 if ln['eptr'] < 0 else 0.0) + (0.5 if (int(rn['end']) - int(rn['start']) > 2 and (int(ln['start']) & 1)) else 0.0)
                    pen[p0 + l * R + r] = a


test_scorer.__test__ = False
test_plugin.__test__ = False


def _lines(golden_dir):
    return [l.rstrip('\n') for l in open(os.path.join(golden_dir, 'mini.txt'), encoding='utf-8')]


def check_host_scorer(lib, ref_tools, golden_dir, tmp, model, weight, beams=None):
    lines = _lines(golden_dir)
    cfg = dict(zip(('beam', 'global_beam', 'right_check', 'right_beam'), beams)) if beams else {}
    ctx = J.Context(os.path.join(golden_dir, model + '.img'), lib_path=lib, host_scorer_weights=[weight], **cfg)
    dev = _device_paths(ctx.analyze_scored(lines, [test_scorer]))
    ref = _ref_top1x(ref_tools, os.path.join(golden_dir, model + '.jppmdl'), lines, tmp, weight, False, beams)
    bad = _compare(dev, ref)
    assert not bad, bad[:5]
    # the scorer must matter: without it the reference takes other paths
    plain = _ref_top1x(ref_tools, os.path.join(golden_dir, model + '.jppmdl'), lines, tmp, None, False, beams)
    assert any(a is not None and b is not None and not np.array_equal(a, b) for a, b in zip(plain[0], ref[0]))


def check_pair_plugin(lib, ref_tools, golden_dir, tmp, model, beams=None):
    lines = _lines(golden_dir)
    cfg = dict(zip(('beam', 'global_beam', 'right_check', 'right_beam'), beams)) if beams else {}
    ctx = J.Context(os.path.join(golden_dir, model + '.img'), lib_path=lib, **cfg)
    dev = _device_paths(ctx.analyze_pairs(lines, test_plugin))
    ref = _ref_top1x(ref_tools, os.path.join(golden_dir, model + '.jppmdl'), lines, tmp, None, True, beams)
    bad = _compare(dev, ref)
    assert not bad, bad[:5]
    plain = _ref_top1x(ref_tools, os.path.join(golden_dir, model + '.jppmdl'), lines, tmp, None, False, beams)
    assert any(a is not None and b is not None and not np.array_equal(a, b) for a, b in zip(plain[0], ref[0]))


CASES = [('mini', 0.5, None), ('mini_rnn', 2.0, None), ('mini', 1.5, [8, 12, 2, 6])]


@pytest.mark.parametrize('model,weight,beams', CASES)
def test_host_scorer_in_scorerdef_others(emu_lib, ref_tools, golden_dir, tmp_path, model, weight, beams):
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    check_host_scorer(emu_lib, ref_tools, golden_dir, str(tmp_path), model, weight, beams)


@pytest.mark.parametrize('model,beams', [('mini', None), ('mini_rnn', None), ('mini', [6, 10, 2, 4])])
def test_per_connection_plugin(emu_lib, ref_tools, golden_dir, tmp_path, model, beams):
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    check_pair_plugin(emu_lib, ref_tools, golden_dir, str(tmp_path), model, beams)


@pytest.mark.gpu
@pytest.mark.parametrize('model,weight,beams', CASES)
def test_gpu_host_scorer_in_scorerdef_others(gpu_lib, ref_tools, golden_dir, tmp_path, model, weight, beams):
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    check_host_scorer(gpu_lib, ref_tools, golden_dir, str(tmp_path), model, weight, beams)


@pytest.mark.gpu
@pytest.mark.parametrize('model,beams', [('mini', None), ('mini_rnn', None), ('mini', [6, 10, 2, 4])])
def test_gpu_per_connection_plugin(gpu_lib, ref_tools, golden_dir, tmp_path, model, beams):
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    check_pair_plugin(gpu_lib, ref_tools, golden_dir, str(tmp_path), model, beams)


def _host_api_case(lib, ref_tools, golden_dir, tmp_path):
    """the C++14 mirror (GpuAnalyzer + ScorerDef::others + ScorePlugin::connectionPenalties) through tests/host/scorer_api_test.cc"""
    host = os.path.join(ROOT, 'jumanpp_amd', 'host')
    srcs = [os.path.join(host, f) for f in sorted(os.listdir(host)) if f.endswith('.cc') and 'main' not in f]
    exe = os.path.join(str(tmp_path), 'scorer_api_test')
    libname = os.path.basename(lib)
    subprocess.check_call(['g++', '-std=c++14', '-O2', '-pthread', '-I' + os.path.join(ROOT, 'include'), '-I' + host,
                           os.path.join(ROOT, 'tests', 'host', 'scorer_api_test.cc')] + srcs +
                          ['-o', exe, '-L' + os.path.dirname(lib), '-l:' + libname, '-Wl,-rpath,' + os.path.dirname(lib),
                           '-Wl,--allow-shlib-undefined'])
    lines = _lines(golden_dir)
    data = ('\n'.join(lines) + '\n').encode('utf-8')
    for model, weight, plugin in (('mini', 0.5, False), ('mini_rnn', 2.0, False), ('mini', None, True), ('mini_rnn', None, True)):
        out = os.path.join(str(tmp_path), 'h.bin')
        args = [exe, os.path.join(golden_dir, model + '.jppmdl'), out, 'none' if weight is None else repr(weight), '1' if plugin else '0']
        if model == 'mini_rnn':
            args.append('--rnn')
        subprocess.run(args, input=data, check=True)
        ref = os.path.join(str(tmp_path), 'x.bin')
        _ref_top1x(ref_tools, os.path.join(golden_dir, model + '.jppmdl'), lines, str(tmp_path), weight, plugin)
        assert open(out, 'rb').read() == open(ref, 'rb').read(), (model, weight, plugin)


def test_host_mirror_of_scorerdef_and_connection_plugin(emu_lib, ref_tools, golden_dir, tmp_path):
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    _host_api_case(emu_lib, ref_tools, golden_dir, tmp_path)


@pytest.mark.gpu
def test_gpu_host_mirror_of_scorerdef_and_connection_plugin(gpu_lib, ref_tools, golden_dir, tmp_path):
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    _host_api_case(gpu_lib, ref_tools, golden_dir, tmp_path)
