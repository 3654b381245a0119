"""GPU parity tests (pytest -m gpu): the HIP path, called through the C ABI,
against (a) the committed golden vectors of the real reference and (b) fresh
golden vectors produced on this box by oracle/_ref (the real reference binary
travels with the snapshot) for a larger seeded synthetic workload."""
import os
import subprocess

import numpy as np
import pytest

import golden_io as G
import jumanpp_amd as J

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _compare_all(res, gold, meta, n):
    errs = []
    for s in range(n):
        errs += G.compare_sentence(res, s, gold[s], meta)
    return errs


def test_gpu_matches_reference_goldens(gpu_lib, golden_dir):
    ctx = J.Context(os.path.join(golden_dir, 'mini.img'), lib_path=gpu_lib)
    lines = [l.rstrip('\n') for l in open(os.path.join(golden_dir, 'mini.txt'), encoding='utf-8')]
    meta, gold = G.read_gold(os.path.join(golden_dir, 'mini.gold'))
    res = ctx.analyze(lines).fetch(full=True)
    errs = _compare_all(res, gold, meta, len(lines))
    assert not errs, errs[:10]


def test_gpu_matches_reference_other_beam_config(gpu_lib, golden_dir):
    ctx = J.Context(os.path.join(golden_dir, 'mini.img'), lib_path=gpu_lib, beam=3, global_beam=4,
                    right_check=2, right_beam=3)
    lines = [l.rstrip('\n') for l in open(os.path.join(golden_dir, 'mini.txt'), encoding='utf-8')]
    meta, gold = G.read_gold(os.path.join(golden_dir, 'mini_b3.gold'))
    res = ctx.analyze(lines).fetch(full=True)
    errs = _compare_all(res, gold, meta, len(lines))
    assert not errs, errs[:10]


@pytest.mark.parametrize('image,gold_name', [('mini.img', 'mini_b4g12.gold'), ('mini_rnn.img', 'mini_rnn_b4g12.gold')])
def test_gpu_matches_reference_quickselect_goldens(gpu_lib, golden_dir, image, gold_name):
    """beam 4 / global beam 12: makeT0Beam's util::partition branch (also at remakeEosBeam with the RNN)"""
    ctx = J.Context(os.path.join(golden_dir, image), lib_path=gpu_lib, beam=4, global_beam=12, right_check=1, right_beam=4)
    lines = [l.rstrip('\n') for l in open(os.path.join(golden_dir, 'mini.txt'), encoding='utf-8')][:8]
    meta, gold = G.read_gold(os.path.join(golden_dir, gold_name))
    assert meta['nsent'] == 8 and meta['beam'] == 4 and meta['gbeam'] == 12
    res = ctx.analyze(lines).fetch(full=True)
    errs = _compare_all(res, gold, meta, len(lines))
    assert not errs, errs[:10]


def test_gpu_matches_reference_with_rnn(gpu_lib, golden_dir):
    ctx = J.Context(os.path.join(golden_dir, 'mini_rnn.img'), lib_path=gpu_lib)
    lines = [l.rstrip('\n') for l in open(os.path.join(golden_dir, 'mini.txt'), encoding='utf-8')]
    meta, gold = G.read_gold(os.path.join(golden_dir, 'mini_rnn.gold'))
    assert meta['nscorers'] == 2
    res = ctx.analyze(lines).fetch(full=True)
    errs = _compare_all(res, gold, meta, len(lines))
    assert not errs, errs[:10]


def test_gpu_status_codes(gpu_lib, golden_dir):
    ctx = J.Context(os.path.join(golden_dir, 'mini.img'), lib_path=gpu_lib)
    sents = [b'\xe3\x81', b'ok', b'\xff\xfe', ('あ' * 1400).encode('utf-8'), b'']
    res = ctx.analyze(sents).fetch(full=True)
    assert list(res.status) == [2, 0, 2, 1, 0]


def _fresh_workload(ref_tools, tmp, n_entries, n_lines, exp, seed, length=40, rnn=None, beams=None, join=0,
                    extra_dict='', extra_lines=()):
    mdic = os.path.join(tmp, 'w.mdic')
    with open(mdic, 'w', encoding='utf-8') as f:
        subprocess.check_call(['python3', os.path.join(ROOT, 'tools', 'gen_dict.py'), str(n_entries), '--seed', str(seed)],
                              stdout=f)
        if extra_dict:
            # the corpus generator draws its words from the dictionary without the extra entries
            f.flush()
            import shutil
            shutil.copyfile(mdic, os.path.join(tmp, 'w.base.mdic'))
            f.write(extra_dict)
    subprocess.check_call([os.path.join(ref_tools, 'jpp_jumandic_bootstrap'), mdic, os.path.join(tmp, 'w.seed')],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([os.path.join(ref_tools, 'ref_dump'), 'mkmodel', os.path.join(tmp, 'w.seed'),
                           os.path.join(tmp, 'w.model'), str(exp), str(seed), '0.1'])
    if rnn is not None:
        hidden, vocab = rnn
        subprocess.check_call(['python3', os.path.join(ROOT, 'tools', 'gen_rnn.py'), mdic, os.path.join(tmp, 'rnn'),
                               '--vocab', str(vocab), '--hidden', str(hidden), '--maxent-size', str(1 << 18),
                               '--seed', str(seed)], stdout=subprocess.DEVNULL)
        os.rename(os.path.join(tmp, 'w.model'), os.path.join(tmp, 'p.model'))
        subprocess.check_call([os.path.join(ref_tools, 'jumanpp_v2_train'), '--model-input=' + os.path.join(tmp, 'p.model'),
                               '--model-output=' + os.path.join(tmp, 'w.model'), '--rnn-model=' + os.path.join(tmp, 'rnn'),
                               '--rnn-fields=surface,pos', '--rnn-nce-bias=5.6', '--rnn-unk-constant=-3.47',
                               '--rnn-unk-length=-2.93', '--feature-weight-perceptron=1', '--feature-weight-rnn=0.0176'],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([os.path.join(ref_tools, 'ref_dump'), 'export', os.path.join(tmp, 'w.model'),
                           os.path.join(tmp, 'w.img')], stderr=subprocess.DEVNULL)
    txt = os.path.join(tmp, 'w.txt')
    with open(txt, 'w', encoding='utf-8') as f:
        subprocess.check_call(['python3', os.path.join(ROOT, 'tools', 'gen_corpus.py'),
                               os.path.join(tmp, 'w.base.mdic') if extra_dict else mdic, str(n_lines), '--seed',
                               str(seed + 1), '--oov', '0.08', '--len', str(length)], stdout=f)
    if extra_lines:
        with open(txt, 'a', encoding='utf-8') as f:
            f.write('\n'.join(extra_lines) + '\n')
    if join:
        # every other output line is `join` generated sentences glued together
        src = [l.rstrip('\n') for l in open(txt, encoding='utf-8')]
        out, i = [], 0
        while i < len(src):
            out.append(src[i])
            out.append(''.join(src[i + 1:i + 1 + join]))
            i += 1 + join
        open(txt, 'w', encoding='utf-8').write('\n'.join(l for l in out if l) + '\n')
    with open(txt, 'rb') as f:
        subprocess.check_call([os.path.join(ref_tools, 'ref_dump'), 'dump', os.path.join(tmp, 'w.model'),
                               os.path.join(tmp, 'w.gold')] + [str(x) for x in (beams or [])],
                              stdin=f, stderr=subprocess.DEVNULL)
    lines = [l.rstrip('\n') for l in open(txt, encoding='utf-8')]
    return os.path.join(tmp, 'w.img'), lines, os.path.join(tmp, 'w.gold')


def test_gpu_rnn_staged_and_unstaged_sentences_in_one_batch(gpu_lib, ref_tools, tmp_path):
    """k_rnn_prep / k_rnn_score stage a sentence's beam records in LDS when it has at most 45 codepoints
    (global beam 6) and read them from HBM otherwise: both kinds in one batch (40 and ~240 codepoints),
    every RNN score and the re-ranked EOS beam bit-identical to the reference."""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    img, lines, gold_path = _fresh_workload(ref_tools, str(tmp_path), 30000, 700, 20, 91, rnn=(128, 3000), join=6)
    assert max(len(l) for l in lines) > 200 and min(len(l) for l in lines) < 60
    ctx = J.Context(img, lib_path=gpu_lib)
    meta, gold = G.read_gold(gold_path)
    res = ctx.analyze(lines).fetch(full=True)
    errs = _compare_all(res, gold, meta, len(lines))
    assert not errs, (len(errs), errs[:10])


def test_gpu_matches_live_reference_on_fresh_workload(gpu_lib, ref_tools, tmp_path):
    """2000 fresh 40-codepoint sentences, 30k-entry dictionary, 2^20 random weights:
    every node, pattern, T0 score, global beam, beam slot and top-1 path must be
    bit-identical to the reference running on this box's CPU."""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    img, lines, gold_path = _fresh_workload(ref_tools, str(tmp_path), 30000, 2000, 20, 77)
    ctx = J.Context(img, lib_path=gpu_lib)
    meta, gold = G.read_gold(gold_path)
    res = ctx.analyze(lines).fetch(full=True)
    errs = _compare_all(res, gold, meta, len(lines))
    assert not errs, (len(errs), errs[:10])


@pytest.mark.parametrize('hidden,vocab,n_lines', [(128, 8000, 1000), (100, 3000, 300), (48, 3000, 300),
                                                 (200, 3000, 300)])
def test_gpu_matches_live_reference_with_rnn(gpu_lib, ref_tools, tmp_path, hidden, vocab, n_lines):
    """config[2] shape: perceptron + RNNLM, fresh sentences vs the live reference.  E=128 is the
    benchmark shape; 100 and 48 exercise the zero-padded LDS layouts, 200 the streamed-W kernel."""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    img, lines, gold_path = _fresh_workload(ref_tools, str(tmp_path), 30000, n_lines, 20, 55, rnn=(hidden, vocab))
    ctx = J.Context(img, lib_path=gpu_lib)
    meta, gold = G.read_gold(gold_path)
    assert meta['nscorers'] == 2
    res = ctx.analyze(lines).fetch(full=True)
    errs = _compare_all(res, gold, meta, len(lines))
    assert not errs, (len(errs), errs[:10])


@pytest.mark.parametrize('beams,rnn', [([5, 10, 1, 5], None), ([4, 12, 2, 6], (64, 3000))])
def test_gpu_partition_branch_of_make_t0_beam(gpu_lib, ref_tools, tmp_path, beams, rnn):
    """global beam > beam*4/3 (util::partition before std::sort in makeT0Beam / remakeEosBeam)"""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    img, lines, gold_path = _fresh_workload(ref_tools, str(tmp_path), 20000, 500, 18, 171, rnn=rnn, beams=beams)
    ctx = J.Context(img, lib_path=gpu_lib, beam=beams[0], global_beam=beams[1], right_check=beams[2], right_beam=beams[3])
    meta, gold = G.read_gold(gold_path)
    res = ctx.analyze(lines).fetch(full=True)
    errs = _compare_all(res, gold, meta, len(lines))
    assert not errs, (len(errs), errs[:10])


def test_gpu_long_sentences(gpu_lib, ref_tools, tmp_path):
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    img, lines, gold_path = _fresh_workload(ref_tools, str(tmp_path), 8000, 100, 18, 91, length=220)
    ctx = J.Context(img, lib_path=gpu_lib)
    meta, gold = G.read_gold(gold_path)
    res = ctx.analyze(lines).fetch(full=True)
    errs = _compare_all(res, gold, meta, len(lines))
    assert not errs, (len(errs), errs[:10])


def test_gpu_config5_wide_beam_long_sentences_rnn(gpu_lib, ref_tools, tmp_path):
    """BASELINE configs[4] shape: beam=32, global beam 32, >= 200-codepoint sentences, RNN on.
    More than 16 beam candidates means libstdc++'s introsort tie order must be replayed exactly."""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    img, lines, gold_path = _fresh_workload(ref_tools, str(tmp_path), 20000, 120, 18, 123, length=210,
                                            rnn=(128, 5000), beams=[32, 32, 1, 32])
    ctx = J.Context(img, lib_path=gpu_lib, beam=32, global_beam=32, right_check=1, right_beam=32)
    meta, gold = G.read_gold(gold_path)
    assert meta['beam'] == 32 and meta['nscorers'] == 2
    res = ctx.analyze(lines).fetch(full=True)
    errs = _compare_all(res, gold, meta, len(lines))
    assert not errs, (len(errs), errs[:10])


def test_gpu_full_beam(gpu_lib, ref_tools, tmp_path):
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    img, lines, gold_path = _fresh_workload(ref_tools, str(tmp_path), 20000, 300, 18, 321, beams=[5, 0, 0, 0])
    ctx = J.Context(img, lib_path=gpu_lib, beam=5, global_beam=0, right_check=0, right_beam=0)
    meta, gold = G.read_gold(gold_path)
    res = ctx.analyze(lines).fetch(full=True)
    errs = _compare_all(res, gold, meta, len(lines))
    assert not errs, (len(errs), errs[:10])


def test_gpu_batch_split_invariance(gpu_lib, golden_dir):
    """size-independent property: analysing a batch in one call or sentence by
    sentence gives identical top-1 paths and totals."""
    ctx = J.Context(os.path.join(golden_dir, 'mini.img'), lib_path=gpu_lib)
    lines = [l.rstrip('\n') for l in open(os.path.join(golden_dir, 'mini.txt'), encoding='utf-8')] * 40
    res = ctx.analyze(lines).fetch(full=True)
    whole = []
    for s in range(len(lines)):
        nb = int(res.node_base[s])
        pl = int(res.path_len[s])
        nodes = res.nodes[nb + res.path_nodes[nb:nb + pl]]
        tot = res.beams[nb + int(res.nnodes[s]) - 1][0]['total'] if res.nnodes[s] else 0.0
        whole.append((tuple((int(x['start']), int(x['end']), int(x['eptr'])) for x in nodes if x['eptr'] >= 0 or True), float(tot)))
    for s in range(0, 28):
        assert whole[s] == whole[s + 28 * 7]
    res1 = ctx.analyze(lines[3:4]).fetch(full=True)
    pl = int(res1.path_len[0])
    nb1 = int(res1.node_base[0])
    nodes = res1.nodes[nb1 + res1.path_nodes[nb1:nb1 + pl]]
    one = tuple((int(x['start']), int(x['end']), int(x['eptr'])) for x in nodes)
    assert one == whole[3][0]


def _packed(ctx, lines):
    """(offsets, items) of the packed top-1 result of one batch, via the harness' host fetch"""
    res = ctx.analyze(lines).fetch(full=False)
    out = []
    for s in range(len(lines)):
        nb = int(res.node_base[s])
        pl = int(res.path_len[s])
        nodes = res.nodes[nb + res.path_nodes[nb:nb + pl][::-1]][:-1] if pl else res.nodes[:0]  # text order, EOS dropped
        out.append(nodes.tobytes())
    return res, out


@pytest.mark.parametrize('model', ['mini.img', 'mini_rnn.img'])
def test_gpu_full_batch_size_invariants(gpu_lib, golden_dir, model):
    """BASELINE batch size (65 536 sentences of 40 codepoints) through size-independent properties:
    every path tiles its sentence exactly, the analysis is deterministic, and analysing the batch in one
    call, in two halves or in shuffled order gives the same per-sentence result."""
    import random
    import subprocess as sp
    import hashlib
    tmp = os.environ.get('TMPDIR', '/tmp')
    mdic = os.path.join(tmp, 'fullsize.mdic')
    with open(mdic, 'w', encoding='utf-8') as f:
        sp.check_call(['python3', os.path.join(ROOT, 'tools', 'gen_dict.py'), '2500', '--seed', '11'], stdout=f)
    out = sp.check_output(['python3', os.path.join(ROOT, 'tools', 'gen_corpus.py'), mdic, '65536', '--seed', '99', '--oov', '0.08',
                           '--len', '40'])
    lines = out.decode('utf-8').split('\n')[:65536]
    assert len(lines) == 65536
    ctx = J.Context(os.path.join(golden_dir, model), lib_path=gpu_lib)
    res, whole = _packed(ctx, lines)
    assert int((res.status != 0).sum()) == 0
    # the top-1 path tiles the input: consecutive spans from 0 to the number of codepoints
    for s in range(0, 65536, 97):
        spans = np.frombuffer(whole[s], dtype=[('eptr', '<i4'), ('start', '<u2'), ('end', '<u2')])
        assert spans['start'][0] == 0 and spans['end'][-1] == len(lines[s])
        assert (spans['start'][1:] == spans['end'][:-1]).all()
    digest = hashlib.sha256(b''.join(whole)).hexdigest()
    # determinism
    _, again = _packed(ctx, lines)
    assert hashlib.sha256(b''.join(again)).hexdigest() == digest
    # split invariance
    _, a = _packed(ctx, lines[:30000])
    _, b = _packed(ctx, lines[30000:])
    assert hashlib.sha256(b''.join(a + b)).hexdigest() == digest
    # order invariance
    perm = list(range(65536))
    random.Random(5).shuffle(perm)
    _, sh = _packed(ctx, [lines[i] for i in perm])
    back = [None] * 65536
    for k, i in enumerate(perm):
        back[i] = sh[k]
    assert hashlib.sha256(b''.join(back)).hexdigest() == digest


@pytest.mark.parametrize('model', ['mini.img', 'mini_rnn.img'])
def test_gpu_top1_fetch_equals_basic_fetch(gpu_lib, golden_dir, model):
    """JPPGPU_FETCH_TOP1: the device-compacted top-1 tables equal the path nodes of the basic fetch
    (3 000 sentences incl. empty and malformed ones)"""
    import test_cpu_parity as tc
    ctx = J.Context(os.path.join(golden_dir, model), lib_path=gpu_lib)
    lines = [l.rstrip('\n').encode('utf-8') for l in open(os.path.join(golden_dir, 'mini.txt'), encoding='utf-8')]
    lines = (lines + [b'', b'\xe3\x81', lines[3][:9]]) * 100
    assert tc.check_top1_fetch_equals_basic_fetch(ctx, lines) > 10000


@pytest.mark.parametrize('rnn', [None, (64, 3000)])
def test_gpu_maximum_size_sentences(gpu_lib, ref_tools, tmp_path, rnn):
    """1 300-codepoint sentences (just under maxInputBytes = 4096), perceptron and RNN: the long-sentence
    branches of every kernel against the live reference"""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    img, lines, gold_path = _fresh_workload(ref_tools, str(tmp_path), 2500, 6, 14, 5, length=1300, rnn=rnn)
    ctx = J.Context(img, lib_path=gpu_lib)
    meta, gold = G.read_gold(gold_path)
    res = ctx.analyze(lines).fetch(full=True)
    assert int((res.status != 0).sum()) == 0
    errs = []
    for s in range(len(lines)):
        errs += G.compare_sentence(res, s, gold[s], meta)
    assert not errs, errs[:10]


def test_gpu_nbest_fetch_equals_full_lattice(gpu_lib, golden_dir):
    """jppgpu_result_fetch_nbest on the device against the full lattice arrays (1 500 sentences, RNN model)"""
    import test_cpu_parity as tc
    ctx = J.Context(os.path.join(golden_dir, 'mini_rnn.img'), lib_path=gpu_lib)
    lines = [l.rstrip('\n').encode('utf-8') for l in open(os.path.join(golden_dir, 'mini.txt'), encoding='utf-8')]
    lines = (lines + [b'', b'\xe3\x81']) * 50
    assert tc.check_nbest_fetch_equals_full_lattice(ctx, lines, 5) > 50000


@pytest.mark.parametrize('beams', [[5, 6, 1, 5], [20, 24, 2, 20]])
def test_gpu_lattice_wider_than_the_lds_staging(gpu_lib, ref_tools, tmp_path, beams):
    """> 512 nodes starting at one boundary (k_sweep<*, 0>: per-right-node arrays in HBM) vs the live reference"""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_cpu_parity as tc
    img, lines, gold_path = tc._wide_boundary_workload(ref_tools, str(tmp_path), beams, n_homographs=700)
    ctx = J.Context(img, lib_path=gpu_lib, beam=beams[0], global_beam=beams[1], right_check=beams[2], right_beam=beams[3])
    meta, gold = G.read_gold(gold_path)
    res = ctx.analyze(lines).fetch(full=True)
    assert list(res.status) == [0, 0, 0] and int(res.bnd_count.max()) > 512
    errs = _compare_all(res, gold, meta, len(lines))
    assert not errs, errs[:10]


@pytest.mark.gpu
def test_gpu_top1_ngram_features_match_the_reference_trainer(gpu_lib, ref_tools, golden_dir, tmp_path):
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_cpu_parity as tc
    tc.check_top1_ngrams_against_reference(gpu_lib, ref_tools, golden_dir, tmp_path)


@pytest.mark.gpu
def test_gpu_top1_ngram_features_on_fresh_workload(gpu_lib, ref_tools, golden_dir, tmp_path):
    """1500 fresh sentences, 30k-entry dictionary: ~37 000 path positions x 73 features against the reference."""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_cpu_parity as tc
    img, lines, _ = _fresh_workload(ref_tools, str(tmp_path), 30000, 1500, 20, 88)
    tc.check_top1_ngrams_against_reference(gpu_lib, ref_tools, golden_dir, tmp_path,
                                           workload=(os.path.join(str(tmp_path), 'w.model'), img, os.path.join(str(tmp_path), 'w.txt')),
                                           min_checked=20000)


@pytest.mark.gpu
def test_gpu_weight_upload(gpu_lib, golden_dir):
    import test_cpu_parity as tc
    tc.check_set_weights(gpu_lib, golden_dir)


@pytest.mark.gpu
def test_gpu_rnn_tiny_and_degenerate_batches(gpu_lib, golden_dir):
    """same as the emulator test: batches of 1, 3 and 33 sentences through the lock-step RNN kernels, a batch of
    only empty / invalid sentences"""
    img = os.path.join(golden_dir, 'mini_rnn.img')
    lines = [l.rstrip('\n') for l in open(os.path.join(golden_dir, 'mini.txt'), encoding='utf-8')]
    ctx = J.Context(img, lib_path=gpu_lib)
    ref = ctx.analyze(lines).fetch(full=True)

    def path_of(res, s):
        nb, pl = int(res.node_base[s]), int(res.path_len[s])
        return [tuple(res.nodes[nb + int(k)]) for k in res.path_nodes[nb:nb + pl]], \
               [float(x) for x in res.beams[nb + int(res.nnodes[s]) - 1]['total']] if int(res.nnodes[s]) > 2 else []

    for pick in ([5], [0, 9, 13], list(range(len(lines))) + [0, 1, 2, 3, 4]):
        sub = [lines[i] for i in pick]
        r = ctx.analyze(sub).fetch(full=True)
        for j, i in enumerate(pick):
            assert int(r.status[j]) == int(ref.status[i])
            assert path_of(r, j) == path_of(ref, i), (pick, j)
    r = ctx.analyze([b'', b'\xff\xfe', b'', b'\xe3\x81']).fetch(full=True)
    assert list(r.status) == [0, 2, 0, 2] and int(r.path_len.sum()) == 0


@pytest.mark.parametrize('rnn', [True, False])
def test_gpu_headline_shape_through_the_bench_path_vs_live_reference(gpu_lib, ref_tools, rnn):
    """The exact headline shape: 2 x 65 536 sentences of bench.py's own model (300 k dictionary entries, 2^22
    weights, E = 128 RNN) and corpus generator through the path bench.py times -- jppgpu_analyze_batch_device on
    device-resident input + jppgpu_result_pack -- and the packed (EntryPtr, start, end) of every morpheme of all
    131 072 sentences equal to the reference Analyzer::analyze (oracle/_ref `ref_dump top1`, one process per core).
    65 536-sentence batches are what sizes the workspaces from batch totals and what k_rnn_order_* / k_rnn_chain
    group by chain length across; smaller batches exercise that code differently."""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import argparse
    import bench
    import hipmem
    args = argparse.Namespace(dict_entries=300000, weights_exp=22, seed=20260925, rnn=True, rnn_hidden=128,
                              rnn_vocab=30000, sent_len=40)
    cache = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'jppgpu_bench_cache')
    mdic, model, img = bench.make_workload(args, cache)
    batch = 65536
    corpus = bench.make_corpus(args, mdic, cache, 2 * batch, args.seed + 1)
    batches = bench.load_batches(corpus, batch, np)
    assert len(batches) == 2
    ctx = J.Context(img, beam=5, global_beam=6, right_check=1, right_beam=5, use_rnn=None if rnn else False, lib_path=gpu_lib)
    cap = batch * 41
    d_offs = hipmem.DeviceArray(batch + 1, np.uint32)
    d_items = hipmem.DeviceArray(cap * 2, np.int32)
    ref_dir, _ = bench.reference_build()
    ref_model = model if rnn else model + '.perceptron'
    total_bad = []
    for bi, (text, offs) in enumerate(batches):
        t = hipmem.DeviceArray.from_numpy(np.frombuffer(text, dtype=np.uint8))
        o = hipmem.DeviceArray.from_numpy(offs.astype(np.uint32))
        r = ctx.analyze_device(t.ptr, o.ptr, batch, len(text), None)   # (the context's default stream)
        r.pack(d_offs.ptr, d_items.ptr, cap)
        ho = d_offs.to_numpy()
        hi = d_items.to_numpy(2 * int(ho[-1])).reshape(-1, 2)
        r.release()
        t.free()
        o.free()
        rs, ro, ri = bench.reference_top1(ref_dir, ref_model, text, offs, np, os.path.join(cache, 'parity_tmp'))
        assert int((rs != 0).sum()) == 0 and len(ro) == batch + 1
        assert int(ro[-1]) > 20 * batch   # ~24 morphemes per sentence: the reference really analysed them
        bad = bench.compare_packed(ho, hi, rs, ro, ri, np)
        total_bad += [(bi, s) for s in bad]
    assert not total_bad, (len(total_bad), total_bad[:8])


@pytest.mark.parametrize('rnn', [None, (128, 3000)])
def test_gpu_sentences_are_routed_to_sweep_variants_one_by_one(gpu_lib, ref_tools, tmp_path, rnn):
    """3 000 ordinary sentences plus a few through surfaces with 100 and 600 dictionary readings in ONE batch: every
    sentence runs the sweep variant of its own widest boundary (three launches), all lattices bit-identical to the
    live reference"""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_cpu_parity as tc
    img, lines, gold_path = tc._mixed_width_workload(ref_tools, str(tmp_path), [5, 6, 1, 5], 3000, rnn=rnn)
    ctx = J.Context(img, lib_path=gpu_lib)
    meta, gold = G.read_gold(gold_path)
    res = ctx.analyze(lines).fetch(full=True)
    cls = ctx.sweep_classes()['sentences']
    # (a few of the random lines run into the wide surfaces by chance)
    assert cls[0] >= 2900 and cls[1] >= 1 and cls[2] >= 2 and sum(cls) == len(lines), cls
    errs = _compare_all(res, gold, meta, len(lines))
    assert not errs, (len(errs), errs[:10])


def test_gpu_normalize_beyond_the_lane_arrays(gpu_lib, ref_tools, tmp_path):
    """the HBM-slice path of the normalize maker with its slot locks contended (16 pool slots, one owner wavefront at a
    time): 200 random sentences + 3 000 with 300 normalised candidates from one start, 1 200 of them with two to four such
    starts in the same wavefront (the case that could hang the per-lane locks of round 3)"""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_cpu_parity as tc
    tc.check_normalize_beyond_the_lane_arrays(gpu_lib, ref_tools, str(tmp_path), 200, copies=600)


def test_gpu_full_beam_beyond_the_lds_staging(gpu_lib, ref_tools, tmp_path):
    """the HBM-slice path of k_sweep_full with its slot locks contended (128 slots): 300 random sentences + 500 with
    boundaries of up to 2 752 candidates"""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_cpu_parity as tc
    tc.check_full_beam_beyond_the_lds_staging(gpu_lib, ref_tools, str(tmp_path), 300, wide_copies=250)


@pytest.mark.parametrize('variant,rnn,beams', [('drop', None, None), ('add', (128, 3000), None), ('drop', None, [5, 0, 0, 0]),
                                               ('add', None, [20, 24, 1, 20]), ('drop', (64, 3000), [32, 32, 1, 32]),
                                               ('cols', None, None), ('cols', (64, 3000), [4, 12, 2, 6])])
def test_gpu_table_driven_kernels_on_a_non_jumandic_spec(gpu_lib, ref_tools, tmp_path, variant, rnn, beams):
    """SURVEY 8 f3 on the MI355X: 1 500 sentences of a 30 k-entry dictionary under a spec whose hash does not match the
    reference's generated code -- table-driven kernels (global-beam and full-beam sweeps) against the reference's dynamic
    feature path, bit for bit"""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_cpu_parity as tc
    tc.check_variant_spec(gpu_lib, ref_tools, str(tmp_path), variant, 1500, beams, rnn, n_entries=30000, exp=20, length=40, seed=37)


def test_gpu_shared_model_contexts(gpu_lib, golden_dir):
    """two contexts on one copy of the model in HBM (jppgpu_ctx_create_shared), different beam configurations, the base
    destroyed first"""
    import test_cpu_parity as tc
    tc.check_shared_model_contexts(gpu_lib, golden_dir)


def test_gpu_one_enqueue_path(gpu_lib, golden_dir):
    """the one-enqueue pipeline (capacity guards on the device), its overflow re-run and jppgpu_ctx_reserve"""
    import test_cpu_parity as tc
    tc.check_one_enqueue_path(gpu_lib, golden_dir)


@pytest.mark.parametrize('beams', [[32, 32, 1, 32], [24, 32, 2, 16]])
def test_gpu_wide_global_beam_candidates(gpu_lib, ref_tools, tmp_path, beams):
    """96 .. 640 global-beam candidates per boundary at beam 32 (keys in registers, prefilter, HBM rounds), 200 sentences"""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_cpu_parity as tc
    tc.check_wide_global_beam_candidates(gpu_lib, ref_tools, str(tmp_path), beams, n_lines=200)


@pytest.mark.gpu
def test_gpu_long_sentence_connectivity(gpu_lib, ref_tools, tmp_path):
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_cpu_parity as tc
    tc.check_long_sentence_connectivity(gpu_lib, ref_tools, str(tmp_path))


@pytest.mark.gpu
def test_gpu_wide_beam_rnn_long_sentences(gpu_lib, ref_tools, tmp_path):
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_cpu_parity as tc
    tc.check_wide_beam_rnn_long_sentences(gpu_lib, ref_tools, str(tmp_path), n_lines=24, length=220)


@pytest.mark.gpu
def test_gpu_length_primitives_spec(gpu_lib, ref_tools, tmp_path):
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_cpu_parity as tc
    tc.check_length_primitives_spec(gpu_lib, ref_tools, str(tmp_path))
