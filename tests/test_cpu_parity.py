"""CPU-side tests (pytest -m "not gpu"): the kernel sources, executed by the
fiber emulator, against golden vectors produced by the real reference; the
libstdc++ nth_element restatement against std::nth_element; C-ABI exports."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import golden_io as G
import jumanpp_amd as J

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_golden(lib, golden_dir, gold_name, image='mini.img', n_lines=None, **cfg):
    ctx = J.Context(os.path.join(golden_dir, image), lib_path=lib, **cfg)
    lines = [l.rstrip('\n') for l in open(os.path.join(golden_dir, 'mini.txt'), encoding='utf-8')][:n_lines]
    meta, gold = G.read_gold(os.path.join(golden_dir, gold_name))
    assert meta['nsent'] == len(lines)
    res = ctx.analyze(lines).fetch(full=True)
    errs = []
    for s in range(len(lines)):
        errs += G.compare_sentence(res, s, gold[s], meta)
    assert not errs, errs[:10]
    return res


def test_emulated_kernels_match_reference_default_config(emu_lib, golden_dir):
    res = _run_golden(emu_lib, golden_dir, 'mini.gold')
    assert int(res.nnodes.sum()) > 2000


def test_emulated_kernels_match_reference_other_beam_config(emu_lib, golden_dir):
    # beam 3, global beam 4, right-check 2, right-beam 3
    _run_golden(emu_lib, golden_dir, 'mini_b3.gold', beam=3, global_beam=4, right_check=2, right_beam=3)


def test_emulated_kernels_match_reference_quickselect_goldens(emu_lib, golden_dir):
    # beam 4 / global beam 12: makeT0Beam takes util::partition (whose result is not always the true top-N,
    # which the goldens pin), perceptron and RNN (remakeEosBeam goes through the same routine)
    _run_golden(emu_lib, golden_dir, 'mini_b4g12.gold', n_lines=8, beam=4, global_beam=12, right_check=1, right_beam=4)
    _run_golden(emu_lib, golden_dir, 'mini_rnn_b4g12.gold', image='mini_rnn.img', n_lines=8, beam=4, global_beam=12,
                right_check=1, right_beam=4)


def test_emulated_kernels_match_reference_with_rnn(emu_lib, golden_dir):
    # perceptron + RNNLM re-ranking: lattice/perceptron cells bit-exact, RNN cells and
    # re-ranked totals within 1e-4 (tests/golden_io.py)
    res = _run_golden(emu_lib, golden_dir, 'mini_rnn.gold', image='mini_rnn.img')
    assert res.cells.shape[2] == 2


def test_emulated_wide_beam_matches_live_reference(emu_lib, golden_dir, ref_tools, tmp_path):
    """beam 20 / global beam 24: more than 16 candidates per node -> libstdc++ introsort order."""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_gpu_parity as tg
    img, lines, gold_path = tg._fresh_workload(ref_tools, str(tmp_path), 2500, 10, 14, 5, length=40,
                                               beams=[20, 24, 1, 20])
    ctx = J.Context(img, lib_path=emu_lib, beam=20, global_beam=24, right_check=1, right_beam=20)
    meta, gold = G.read_gold(gold_path)
    res = ctx.analyze(lines).fetch(full=True)
    errs = []
    for s in range(len(lines)):
        errs += G.compare_sentence(res, s, gold[s], meta)
    assert not errs, errs[:10]


@pytest.mark.parametrize('beams', [[5, 10, 1, 5], [3, 8, 2, 4], [6, 30, 1, 8]])
def test_emulated_partition_branch_of_make_t0_beam(emu_lib, golden_dir, ref_tools, tmp_path, beams):
    """global beam > beam*4/3: makeT0Beam quickselects (util::partition) before sorting
    (score_processor.cc:434-437); exact replay incl. the order among equal scores."""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_gpu_parity as tg
    img, lines, gold_path = tg._fresh_workload(ref_tools, str(tmp_path), 2500, 10, 14, 9, length=40, beams=beams)
    ctx = J.Context(img, lib_path=emu_lib, beam=beams[0], global_beam=beams[1], right_check=beams[2], right_beam=beams[3])
    meta, gold = G.read_gold(gold_path)
    res = ctx.analyze(lines).fetch(full=True)
    errs = []
    for s in range(len(lines)):
        errs += G.compare_sentence(res, s, gold[s], meta)
    assert not errs, errs[:10]


@pytest.mark.parametrize('hidden', [48, 128])
def test_emulated_rnn_staged_and_unstaged_sentences_in_one_batch(emu_lib, ref_tools, tmp_path, hidden):
    """k_rnn_score stages a sentence's beam records in LDS when it has at most 45 codepoints (global beam 6);
    k_rnn_score_long serves the others from the row records: a batch with both kinds, bit-exact RNN scores
    (E = 48 and 128)."""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_gpu_parity as tg
    img, lines, gold_path = tg._fresh_workload(ref_tools, str(tmp_path), 2500, 14, 14, 23, length=40,
                                               rnn=(hidden, 600), join=5)
    assert max(len(l) for l in lines) > 180 and min(len(l) for l in lines) < 60
    ctx = J.Context(img, lib_path=emu_lib)
    meta, gold = G.read_gold(gold_path)
    assert meta['nscorers'] == 2
    res = ctx.analyze(lines).fetch(full=True)
    errs = []
    for s in range(len(lines)):
        errs += G.compare_sentence(res, s, gold[s], meta)
    assert not errs, errs[:10]


def test_emulated_full_beam_matches_live_reference(emu_lib, golden_dir, ref_tools, tmp_path):
    """--global-beam 0: AnalyzerImpl::computeScoresFull (k_sweep_full)."""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_gpu_parity as tg
    img, lines, gold_path = tg._fresh_workload(ref_tools, str(tmp_path), 2500, 12, 14, 6, length=40,
                                               beams=[5, 0, 0, 0])
    ctx = J.Context(img, lib_path=emu_lib, beam=5, global_beam=0, right_check=0, right_beam=0)
    meta, gold = G.read_gold(gold_path)
    assert meta['gbeam'] == 0
    res = ctx.analyze(lines).fetch(full=True)
    errs = []
    for s in range(len(lines)):
        errs += G.compare_sentence(res, s, gold[s], meta)
    assert not errs, errs[:10]


def check_full_beam_beyond_the_lds_staging(lib, ref_tools, tmp, n_lines, wide_copies=1):
    """full-beam scoring where a boundary has more live (left node, slot) candidates than k_sweep_full stages in LDS
    (512): 80 readings of one surface x beam 32 = up to 2 752 candidates -- the kernel takes an HBM slice of its scratch
    pool; the reference has no limit (score_processor.cc:165-191)"""
    import test_gpu_parity as tg
    extra = ''.join('かき,0,0,0,名詞,普通名詞,*,*,かき,よみ%d,かき/よみ%d,代表表記:かき/よみ%d\n' % (i, i, i) for i in range(80))
    img, lines, gold_path = tg._fresh_workload(ref_tools, tmp, 2500, n_lines, 14, 7, length=30, beams=[32, 0, 0, 0], extra_dict=extra,
                                               extra_lines=['かきかきかきの', 'あかきかきい'] * wide_copies)
    ctx = J.Context(img, lib_path=lib, beam=32, global_beam=0, right_check=0, right_beam=0)
    meta, gold = G.read_gold(gold_path)
    res = ctx.analyze(lines).fetch(full=True)
    errs, widest = [], 0
    for s in range(len(lines)):
        errs += G.compare_sentence(res, s, gold[s], meta)
        bb = int(res.bnd_base[s])
        widest = max(widest, int(res.end_count[bb + 2:bb + int(res.ncp[s]) + 3].max()))
    assert widest * 32 > 2048, widest
    assert (res.status == 0).all()
    assert not errs, errs[:10]


def check_normalize_beyond_the_lane_arrays(lib, ref_tools, tmp, n_lines, copies=1):
    """the normalize maker with more results from one start than a lane's arrays hold (160): 300 readings of one surface,
    reached through prolongation / small-kana variants of it -- the traversal is repeated in the lane's HBM slice; the
    reference's lists are unbounded (charlattice.cc:266-353)"""
    import test_gpu_parity as tg
    extra = ''.join('すごい,0,0,0,名詞,普通名詞,*,*,すごい,よみ%d,すごい/よみ%d,代表表記:すごい/よみ%d\n' % (i, i, i) for i in range(300))
    # (the last two: several starts of ONE wavefront beyond the lane arrays at once -- the pool slot is taken per wavefront)
    wide = ['すごーい', 'あすごーーいね', 'すっごーい', 'すごーいすごーい', 'すごーいすっごーいすごーーいすごーい']
    img, lines, gold_path = tg._fresh_workload(ref_tools, tmp, 2500, n_lines, 14, 7, length=30, extra_dict=extra, extra_lines=wide * copies)
    ctx = J.Context(img, lib_path=lib)
    meta, gold = G.read_gold(gold_path)
    res = ctx.analyze(lines).fetch(full=True)
    assert (res.status == 0).all(), [int(x) for x in res.status if x != 0][:5]
    errs = []
    for s in range(len(lines)):
        errs += G.compare_sentence(res, s, gold[s], meta)
    assert not errs, errs[:10]
    for s in range(len(lines) - len(wide), len(lines)):
        bb = int(res.bnd_base[s])
        assert int(res.bnd_count[bb:bb + int(res.ncp[s]) + 3].max()) >= 300


def test_emulated_normalize_beyond_the_lane_arrays(emu_lib, ref_tools, tmp_path):
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    check_normalize_beyond_the_lane_arrays(emu_lib, ref_tools, str(tmp_path), 6)


def test_emulated_full_beam_beyond_the_lds_staging(emu_lib, ref_tools, tmp_path):
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    check_full_beam_beyond_the_lds_staging(emu_lib, ref_tools, str(tmp_path), 10)


def _wide_boundary_workload(ref_tools, tmp, beams, n_homographs=600):
    """a dictionary that puts more right nodes on one boundary than the LDS variants of k_sweep stage
    (kMaxRight = 512): `n_homographs` distinct readings of one surface"""
    mdic = os.path.join(tmp, 'wide.mdic')
    with open(mdic, 'w', encoding='utf-8') as f:
        subprocess.check_call(['python3', os.path.join(ROOT, 'tools', 'gen_dict.py'), '1500', '--seed', '13'], stdout=f)
        for i in range(n_homographs):
            f.write('かき,0,0,0,名詞,普通名詞,*,*,かき,よみ%d,かき/よみ%d,代表表記:かき/よみ%d\n' % (i, i, i))
    subprocess.check_call([os.path.join(ref_tools, 'jpp_jumandic_bootstrap'), mdic, os.path.join(tmp, 'wide.seed')],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([os.path.join(ref_tools, 'ref_dump'), 'mkmodel', os.path.join(tmp, 'wide.seed'),
                           os.path.join(tmp, 'wide.model'), '16', '7', '0.1'])
    subprocess.check_call([os.path.join(ref_tools, 'ref_dump'), 'export', os.path.join(tmp, 'wide.model'),
                           os.path.join(tmp, 'wide.img')], stderr=subprocess.DEVNULL)
    lines = ['かきをかきかきとかきく', 'あかきい', 'かき']
    txt = os.path.join(tmp, 'wide.txt')
    open(txt, 'w', encoding='utf-8').write('\n'.join(lines) + '\n')
    with open(txt, 'rb') as f:
        subprocess.check_call([os.path.join(ref_tools, 'ref_dump'), 'dump', os.path.join(tmp, 'wide.model'),
                               os.path.join(tmp, 'wide.gold')] + [str(x) for x in beams], stdin=f, stderr=subprocess.DEVNULL)
    return os.path.join(tmp, 'wide.img'), lines, os.path.join(tmp, 'wide.gold')


@pytest.mark.parametrize('beams', [[5, 6, 1, 5], [5, 6, 3, 5], [20, 24, 1, 20]])
def test_emulated_lattice_wider_than_the_lds_staging(emu_lib, ref_tools, tmp_path, beams):
    """> 512 nodes starting at one boundary: the reference has no such limit (lattice_builder.cc:70-93);
    k_sweep<*, 0> keeps the per-right-node arrays in HBM instead of failing the sentence."""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    img, lines, gold_path = _wide_boundary_workload(ref_tools, str(tmp_path), beams)
    ctx = J.Context(img, lib_path=emu_lib, beam=beams[0], global_beam=beams[1], right_check=beams[2], right_beam=beams[3])
    meta, gold = G.read_gold(gold_path)
    res = ctx.analyze(lines).fetch(full=True)
    assert list(res.status) == [0, 0, 0]
    assert int(res.bnd_count.max()) > 512
    errs = []
    for s in range(len(lines)):
        errs += G.compare_sentence(res, s, gold[s], meta)
    assert not errs, errs[:10]


def _mixed_width_workload(ref_tools, tmp, beams, n_lines, rnn=None):
    """sentences of all three sweep classes in one batch: ordinary generated lines (at most 64 right nodes per
    boundary), lines through a surface with 100 readings (class 1) and through one with 600 (class 2)"""
    import test_gpu_parity as tg
    mdic = os.path.join(tmp, 'w.mdic')
    extra = ''.join('さけ,0,0,0,名詞,普通名詞,*,*,さけ,よみ%d,さけ/よみ%d,代表表記:さけ/よみ%d\n' % (i, i, i) for i in range(100))
    extra += ''.join('かき,0,0,0,名詞,普通名詞,*,*,かき,よみ%d,かき/よみ%d,代表表記:かき/よみ%d\n' % (i, i, i) for i in range(600))
    img, lines, gold = tg._fresh_workload(ref_tools, tmp, 2500, n_lines, 16, 41, length=30, rnn=rnn, beams=beams,
                                          extra_dict=extra,
                                          extra_lines=['さけをのむ', 'かきをかきかきとかきく', 'あさけとかきい', 'さけさけ'])
    return img, lines, gold


@pytest.mark.parametrize('beams,rnn', [([5, 6, 1, 5], None), ([5, 6, 1, 5], (32, 600)), ([20, 24, 1, 20], None)])
def test_emulated_sentences_are_routed_to_sweep_variants_one_by_one(emu_lib, ref_tools, tmp_path, beams, rnn):
    """one batch holds sentences of every sweep class (widest boundary <= 64 / <= 512 / wider): each runs the variant
    of its class (k_sweep_classify), and the whole lattice of every sentence is the reference's"""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    img, lines, gold_path = _mixed_width_workload(ref_tools, str(tmp_path), beams, 40, rnn=rnn)
    ctx = J.Context(img, lib_path=emu_lib, beam=beams[0], global_beam=beams[1], right_check=beams[2], right_beam=beams[3])
    meta, gold = G.read_gold(gold_path)
    res = ctx.analyze(lines).fetch(full=True)
    cls = ctx.sweep_classes()['sentences']
    assert cls[0] >= 40 and cls[1] >= 1 and cls[2] >= 2 and sum(cls) == len(lines), cls
    errs = []
    for s in range(len(lines)):
        errs += G.compare_sentence(res, s, gold[s], meta)
    assert not errs, errs[:10]


def _variant_spec_workload(ref_tools, tmp, variant, n_lines, beams=None, rnn=None, n_entries=2500, exp=14, length=30, seed=21, gold=True):
    """a model whose spec is NOT the built-in jumandic one (oracle/ref_dump.cc `bootstrapv`: the jumandic spec with its
    last n-gram feature dropped / with a unigram and a bigram added and two bigrams swapped): the reference's spec hash
    no longer matches its generated code and it runs its dynamic feature objects (features_api.cc:20-60)"""
    os.makedirs(tmp, exist_ok=True)
    mdic = os.path.join(tmp, 'v.mdic')
    with open(mdic, 'w', encoding='utf-8') as f:
        subprocess.check_call(['python3', os.path.join(ROOT, 'tools', 'gen_dict.py'), str(n_entries), '--seed', str(seed)], stdout=f)
    if variant == 'cols':
        # four more dictionary columns (CSV columns 13-16) for the spec variant with 12 feature columns per entry row
        import random
        rng = random.Random(seed + 5)
        rows = [l.rstrip('\n') for l in open(mdic, encoding='utf-8')]
        with open(mdic, 'w', encoding='utf-8') as f:
            for l in rows:
                f.write('%s,%s,%s,%s,%s\n' % (l, rng.choice('ABCDE'), rng.choice(['p', 'q', 'r']), rng.choice(['x', 'y']),
                                             rng.choice(['m', 'n', 'o', '*'])))
    rd = os.path.join(ref_tools, 'ref_dump')
    subprocess.check_call([rd, 'bootstrapv', mdic, os.path.join(tmp, 'v.seed'), variant], stderr=subprocess.DEVNULL)
    subprocess.check_call([rd, 'mkmodel', os.path.join(tmp, 'v.seed'), os.path.join(tmp, 'v.model'), str(exp), '3', '0.1'],
                          stderr=subprocess.DEVNULL)
    if rnn is not None:
        hidden, vocab = rnn
        subprocess.check_call(['python3', os.path.join(ROOT, 'tools', 'gen_rnn.py'), mdic, os.path.join(tmp, 'rnn'),
                               '--vocab', str(vocab), '--hidden', str(hidden), '--maxent-size', str(1 << 18),
                               '--seed', str(seed)], stdout=subprocess.DEVNULL)
        os.rename(os.path.join(tmp, 'v.model'), os.path.join(tmp, 'p.model'))
        subprocess.check_call([os.path.join(ref_tools, 'jumanpp_v2_train'), '--model-input=' + os.path.join(tmp, 'p.model'),
                               '--model-output=' + os.path.join(tmp, 'v.model'), '--rnn-model=' + os.path.join(tmp, 'rnn'),
                               '--rnn-fields=surface,pos', '--rnn-nce-bias=5.6', '--rnn-unk-constant=-3.47',
                               '--rnn-unk-length=-2.93', '--feature-weight-perceptron=1', '--feature-weight-rnn=0.0176'],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([rd, 'export', os.path.join(tmp, 'v.model'), os.path.join(tmp, 'v.img')], stderr=subprocess.DEVNULL)
    txt = os.path.join(tmp, 'v.txt')
    with open(txt, 'w', encoding='utf-8') as f:
        subprocess.check_call(['python3', os.path.join(ROOT, 'tools', 'gen_corpus.py'), mdic, str(n_lines), '--seed', str(seed + 1),
                               '--oov', '0.08', '--len', str(length)], stdout=f)
    if gold:
        with open(txt, 'rb') as f:
            subprocess.check_call([rd, 'dump', os.path.join(tmp, 'v.model'), os.path.join(tmp, 'v.gold')] + [str(x) for x in (beams or [])],
                                  stdin=f, stderr=subprocess.DEVNULL)
    lines = [l.rstrip('\n') for l in open(txt, encoding='utf-8')]
    return os.path.join(tmp, 'v.img'), lines, os.path.join(tmp, 'v.gold')


def check_variant_spec(lib, ref_tools, tmp, variant, n_lines, beams, rnn, **kw):
    img, lines, gold_path = _variant_spec_workload(ref_tools, tmp, variant, n_lines, beams=beams, rnn=rnn, **kw)
    b = beams or [5, 6, 1, 5]
    ctx = J.Context(img, lib_path=lib, beam=b[0], global_beam=b[1], right_check=b[2], right_beam=b[3])
    meta, gold = G.read_gold(gold_path)
    assert meta['npat'] > 14   # the reference's dynamic lattice keeps every pattern: it did run its dynamic code
    slots = G.stored_pattern_slots(img)
    res = ctx.analyze(lines).fetch(full=True)
    errs = []
    for s in range(len(lines)):
        errs += G.compare_sentence(res, s, gold[s], meta, pat_slots=slots, verbose=False)
    assert not errs, (len(errs), errs[:8])
    return ctx


@pytest.mark.parametrize('variant,beams,rnn', [('drop', None, None), ('add', None, (32, 600)), ('add', [20, 24, 1, 20], None),
                                               ('drop', [5, 0, 0, 0], None), ('cols', None, None), ('cols', [4, 12, 2, 6], (32, 600)),
                                               ('cols', [5, 0, 0, 0], None)])
def test_emulated_table_driven_kernels_on_a_non_jumandic_spec(emu_lib, ref_tools, tmp_path, variant, beams, rnn):
    """a spec other than the compiled-in tables is analysed by the table-driven kernels (k_t0_dyn,
    k_sweep<.., DYN>, k_sweep_full<DYN> without a global beam) with the summation orders of the reference's DYNAMIC
    feature code: whole lattice bit-identical.  `cols`: 12 feature columns per entry row (rows of 16 in node_entry;
    JPP_MAX_DIC_FIELDS = 16 in the reference)"""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    check_variant_spec(emu_lib, ref_tools, str(tmp_path), variant, 40, beams, rnn)


def test_status_codes_bad_utf8_and_too_long(emu_lib, golden_dir):
    ctx = J.Context(os.path.join(golden_dir, 'mini.img'), lib_path=emu_lib)
    # reference: invalid UTF-8 -> InvalidParameter (characters.cc:267-269);
    # > 4096 bytes -> InvalidParameter (analysis_input.cc:13-17)
    sents = [b'\xe3\x81', b'ok', b'\xff\xfe', ('あ' * 1400).encode('utf-8'), b'']
    res = ctx.analyze(sents).fetch(full=True)
    assert list(res.status) == [2, 0, 2, 1, 0]
    assert res.path_len[1] > 0 and res.path_len[4] == 0


def test_config_validation_mirrors_reference(emu_lib, golden_dir):
    img = os.path.join(golden_dir, 'mini.img')
    with pytest.raises(J.JppGpuError, match='beam size can not be zero'):
        J.Context(img, lib_path=emu_lib, beam=0)
    with pytest.raises(J.JppGpuError, match='right global beam size'):
        J.Context(img, lib_path=emu_lib, right_check=1, right_beam=0)
    with pytest.raises(J.JppGpuError, match='not supported'):
        J.Context(img, lib_path=emu_lib, beam=40, global_beam=40)
    with pytest.raises(J.JppGpuError, match='only with global beam'):
        J.Context(os.path.join(golden_dir, 'mini_rnn.img'), lib_path=emu_lib, global_beam=0)
    with pytest.raises(J.JppGpuError, match='not supported'):
        J.Context(img, lib_path=emu_lib, beam=40, global_beam=0)
    # a beam beyond 32 WITH a global beam is the lattice of beam 32 (a node's beam holds at most global-beam entries)
    lines = [l for l in open(os.path.join(golden_dir, 'mini.txt'), encoding='utf-8').read().split('\n') if l][:12]
    got = []
    for beam in (32, 40, 500):
        ctx = J.Context(img, lib_path=emu_lib, beam=beam, global_beam=6, right_check=1, right_beam=5)
        r = ctx.analyze(lines).fetch(full=True)
        assert r.beam == 32
        got.append((r.path_nodes.tolist(), r.beams.tobytes(), r.cells.tobytes()))
    assert got[0] == got[1] == got[2]


def check_top1_fetch_equals_basic_fetch(ctx, lines):
    """JPPGPU_FETCH_TOP1 (device-compacted path nodes) against the node table of the basic fetch"""
    import numpy as np
    r = ctx.analyze(lines)
    b = r.fetch()
    status, nbase, plen = b.status.copy(), b.node_base.copy(), b.path_len.copy()
    pnodes, nodes, unk = b.path_nodes.copy(), b.nodes.copy(), b.unk.copy()
    t = r.fetch(top1=True)
    assert t.n == len(lines) and np.array_equal(t.status, status) and np.array_equal(t.path_len, plen)
    assert np.array_equal(t.nnodes, plen)
    assert int(t.view.total_nodes) == int(plen.sum())
    for s in range(len(lines)):
        o, n0 = int(t.node_base[s]), int(nbase[s])
        L = int(plen[s])
        assert np.array_equal(t.path_nodes[o:o + L], np.arange(L))
        idx = n0 + pnodes[n0:n0 + L].astype(np.int64)
        assert np.array_equal(t.nodes[o:o + L], nodes[idx]), s
        assert np.array_equal(t.unk[o:o + L], unk[idx]), s
        if L:
            assert int(t.nodes[o]['eptr']) == -0x7ffffffe  # EOS first
    return int(plen.sum())


def check_nbest_fetch_equals_full_lattice(ctx, lines, n_best):
    """jppgpu_result_fetch_nbest (device-gathered paths) against a walk through the full lattice arrays"""
    import numpy as np
    r = ctx.analyze(lines)
    f = r.fetch(full=True)
    status, nbase, nnodes = f.status.copy(), f.node_base.copy(), f.nnodes.copy()
    beams, cells, nodes, unk = f.beams.copy(), f.cells.copy(), f.nodes.copy(), f.unk.copy()
    eos, first, items, nn = r.fetch_nbest(n_best)
    assert np.array_equal(nn, nnodes)
    walked = 0
    for s in range(len(lines)):
        for i in range(n_best):
            lo, hi = int(first[s * n_best + i]), int(first[s * n_best + i + 1])
            if status[s] != 0 or nnodes[s] <= 3:
                assert lo == hi and eos[s, i]['left'] == 0xffff
                continue
            nb, N = int(nbase[s]), int(nnodes[s])
            el = beams[nb + N - 1][i] if i < beams.shape[1] else None
            if el is None or (el['left'] == 0xffff and el['beam'] == 0xffff):
                assert lo == hi and eos[s, i]['left'] == 0xffff and eos[s, i]['beam'] == 0xffff
                continue
            assert eos[s, i] == el
            node, slot, k = int(el['prev_node']), int(el['beam']), lo
            while node >= 2 and node != 0xffffffff:
                c = beams[nb + node][slot]
                it = items[k]
                assert k < hi and it['node'] == node and it['slot'] == slot and it['beam'] == c
                assert it['info'] == nodes[nb + node] and it['unk'] == unk[nb + node]
                assert np.array_equal(it['cells'][:cells.shape[2]].view('<u4'), cells[nb + node][int(c['pad'])].view('<u4'))
                node, slot, k = int(c['prev_node']), int(c['beam']), k + 1
            assert k == hi
            walked += hi - lo
    return walked


def test_nbest_fetch_equals_full_lattice(emu_lib, golden_dir):
    lines = [l.rstrip('\n') for l in open(os.path.join(golden_dir, 'mini.txt'), encoding='utf-8')]
    lines = lines[:6] + ['', b'\xff bad'] + lines[6:14]
    for image, n_best in (('mini.img', 5), ('mini_rnn.img', 3), ('mini.img', 9)):   # 9 > beam: the extra paths are empty
        ctx = J.Context(os.path.join(golden_dir, image), lib_path=emu_lib)
        assert check_nbest_fetch_equals_full_lattice(ctx, lines, n_best) > 100


def test_top1_fetch_equals_basic_fetch(emu_lib, golden_dir):
    ctx = J.Context(os.path.join(golden_dir, 'mini.img'), lib_path=emu_lib)
    lines = [l.rstrip('\n') for l in open(os.path.join(golden_dir, 'mini.txt'), encoding='utf-8')]
    lines = lines[:10] + [b'', b'\xff\xfe bad utf-8'] + lines[10:]
    assert check_top1_fetch_equals_basic_fetch(ctx, lines) > 100
    assert check_top1_fetch_equals_basic_fetch(ctx, []) == 0


def test_result_invalidated_by_next_batch(emu_lib, golden_dir):
    ctx = J.Context(os.path.join(golden_dir, 'mini.img'), lib_path=emu_lib)
    r1 = ctx.analyze(['あいう'])
    ctx.analyze(['かきく'])
    with pytest.raises(J.JppGpuError, match='invalidated'):
        r1.fetch()


def test_oracle_rnn_arithmetic_is_pinned_by_the_legacy_evaluator(ref_tools, tmp_path):
    """the oracle build's RNN (src/rnn/mikolov_rnn.cc over the Eigen stand-in) against the
    reference's in-tree legacy faster-rnnlm evaluator (src/rnn/legacy/rnnlmlib_static.cpp) on a synthetic
    model: hidden states and log10 scores within 1e-4 over seeded random word chains (oracle/legacy_check.cc)."""
    if ref_tools is None or not os.path.exists(os.path.join(ref_tools, 'legacy_check')):
        pytest.skip('oracle/_ref/legacy_check not built')
    mdic = tmp_path / 'd.mdic'
    with open(mdic, 'w', encoding='utf-8') as f:
        subprocess.check_call(['python3', os.path.join(ROOT, 'tools', 'gen_dict.py'), '3000', '--seed', '5'], stdout=f)
    for hidden, order in ((128, 3), (48, 2)):
        rnn = str(tmp_path / ('rnn%d' % hidden))
        subprocess.check_call(['python3', os.path.join(ROOT, 'tools', 'gen_rnn.py'), str(mdic), rnn, '--vocab', '2000',
                               '--hidden', str(hidden), '--maxent-order', str(order), '--maxent-size', str(1 << 16),
                               '--seed', '9'], stdout=subprocess.DEVNULL)
        out = subprocess.run([os.path.join(ref_tools, 'legacy_check'), rnn, '150', '8', '3'], capture_output=True)
        assert out.returncode == 0, (out.stdout[-300:], out.stderr[-500:])
        r = __import__('json').loads(out.stdout.decode().strip().splitlines()[-1])
        assert r['steps'] == 1200 and r['mismatches'] == 0 and r['max_abs_context_diff'] < 1e-5


def test_device_expf_is_the_host_libms(tmp_path):
    """expf_libm (jpp_device.h) restates glibc's expf with the contractions of its x86-64 FMA build: it must
    return this host's expf() bit for bit (the RNN sigmoid of the reference goes through std::exp)."""
    src = tmp_path / 'e.cc'
    src.write_text(r'''
#include <cmath>
#include <cstdio>
#include <cstring>
#include "jpp_device.h"
int main() {
  u64 tab[jpp::kExp2fN];
  for (int i = 0; i < jpp::kExp2fN; ++i) tab[i] = jpp::exp2f_tab(i);
  unsigned long long seed = 88172645463325252ULL;
  long bad = 0, n = 30000000;
  for (long i = 0; i < n; ++i) {
    seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17;
    float x;
    if (i % 4 == 0) { unsigned b = (unsigned)seed; memcpy(&x, &b, 4); if (!(x == x)) continue; }
    else x = (float)(((double)(seed >> 11) / 9007199254740992.0 - 0.5) * ((i % 4 == 1) ? 60.0 : (i % 4 == 2) ? 8.0 : 220.0));
    volatile float xv = x;
    float e = expf(xv), m = jpp::expf_libm(x, tab);
    if (memcmp(&e, &m, 4) != 0) { if (bad < 5) printf("x=%a libm=%a mine=%a\n", x, e, m); ++bad; }
  }
  // the sigmoid of MikolovRnnImplParallel::computeNewContext on a grid
  for (int i = -4000; i <= 4000; ++i) {
    volatile float x = (float)i * 0.01f;
    float e = 1.0f / (1.0f + expf(-x)), m = jpp::sigmoid_ref(x, tab);
    if (memcmp(&e, &m, 4) != 0) ++bad;
  }
  printf("%ld %ld\n", n, bad);
  return bad != 0;
}
''')
    exe = tmp_path / 'e'
    subprocess.check_call(['g++', '-std=c++17', '-O2', '-ffp-contract=off', '-DJPP_EMU', '-I', os.path.join(ROOT, 'tests', 'emu'),
                           '-I', os.path.join(ROOT, 'jumanpp_amd', 'csrc'), str(src), '-o', str(exe), '-lm'])
    out = subprocess.check_output([str(exe)]).decode().split()
    assert int(out[-1]) == 0, out


def test_std_sort_is_partitioning_plus_a_stable_rank(tmp_path):
    """k_sweep<32,*> 5c with ties: libstdc++'s std::sort = the introsort partitions followed by a final
    insertion pass that is a stable sort of the partitioned array.  The kernel replays only the partitions
    serially (std_sort_partition_only) and ranks in parallel; here that is checked against std::sort itself
    on tie-heavy inputs (and the heap-sort fallback is reported, not mis-sorted)."""
    src = tmp_path / 'ps.cc'
    src.write_text(r'''
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "jpp_select.h"
int main() {
  std::mt19937 rng(4242);
  long bad = 0, cases = 0, fallback = 0;
  for (int n = 1; n <= 64; ++n) {
    for (int rep = 0; rep < 400; ++rep) {
      int levels = 1 + rng() % (rep % 3 == 0 ? 3 : rep % 3 == 1 ? 8 : 40);
      std::vector<float> sc(n);
      for (auto& x : sc) x = (float)(rng() % levels) * 0.5f - 1.f;
      std::vector<unsigned> ref(n);
      for (int i = 0; i < n; ++i) ref[i] = i;
      std::sort(ref.begin(), ref.end(), [&](unsigned a, unsigned b) { return sc[a] > sc[b]; });
      std::vector<u64> keys(n);
      for (int i = 0; i < n; ++i) { u32 bits; memcpy(&bits, &sc[i], 4); keys[i] = ((u64)bits << 32) | (u32)i; }
      auto comp = [](u64 a, u64 b) { u32 xa = (u32)(a >> 32), xb = (u32)(b >> 32); float fa, fb; memcpy(&fa, &xa, 4); memcpy(&fb, &xb, 4); return fa > fb; };
      ++cases;
      if (n <= 32) {  // the stackless form the kernel uses must leave the same array
        std::vector<u64> k2(keys);
        bool ok2 = jpp::std_sort_partition_only_le32(k2.data(), k2.data() + n, comp);
        std::vector<u64> k1(keys);
        bool ok1 = jpp::std_sort_partition_only(k1.data(), k1.data() + n, comp);
        if (ok1 != ok2 || (ok1 && k1 != k2)) ++bad;
      }
      if (!jpp::std_sort_partition_only(keys.data(), keys.data() + n, comp)) { ++fallback; continue; }
      std::vector<unsigned> mine(n);
      for (int i = 0; i < n; ++i) {
        u32 vb = (u32)(keys[i] >> 32); float mv; memcpy(&mv, &vb, 4);
        int pos = 0;
        for (int p = 0; p < n; ++p) { u32 ob = (u32)(keys[p] >> 32); float ov; memcpy(&ov, &ob, 4); if (ov > mv || (ov == mv && p < i)) ++pos; }
        mine[pos] = (unsigned)(keys[i] & 0xff);
      }
      if (mine != ref) ++bad;
    }
  }
  printf("%ld %ld %ld\n", cases, bad, fallback);
  return bad != 0;
}
''')
    exe = tmp_path / 'ps'
    subprocess.check_call(['g++', '-std=c++17', '-O1', '-DJPP_EMU', '-I', os.path.join(ROOT, 'tests', 'emu'),
                           '-I', os.path.join(ROOT, 'jumanpp_amd', 'csrc'), str(src), '-o', str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    assert int(out[0]) > 20000 and int(out[1]) == 0 and int(out[2]) < int(out[0]) // 20, out


def test_hoare_partition_in_closed_form_is_the_loop():
    """k_sweep<32,*> pass B1 replays libstdc++'s __unguarded_partition_pivot without its loop: the k-th stop of the upward
    scan (element not greater than the pivot, from f + 1 up) is exchanged with the k-th stop of the downward scan (not
    less, from l - 1 down to f) as long as they have not crossed, and the function returns a_1 (no exchange) or
    min(a_{K+1}, b_K).  A plain restatement of that rule against a plain restatement of the loop (jpp_select.h:
    sel_move_median_to_first + sel_unguarded_partition), on arrays with heavy ties and on sub-ranges; the kernel code itself
    is checked by the reference on every test with a wide beam."""
    import random

    def loop(v, f, l):
        v = v[:]

        def swap(i, j):
            v[i], v[j] = v[j], v[i]
        a, mid, c = f + 1, f + (l - f) // 2, l - 1
        if v[a][0] > v[mid][0]:
            swap(f, mid) if v[mid][0] > v[c][0] else swap(f, c) if v[a][0] > v[c][0] else swap(f, a)
        elif v[a][0] > v[c][0]:
            swap(f, a)
        elif v[mid][0] > v[c][0]:
            swap(f, c)
        else:
            swap(f, mid)
        first, last = f + 1, l
        while True:
            while v[first][0] > v[f][0]:
                first += 1
            last -= 1
            while v[f][0] > v[last][0]:
                last -= 1
            if not first < last:
                return v, first
            swap(first, last)
            first += 1

    def closed(v, f, l):
        n = len(v)
        a, mid, c = f + 1, f + (l - f) // 2, l - 1
        va, vb, vc, vf = v[a][0], v[mid][0], v[c][0], v[f][0]
        if va > vb:
            m = mid if vb > vc else c if va > vc else a
        else:
            m = a if va > vc else c if vb > vc else mid
        pv = va if m == a else vb if m == mid else vc
        my = [pv if h == f else vf if h == m else v[h][0] for h in range(n)]
        ups = [h for h in range(f + 1, l) if not my[h] > pv]          # a_1 < a_2 < ...
        downs = [h for h in range(l - 1, f - 1, -1) if not pv > my[h]]  # b_1 > b_2 > ...
        src = list(range(n))
        K = 0
        for k in range(min(len(ups), len(downs))):
            if ups[k] < downs[k]:
                src[ups[k]], src[downs[k]] = downs[k], ups[k]
                K += 1
        # (a prefix: once a pair has crossed, every later one has)
        assert all(not ups[k] < downs[k] for k in range(K, min(len(ups), len(downs))))
        cut = (ups[0] if ups else l) if K == 0 else min(ups[K] if K < len(ups) else 64, downs[K - 1])
        src = [m if q == f else f if q == m else q for q in src]   # the pivot's exchange came first
        return [v[q] for q in src], cut

    rnd = random.Random(4)
    for _ in range(20000):
        n = rnd.randint(17, 32)
        f, l = 0, n
        if rnd.random() < 0.4:
            f = rnd.randint(0, n - 17)
            l = rnd.randint(f + 17, n)
        top = rnd.choice([1, 2, 3, 5, 50])
        v = [(rnd.randint(0, top), i) for i in range(n)]
        assert loop(v, f, l) == closed(v, f, l), (v, f, l)


def test_make_t0_beam_is_a_rank_when_totals_are_distinct(tmp_path):
    """k_sweep<32,512> 5c / remakeEosBeam fast path: with pairwise distinct totals, util::partition
    (beyond beam*4/3) followed by std::sort (introsort beyond 16) yields the first `beam` entries of the
    unique descending order, i.e. the parallel rank.  Checked against the step-by-step replay."""
    src = tmp_path / 'rank.cc'
    src.write_text(r'''
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>
#include "jpp_select.h"
int main() {
  std::mt19937 rng(777);
  long bad = 0, cases = 0;
  for (int beam = 1; beam <= 32; ++beam) {
    const int partB = beam * 4 / 3;
    for (int cnt = 1; cnt <= 32; ++cnt) {
      for (int rep = 0; rep < 40; ++rep) {
        std::vector<float> tot(cnt);
        for (int i = 0; i < cnt; ++i) tot[i] = (float)i * 0.37f - 3.f;   // distinct
        std::shuffle(tot.begin(), tot.end(), rng);
        u8 idx[32];
        for (int z = 0; z < cnt; ++z) idx[z] = (u8)z;
        const float* tr = tot.data();
        auto comp = [tr](u8 a, u8 b) { return tr[a] > tr[b]; };
        u8* itr = idx + cnt;
        if (cnt > partB) itr = jpp::jpp_partition(idx, itr, comp, (long)beam, (long)partB);
        jpp::std_sort(idx, itr, comp);
        const int have = (int)(itr - idx);
        // parallel rank
        std::vector<int> slot(beam, -1);
        for (int i = 0; i < cnt; ++i) {
          int rank = 0;
          for (int j = 0; j < cnt; ++j) if (tot[j] > tot[i] || (tot[j] == tot[i] && j < i)) ++rank;
          if (rank < beam) slot[rank] = i;
        }
        ++cases;
        for (int z = 0; z < beam; ++z) {
          int want = z < have ? (int)idx[z] : -1;
          if (slot[z] != want) { ++bad; break; }
        }
      }
    }
  }
  printf("%ld %ld\n", cases, bad);
  return bad != 0;
}
''')
    exe = tmp_path / 'rank'
    subprocess.check_call(['g++', '-std=c++17', '-O1', '-DJPP_EMU', '-I', os.path.join(ROOT, 'tests', 'emu'),
                           '-I', os.path.join(ROOT, 'jumanpp_amd', 'csrc'), str(src), '-o', str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    assert int(out[0]) > 30000 and int(out[1]) == 0


def test_nth_element_restatement_matches_libstdcxx(tmp_path):
    src = tmp_path / 'sel.cc'
    src.write_text(r'''
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>
#include "jpp_select.h"
int main() {
  std::mt19937 rng(12345);
  long bad = 0, cases = 0;
  for (int n = 1; n <= 200; ++n) {
    for (int rep = 0; rep < 60; ++rep) {
      int levels = 1 + rng() % 6;  // few distinct scores => many ties
      std::vector<float> sc(n);
      for (auto& x : sc) x = (float)(rng() % levels) * 0.25f - (rep % 3 == 0 ? 0.f : (float)(rng() % 1000) * (rep % 2 ? 0.f : 1e-3f));
      for (int k = 0; k <= n; ++k) {
        if (rep % 7 && k != (int)(rng() % (n + 1))) continue;
        std::vector<u16> a(n), b(n);
        for (int i = 0; i < n; ++i) a[i] = b[i] = (u16)i;
        auto cmp = [&](u16 x, u16 y) { return sc[x] > sc[y]; };
        std::nth_element(a.begin(), a.begin() + k, a.end(), cmp);
        jpp::ScoreGreater g{sc.data()};
        jpp::nth_element_u16(b.data(), b.data() + k, b.data() + n, g);
        ++cases;
        if (a != b) ++bad;
      }
    }
  }
  // std::sort restatement on structs with a partial key (ties keep libstdc++'s order)
  struct It { int key; int tag; bool operator==(const It& o) const { return key == o.key && tag == o.tag; } };
  for (int n = 1; n <= 300; ++n) {
    for (int rep = 0; rep < 40; ++rep) {
      int levels = 1 + rng() % (rep % 2 ? 5 : 50);
      std::vector<It> a(n), b;
      for (int i = 0; i < n; ++i) a[i] = It{(int)(rng() % levels), i};
      b = a;
      auto cmp = [](const It& x, const It& y) { return x.key < y.key; };
      std::sort(a.begin(), a.end(), cmp);
      jpp::std_sort(b.data(), b.data() + n, cmp);
      ++cases;
      if (!(a == b)) ++bad;
    }
  }
  printf("%ld %ld\n", cases, bad);
  return bad != 0;
}
''')
    exe = tmp_path / 'sel'
    subprocess.check_call(['g++', '-std=c++17', '-O1', '-DJPP_EMU', '-I', os.path.join(ROOT, 'tests', 'emu'),
                           '-I', os.path.join(ROOT, 'jumanpp_amd', 'csrc'), str(src), '-o', str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    assert int(out[0]) > 5000 and int(out[1]) == 0


def test_c_abi_exports_every_declared_symbol(gpu_lib):
    hdr = open(os.path.join(ROOT, 'include', 'jppgpu.h')).read()
    names = set(re.findall(r'\b(jppgpu_[a-z_]+)\s*\(', hdr))
    assert {'jppgpu_ctx_create', 'jppgpu_analyze_batch', 'jppgpu_analyze_batch_device', 'jppgpu_result_fetch',
            'jppgpu_result_release', 'jppgpu_ctx_destroy', 'jppgpu_last_error'} <= names
    lib = ctypes.CDLL(gpu_lib)  # loads without a GPU; only symbol presence is checked here
    for n in names:
        assert hasattr(lib, n), n


def test_product_library_fails_loudly_without_gpu(gpu_lib, golden_dir):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(J.JppGpuError, match='no HIP device'):
        J.Context(os.path.join(golden_dir, 'mini.img'), lib_path=gpu_lib)


def test_oracle_restatement_is_pinned_by_reference_goldens(golden_dir):
    """oracle/jpp_oracle.cc (plain C++ restatement of the scoring core, using libstdc++'s own
    nth_element/sort) must reproduce the reference's golden vectors: bit-for-bit on the perceptron
    path, and with the RNN re-ranker the same RNN lattice, score cells and totals within 1e-4."""
    import __graft_entry__ as ge
    ge.build_oracle_port()
    exe = os.path.join(ROOT, 'oracle', '_port', 'jpp_oracle')
    for image, gold in (('mini.img', 'mini.gold'), ('mini.img', 'mini_b3.gold'), ('mini_rnn.img', 'mini_rnn.gold'),
                        ('mini.img', 'mini_b4g12.gold'), ('mini_rnn.img', 'mini_rnn_b4g12.gold')):
        # (the quickselect goldens hold the first 8 sentences; the checker stops at the shorter of corpus and golden)
        out = subprocess.run([exe, 'check', os.path.join(golden_dir, image), os.path.join(golden_dir, 'mini.txt'),
                              os.path.join(golden_dir, gold)], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        assert ' 0 mismatches' in out.stdout


def _fuzz_lines(n, seed):
    """short lines over a deliberately nasty alphabet: every character class of
    util/characters.cc (digits and separators, alphabets, kana with prolong marks and small
    kana, kanji numerals, symbols, brackets, spaces, 4-byte code points, Greek/Cyrillic...)"""
    import random
    rnd = random.Random(seed)
    pools = [
        'あいうえおかきくけこさしすせそたちつてとなにぬねのはひふへほまみむめもやゆよらりるれろわをんがぎぐげござじずぜぞだぢづでどばびぶべぼぱぴぷぺぽ',
        'ぁぃぅぇぉっゃゅょゎー〜～ｰ',
        'アイウエオカキクケコサシスセソタチツテトナニヌネノハヒフヘホマミムメモヤユヨラリルレロワヲンヴヵヶッャュョァィゥェォ',
        '一二三四五六七八九十百千万億兆〇零何数幾', '0123456789０１２３４５６７８９', '.,．，・／/:：', 'abcXYZａｂｃＸＹＺ',
        '漢字語彙形態素解析東京都大阪市食べる走った美しい', '　 \t', '！？!?、。「」（）()［］[]【】『』〈〉《》', '＋－＝×÷％＃＄＆＠※○●◎△▽☆★♪→←↑↓〒',
        'αβγΩабвгД', '𠮷𩸽😀🎉', 'ｱｲｳｴｵｶﾞﾊﾟ', 'ゝゞヽヾ々〆',
    ]
    lines = []
    for _ in range(n):
        k = rnd.randint(1, 28)
        s = []
        while len(s) < k:
            pool = rnd.choice(pools)
            run = rnd.randint(1, 4)
            for _ in range(run):
                s.append(rnd.choice(pool))
            if rnd.random() < 0.15:  # onomatopoeia-like repetitions
                a, b = rnd.choice(pools[0]), rnd.choice(pools[0])
                s.extend([a, b, a, b])
        lines.append(''.join(s[:40]))
    return lines


def test_emulated_kernels_on_fuzzed_character_classes(emu_lib, ref_tools, tmp_path):
    """300 fuzzed lines: node sets of every UNK maker, patterns, scores, beams and paths vs the live reference"""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_gpu_parity as tg
    tmp = str(tmp_path)
    img, _, _ = tg._fresh_workload(ref_tools, tmp, 2500, 2, 14, 61, length=10)
    lines = _fuzz_lines(300, 5)
    txt = os.path.join(tmp, 'fuzz.txt')
    open(txt, 'w', encoding='utf-8').write('\n'.join(lines) + '\n')
    with open(txt, 'rb') as f:
        subprocess.check_call([os.path.join(ref_tools, 'ref_dump'), 'dump', os.path.join(tmp, 'w.model'),
                               os.path.join(tmp, 'fuzz.gold')], stdin=f, stderr=subprocess.DEVNULL)
    meta, gold = G.read_gold(os.path.join(tmp, 'fuzz.gold'))
    ctx = J.Context(img, lib_path=emu_lib)
    res = ctx.analyze(lines).fetch(full=True)
    errs = []
    for s in range(len(lines)):
        errs += G.compare_sentence(res, s, gold[s], meta)
    assert not errs, (len(errs), errs[:10])


def test_emulated_maximum_size_sentences(emu_lib, ref_tools, tmp_path):
    """1 300 codepoints (≈3.9 KB, just under maxInputBytes = 4096): every LDS-staged path falls back to
    its long-sentence branch (u64 reach mask, LDS ends lists, RNN bookkeeping) and must still match."""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import test_gpu_parity as tg
    img, lines, gold_path = tg._fresh_workload(ref_tools, str(tmp_path), 2500, 2, 14, 5, length=1300)
    assert all(3500 < len(l.encode('utf-8')) <= 4096 for l in lines)
    ctx = J.Context(img, lib_path=emu_lib)
    meta, gold = G.read_gold(gold_path)
    res = ctx.analyze(lines).fetch(full=True)
    assert list(res.status) == [0, 0]
    errs = []
    for s in range(len(lines)):
        errs += G.compare_sentence(res, s, gold[s], meta)
    assert not errs, errs[:10]


def check_wide_beam_rnn_long_sentences(lib, ref_tools, tmp, n_lines=3, length=110):
    """the configs[4] shape with the RNN: beam = global beam = 32 on sentences far beyond the LDS staging of
    k_rnn_score (k_rnn_prep: 32 paths side by side, several histories reaching the same word at one boundary;
    k_rnn_score_long)"""
    import test_gpu_parity as tg
    beams = [32, 32, 1, 32]
    img, lines, gold_path = tg._fresh_workload(ref_tools, tmp, 2500, n_lines, 14, 29, length=length, rnn=(32, 600), beams=beams)
    ctx = J.Context(img, lib_path=lib, beam=32, global_beam=32, right_check=1, right_beam=32)
    meta, gold = G.read_gold(gold_path)
    assert meta['nscorers'] == 2
    res = ctx.analyze(lines).fetch(full=True)
    errs = []
    for s in range(len(lines)):
        errs += G.compare_sentence(res, s, gold[s], meta)
    assert not errs, errs[:10]


def test_emulated_wide_beam_rnn_long_sentences(emu_lib, ref_tools, tmp_path):
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    check_wide_beam_rnn_long_sentences(emu_lib, ref_tools, str(tmp_path))


def test_emulated_rnn_lattice_path_by_path_replay(golden_dir, ref_tools, tmp_path):
    """k_rnn_prep builds a boundary's rnn nodes for all paths at once and keeps the reference's order of events -- one
    distinct connection after the other, the boundary's nodes searched for each -- for boundaries where equal prefix
    hashes carry different (id, length), which no corpus produces.  -DJPP_RNN_PREP_SERIAL sends every boundary that
    way: the same goldens, default and wide beams."""
    import __graft_entry__ as ge
    lib = ge.build_emu_variant('serial', ['-DJPP_RNN_PREP_SERIAL'])
    _run_golden(lib, golden_dir, 'mini_rnn.gold', image='mini_rnn.img')
    _run_golden(lib, golden_dir, 'mini_rnn_b4g12.gold', image='mini_rnn.img', n_lines=8, beam=4, global_beam=12,
                right_check=1, right_beam=4)
    if ref_tools is not None:
        check_wide_beam_rnn_long_sentences(lib, ref_tools, str(tmp_path), n_lines=2, length=80)


@pytest.mark.parametrize('beams', [[3, 7, 1, 3], [8, 16, 2, 8], [16, 31, 1, 16]])
def test_emulated_rnn_lattice_beam_shapes(emu_lib, ref_tools, tmp_path, beams):
    """k_rnn_prep packs 64 / G boundaries into a round (G = global beam): global beams that do not divide 64, leave idle
    lanes in a round or fill half a wavefront, on sentences long enough for several histories to reach the same word;
    the all-paths-at-once construction and the path-by-path replay against the reference's lattice, cells and EOS beam."""
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    import __graft_entry__ as ge
    import test_gpu_parity as tg
    img, lines, gold_path = tg._fresh_workload(ref_tools, str(tmp_path), 2500, 6, 14, 37 + beams[1], length=70, rnn=(32, 600),
                                               beams=beams)
    meta, gold = G.read_gold(gold_path)
    assert meta['nscorers'] == 2
    for lib in (emu_lib, ge.build_emu_variant('serial', ['-DJPP_RNN_PREP_SERIAL'])):
        ctx = J.Context(img, lib_path=lib, beam=beams[0], global_beam=beams[1], right_check=beams[2], right_beam=beams[3])
        res = ctx.analyze(lines).fetch(full=True)
        errs = []
        for s in range(len(lines)):
            errs += G.compare_sentence(res, s, gold[s], meta)
        assert not errs, (lib, errs[:10])


def check_long_sentence_connectivity(lib, ref_tools, tmp):
    """sentences of more than 63 codepoints through k_connect's sliding window: plain ones, ones with a node longer
    than the window (a run of 70 / 150 digits: the sequential pass), ones whose stretches of unknown characters only
    connect through the stage-2 makers, and one of each kind next to the 64-codepoint limit"""
    import test_gpu_parity as tg
    img, base, _ = tg._fresh_workload(ref_tools, tmp, 2500, 4, 14, 77, length=90)
    fuzz = _fuzz_lines(40, 9)
    lines = list(base)
    lines.append(base[0][:30] + '1' * 70 + base[1][:20])
    lines.append('9' * 150 + base[2][:10])
    lines.append(base[3][:20] + 'ａｂｃ' * 25 + base[0][:20])
    lines.append(''.join(fuzz[:4]))
    lines.append(''.join(fuzz[4:9]))
    lines.append(''.join(fuzz[9:12])[:64])
    lines.append(''.join(fuzz[12:15])[:63])
    lines.append(''.join(fuzz[15:18])[:65])
    lines.append(base[1][:60] + '𠮷😀' * 3 + 'ゝゞ々' + base[2][:30])
    txt = os.path.join(tmp, 'long.txt')
    open(txt, 'w', encoding='utf-8').write('\n'.join(lines) + '\n')
    with open(txt, 'rb') as f:
        subprocess.check_call([os.path.join(ref_tools, 'ref_dump'), 'dump', os.path.join(tmp, 'w.model'),
                               os.path.join(tmp, 'long.gold')], stdin=f, stderr=subprocess.DEVNULL)
    meta, gold = G.read_gold(os.path.join(tmp, 'long.gold'))
    ctx = J.Context(img, lib_path=lib)
    res = ctx.analyze(lines).fetch(full=True)
    errs = []
    for s in range(len(lines)):
        errs += G.compare_sentence(res, s, gold[s], meta)
    assert not errs, (len(errs), errs[:10])
    assert max(len(l) for l in lines) > 150


def test_emulated_long_sentence_connectivity(emu_lib, ref_tools, tmp_path):
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    check_long_sentence_connectivity(emu_lib, ref_tools, str(tmp_path))


# ---- training hook: top-1 n-gram feature values, weight upload ----

def check_top1_ngrams_against_reference(lib, ref_tools, golden_dir, tmp_path, workload=None, min_checked=300):
    """jppgpu_result_fetch_top1_ngrams against NgramFeaturesComputer::calculateNgramFeatures on the reference's own
    lattice (oracle/ref_dump.cc `ngrams`, what LossCalculator::addTopNgrams reads): every u32, bit for bit.
    workload = (model.jppmdl, image, text file) instead of the golden mini model."""
    import struct
    out = str(tmp_path / 'ng.bin')
    model, img, txt = workload or (os.path.join(golden_dir, 'mini.jppmdl'), os.path.join(golden_dir, 'mini.img'),
                                   os.path.join(golden_dir, 'mini.txt'))
    with open(txt, 'rb') as f:
        subprocess.check_call([os.path.join(ref_tools, 'ref_dump'), 'ngrams', model, out], stdin=f, stderr=subprocess.DEVNULL)
    raw = open(out, 'rb').read()
    magic, ns, nf = struct.unpack_from('<III', raw, 0)
    assert magic == 0x3152474e and nf == 73
    lines = [l.rstrip('\n') for l in open(txt, encoding='utf-8')]
    assert ns == len(lines)
    ctx = J.Context(img, lib_path=lib)
    r = ctx.analyze(lines)
    full = r.fetch(full=True)
    first, nodes, feats = r.fetch_top1_ngrams()
    assert feats.shape[1] == nf and len(first) == ns + 1
    pos = 12
    checked = 0
    for s in range(ns):
        st, np_ = struct.unpack_from('<II', raw, pos)
        pos += 8
        assert (st == 0) == (int(full.status[s]) == 0)
        lo, hi = int(first[s]), int(first[s + 1])
        assert hi - lo == np_ == (int(full.path_len[s]) if st == 0 else 0)
        bb = int(full.bnd_base[s])
        for j in range(np_):
            b, rpos = struct.unpack_from('<HH', raw, pos)
            pos += 4
            ref = np.frombuffer(raw, dtype='<u4', count=nf, offset=pos)
            pos += 4 * nf
            node = int(nodes[lo + j])
            assert node == int(full.bnd_first[bb + b]) + rpos, (s, j)
            assert np.array_equal(feats[lo + j], ref), (s, j, np.nonzero(feats[lo + j] != ref)[0][:8])
            checked += 1
    assert checked > min_checked
    r.release()


def test_emulated_top1_ngram_features_match_the_reference_trainer(emu_lib, ref_tools, golden_dir, tmp_path):
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    check_top1_ngrams_against_reference(emu_lib, ref_tools, golden_dir, tmp_path)


def check_set_weights(lib, golden_dir):
    """jppgpu_ctx_set_weights: doubling every weight doubles every perceptron score exactly (powers of two) and
    keeps every decision; the original table restores the original results bit for bit."""
    lines = [l.rstrip('\n') for l in open(os.path.join(golden_dir, 'mini.txt'), encoding='utf-8')]
    ctx = J.Context(os.path.join(golden_dir, 'mini.img'), lib_path=lib)
    a = ctx.analyze(lines).fetch(full=True)
    t0, beams, pn = a.t0.copy(), a.beams.copy(), a.path_nodes.copy()
    real = np.zeros(len(t0), dtype=bool)   # (the two BOS nodes of a sentence have no T0 score, and sentences that
    for s in range(len(lines)):             # needed stage 2 were relocated: their first region is never written)
        if int(a.status[s]) == 0:
            real[int(a.node_base[s]) + 2:int(a.node_base[s]) + int(a.nnodes[s])] = True
    ctx.set_weights(ctx.weights * np.float32(2.0))
    b = ctx.analyze(lines).fetch(full=True)
    assert np.array_equal(b.t0[real], t0[real] * np.float32(2.0)) and np.array_equal(b.path_nodes, pn)
    live = (beams['left'] != 0xffff) & real[:, None]
    assert np.array_equal(b.beams['total'][live], beams['total'][live] * np.float32(2.0))
    assert np.array_equal(b.beams['prev_node'][live], beams['prev_node'][live])
    ctx.set_weights(ctx.weights)
    c = ctx.analyze(lines).fetch(full=True)
    assert np.array_equal(c.t0[real], t0[real]) and np.array_equal(c.beams[real], beams[real])
    with pytest.raises(J.JppGpuError, match='weight count'):
        ctx.set_weights(ctx.weights[:-1])


def test_emulated_weight_upload(emu_lib, golden_dir):
    check_set_weights(emu_lib, golden_dir)


def test_emulated_rnn_tiny_and_degenerate_batches(emu_lib, golden_dir):
    """the lock-step RNN kernels with almost empty workgroups: batches of 1, 3 and 33 sentences must give what the
    same sentences give inside the golden batch; a batch of only empty / invalid sentences must not hang or crash."""
    img = os.path.join(golden_dir, 'mini_rnn.img')
    lines = [l.rstrip('\n') for l in open(os.path.join(golden_dir, 'mini.txt'), encoding='utf-8')]
    ctx = J.Context(img, lib_path=emu_lib)
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


def check_shared_model_contexts(emu_lib, golden_dir):
    """jppgpu_ctx_create_shared: a second context on the first one's copy of the model -- own configuration, same
    results as a context of its own; the tables outlive the base context"""
    lines = [l.rstrip('\n') for l in open(os.path.join(golden_dir, 'mini.txt'), encoding='utf-8')]
    for image, gold_a, gold_b in (('mini.img', 'mini.gold', 'mini_b3.gold'), ('mini_rnn.img', 'mini_rnn.gold', None)):
        base = J.Context(os.path.join(golden_dir, image), lib_path=emu_lib)
        cfg_b = dict(beam=3, global_beam=4, right_check=2, right_beam=3) if gold_b else {}
        other = J.Context(os.path.join(golden_dir, image), lib_path=emu_lib, share_with=base, **cfg_b)
        for ctx, gold_name in ((base, gold_a), (other, gold_b or gold_a)):
            meta, gold = G.read_gold(os.path.join(golden_dir, gold_name))
            res = ctx.analyze(lines).fetch(full=True)
            errs = []
            for s_ in range(len(lines)):
                errs += G.compare_sentence(res, s_, gold[s_], meta)
            assert not errs, (image, gold_name, errs[:5])
            res.release()
        base.close()   # the shared tables must survive their first owner
        meta, gold = G.read_gold(os.path.join(golden_dir, gold_b or gold_a))
        res = other.analyze(lines).fetch(full=True)
        errs = []
        for s_ in range(len(lines)):
            errs += G.compare_sentence(res, s_, gold[s_], meta)
        assert not errs, errs[:5]


def test_shared_model_contexts(emu_lib, golden_dir):
    check_shared_model_contexts(emu_lib, golden_dir)


def check_length_primitives_spec(lib, ref_tools, tmp):
    """SURVEY 8 f3, LENGTH primitives (ByteLength / CodepointSize, feature_impl_prim.h:114-156): `ref_dump bootstrapv len`
    builds a jumandic variant with a unigram over three of them.  THE REFERENCE CANNOT ANALYSE WITH SUCH A SPEC: its
    pattern pass over the EOS node sends the EOS entry pointer to ExtraNodesContext::lengthOf and segfaults on the first
    sentence (profiles/r05_d_length_primitives.txt), so there are no goldens to compare scores with.  What is checked:
    the model loads only when the column storages are handed in (jppgpu_config::field_storages), the analysis runs, the
    lattice (which no feature influences) is the one of the same dictionary under the `drop` variant, and the T0 scores
    differ from it exactly on nodes -- i.e. the new unigram is evaluated."""
    img_len, lines, _ = _variant_spec_workload(ref_tools, os.path.join(tmp, 'len'), 'len', 30, gold=False)
    img_drop, lines2, _ = _variant_spec_workload(ref_tools, os.path.join(tmp, 'drop'), 'drop', 30, gold=False)
    assert lines == lines2
    a = J.Context(img_len, lib_path=lib).analyze(lines).fetch(full=True)
    b = J.Context(img_drop, lib_path=lib).analyze(lines).fetch(full=True)
    assert list(a.status) == list(b.status) and int((a.status == 0).sum()) >= 25
    assert np.array_equal(a.nnodes, b.nnodes)
    differ = 0
    for s_ in range(len(lines)):   # (per sentence: the node tables have unwritten gaps between the sentences)
        if a.status[s_] != 0:
            continue
        na, nb_, k = int(a.node_base[s_]), int(b.node_base[s_]), int(a.nnodes[s_])
        xa, xb = a.nodes[na:na + k], b.nodes[nb_:nb_ + k]
        assert np.array_equal(xa['eptr'], xb['eptr']) and np.array_equal(xa['start'], xb['start']) and np.array_equal(xa['end'], xb['end'])
        differ += int((a.t0[na + 2:na + k] != b.t0[nb_ + 2:nb_ + k]).sum())
    assert differ > 0
    # without the storages the spec is refused, with the reason
    import ctypes as C
    ctx = J.Context.__new__(J.Context)
    try:
        J.Context.__init__(ctx, img_len, lib_path=lib, _no_field_storages=True)
        assert False, 'a spec with length primitives was accepted without the column storages'
    except J.JppGpuError as e:
        assert 'value storage was not given' in str(e), str(e)


def test_emulated_length_primitives_spec(emu_lib, ref_tools, tmp_path):
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    check_length_primitives_spec(emu_lib, ref_tools, str(tmp_path))


def check_one_enqueue_path(lib, golden_dir):
    """round 5: a batch is ONE enqueue against the capacity the context holds (k_cap_guard); the first batch of an
    unreserved context and a batch that does not fit run the sized way.  Every way must give the reference's lattice."""
    lines = [l.rstrip('\n') for l in open(os.path.join(golden_dir, 'mini.txt'), encoding='utf-8')]
    for image, gold_name in (('mini.img', 'mini.gold'), ('mini_rnn.img', 'mini_rnn.gold')):
        meta, gold = G.read_gold(os.path.join(golden_dir, gold_name))

        def check(res, idx):
            errs = []
            for k, s_ in enumerate(idx):
                errs += G.compare_sentence(res, k, gold[s_], meta)
            assert not errs, (image, errs[:5])
        all_idx = list(range(len(lines)))
        # (a) unreserved: sized, then one enqueue twice; the same lattice every time
        ctx = J.Context(os.path.join(golden_dir, image), lib_path=lib)
        for rep in range(3):
            res = ctx.analyze(lines).fetch(full=True)
            check(res, all_idx)
            res.release()
        st = ctx.stats()
        assert st['sized_batches'] == 1 and st['one_enqueue_batches'] == 2 and st['one_enqueue_overflows'] == 0, st
        allocs = st['device_allocations']
        res = ctx.analyze(lines).fetch(full=True)
        check(res, all_idx)
        res.release()
        assert ctx.stats()['device_allocations'] == allocs, 'a steady batch allocated device memory'
        ctx.close()
        # (b) a small batch first, then the whole file: the second batch does not fit what the first one left, is run
        # again the sized way and comes out right; the third fits
        ctx = J.Context(os.path.join(golden_dir, image), lib_path=lib)
        res = ctx.analyze(lines[:2]).fetch(full=True)
        check(res, [0, 1])
        res.release()
        many = lines * 3
        idx3 = all_idx * 3
        res = ctx.analyze(many).fetch(full=True)
        check(res, idx3)
        res.release()
        st = ctx.stats()
        assert st['one_enqueue_overflows'] == 1 and st['sized_batches'] == 2, st
        res = ctx.analyze(many).fetch(full=True)
        check(res, idx3)
        res.release()
        st = ctx.stats()
        assert st['one_enqueue_overflows'] == 1 and st['one_enqueue_batches'] == 2, st
        ctx.close()
        # (c) reserved: one enqueue from the first batch on, nothing allocated by the batches
        ctx = J.Context(os.path.join(golden_dir, image), lib_path=lib)
        nbytes = sum(len(l.encode('utf-8')) for l in lines)
        ctx.reserve(len(lines), nbytes, nodes_per_byte=8.0)
        allocs = ctx.stats()['device_allocations']
        for rep in range(2):
            res = ctx.analyze(lines).fetch(full=True)
            check(res, all_idx)
            res.release()
        st = ctx.stats()
        assert st['sized_batches'] == 0 and st['one_enqueue_batches'] == 2 and st['one_enqueue_overflows'] == 0, st
        assert st['device_allocations'] == allocs, (allocs, st)
        # ... and a reservation that is too small only costs the second run
        ctx.close()
        ctx = J.Context(os.path.join(golden_dir, image), lib_path=lib)
        ctx.reserve(len(lines), nbytes, nodes_per_byte=0.05)
        res = ctx.analyze(lines).fetch(full=True)
        check(res, all_idx)
        res.release()
        st = ctx.stats()
        assert st['one_enqueue_overflows'] == 1 and st['sized_batches'] == 1, st
        ctx.close()


def test_emulated_one_enqueue_path(emu_lib, golden_dir):
    check_one_enqueue_path(emu_lib, golden_dir)


def check_wide_global_beam_candidates(lib, ref_tools, tmp, beams, nhom=(3, 9, 14, 20), n_lines=6):
    """the wide sweep variant's global beam by candidate count: surfaces with 3 / 9 / 14 / 20 homographs put 96, 288, 448
    and 640 (left node, slot) candidates on a boundary at beam 32 -- one key per lane, the per-lane-maxima prefilter
    with one and with several keys per lane, and beyond 512 the G rounds over HBM; ties among homographs everywhere"""
    import test_gpu_parity as tg
    surf = ['かき', 'さけ', 'くも', 'はし']
    extra = ''
    for sf, k in zip(surf, nhom):
        extra += ''.join('%s,0,0,0,名詞,普通名詞,*,*,%s,よみ%d,%s/よみ%d,代表表記:%s/よみ%d\n' % (sf, sf, i, sf, i, sf, i) for i in range(k))
    lines_x = ['かきをさけとくもにはしで', 'かきかきさけさけくもくもはしはし', 'はしのくものさけのかき', 'くもくもくもさけ', 'はしはしかき']
    img, lines, gold_path = tg._fresh_workload(ref_tools, tmp, 2500, n_lines, 15, 53, length=40, beams=beams, extra_dict=extra,
                                               extra_lines=lines_x)
    ctx = J.Context(img, lib_path=lib, beam=beams[0], global_beam=beams[1], right_check=beams[2], right_beam=beams[3])
    meta, gold = G.read_gold(gold_path)
    res = ctx.analyze(lines).fetch(full=True)
    errs, most = [], 0
    for s_ in range(len(lines)):
        errs += G.compare_sentence(res, s_, gold[s_], meta)
        bb = int(res.bnd_base[s_])
        most = max(most, int(res.end_count[bb + 2:bb + int(res.ncp[s_]) + 3].max()))
    assert most >= max(nhom), most
    assert not errs, (len(errs), errs[:10])


@pytest.mark.parametrize('beams', [[32, 32, 1, 32], [24, 32, 2, 16], [32, 20, 1, 32]])
def test_emulated_wide_global_beam_candidates(emu_lib, ref_tools, tmp_path, beams):
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    check_wide_global_beam_candidates(emu_lib, ref_tools, str(tmp_path), beams)
