"""Trainer hook-up (second half): gold-seed injection through the C ABI, and the device-driven training
loop (`jumanpp_gpu_train`) against the reference's own trainer binary `oracle/_ref/jumanpp_v2_train` -- the model FILES the
two write must be byte-identical for --batch 1 (the reference's deterministic case; with more examples in flight its
result depends on thread timing, see host/train/train_env.h).

Corpus: synthetic sentences analysed by the reference with a random-weight "teacher" model and printed with
`jumanpp_v2 --full-morph` = the trainer's Morph input format; about one gold word per sentence is an UNK analysis that the
trained lattice does not contain, so the gold-node path (TrainingExampleAdapter::ensureNodes) is exercised.
oracle/_ref's scw.cc is built with -ffp-contract=off (oracle/Makefile), like the host trainer here."""
import os
import subprocess
import sys

import numpy as np
import pytest

import jumanpp_amd as J

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GB = ['--gb-left-min=6', '--gb-left-max=6', '--gb-rcheck-min=1', '--gb-rcheck-max=1', '--gb-right-min=5', '--gb-right-max=5']


def _training_set(ref, tmp, n_lines, dict_entries=20000, seed=7):
    """(seed model, Morph-format corpus, raw sentences) in directory tmp"""
    mdic = os.path.join(tmp, 'd.mdic')
    with open(mdic, 'w', encoding='utf-8') as f:
        subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'gen_dict.py'), str(dict_entries), '--seed', str(seed)], stdout=f)
    seed_model = os.path.join(tmp, 'seed.model')
    subprocess.check_call([os.path.join(ref, 'jpp_jumandic_bootstrap'), mdic, seed_model], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    teacher = os.path.join(tmp, 'teacher.model')
    subprocess.check_call([os.path.join(ref, 'ref_dump'), 'mkmodel', seed_model, teacher, '18', '11', '0.1'])
    raw = os.path.join(tmp, 'raw.txt')
    gen = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'gen_corpus.py'), mdic, str(n_lines + n_lines // 20 + 8), '--seed', '5',
                          '--len', '24', '--oov', '0.08'], stdout=subprocess.PIPE, check=True).stdout.decode('utf-8')
    # (the Morph corpus format cannot express words that contain its separators: jumanpp_v2 --full-morph does not quote)
    keep = [l for l in gen.split('\n') if l and not any(c in l for c in ' _"#,')][:n_lines]
    assert len(keep) == n_lines
    with open(raw, 'w', encoding='utf-8') as f:
        f.write('\n'.join(keep) + '\n')
    out = subprocess.run([os.path.join(ref, 'jumanpp_v2'), '--model=' + teacher, '--full-morph', raw], stdout=subprocess.PIPE,
                         stderr=subprocess.DEVNULL, check=True).stdout.decode('utf-8')
    corpus = os.path.join(tmp, 'train.txt')
    with open(corpus, 'w', encoding='utf-8') as f:
        for line in out.split('\n'):
            if line.strip():
                f.write(line.rstrip(' ') + '\n')   # (a trailing separator is an empty word to the reference's reader)
    return seed_model, corpus, raw


def _train_both(ref, trainer, tmp, seed_model, corpus, flags, tag):
    r = os.path.join(tmp, 'ref_%s.model' % tag)
    g = os.path.join(tmp, 'gpu_%s.model' % tag)
    for p in (r, g):
        if os.path.exists(p):
            os.remove(p)   # (the reference does not truncate an existing output file)
    subprocess.check_call([os.path.join(ref, 'jumanpp_v2_train'), '--model-input=' + seed_model, '--model-output=' + r, '--corpus=' + corpus,
                           '--batch=1', '--threads=1'] + flags, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    p = subprocess.run([trainer, '--model-input=' + seed_model, '--model-output=' + g, '--corpus=' + corpus, '--batch=1'] + flags,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    return r, g, p.stderr.decode()


def _gold_added(log):
    return int(log.split(' gold nodes added')[0].split()[-1])


@pytest.fixture(scope='module')
def emu_trainer(emu_lib):
    import __graft_entry__ as ge
    ge.build_host_emu()
    return ge.TRAIN_CLI_EMU


@pytest.fixture(scope='module')
def small_set(ref_tools, tmp_path_factory):
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    if not os.path.exists(os.path.join(ref_tools, 'jumanpp_v2_train')):
        pytest.skip('oracle/_ref has no trainer binary')
    return _training_set(ref_tools, str(tmp_path_factory.mktemp('train')), 300)


def test_trained_model_file_identical_to_reference_trainer(emu_trainer, ref_tools, small_set, tmp_path):
    """two epochs over 300 examples, the reference's defaults (full mode, beam 5) with a global beam: same bytes in the output model"""
    seed_model, corpus, _ = small_set
    r, g, log = _train_both(ref_tools, emu_trainer, str(tmp_path), seed_model, corpus, ['--size=15', '--max-epochs=2', '--epsilon=0'] + GB, 'a')
    assert _gold_added(log) > 200
    assert open(r, 'rb').read() == open(g, 'rb').read()


@pytest.mark.parametrize('flags', [
    # two epochs with the global beam interpolated between them, max-violation updates
    ['--training-mode=violation', '--max-epochs=2', '--epsilon=0', '--gb-left-min=4', '--gb-left-max=8', '--gb-rcheck-min=1',
     '--gb-rcheck-max=2', '--gb-right-min=3', '--gb-right-max=6', '--size=14', '--seed=77'],
    # fall-off-the-beam updates, left beam only, other SCW parameters
    ['--training-mode=falloff', '--beam=3', '--gb-left-min=5', '--gb-left-max=5', '--size=16', '--scw-c=0.5', '--scw-phi=2'],
    # several passes over every batch
    ['--max-batch-iters=3', '--epsilon=0'] + GB,
    # the reference trainer's own defaults: no global beam at all (full-beam scoring, k_sweep_full<DYN>)
    ['--size=15'],
    ['--beam=3', '--training-mode=violation', '--max-epochs=2', '--epsilon=0'],
    # full beam on the first pass over every batch, the global beam on the second
    ['--gb-first-full', '--max-batch-iters=2', '--epsilon=0'] + GB,
])
def test_training_modes_epochs_and_batch_iterations(emu_trainer, ref_tools, small_set, tmp_path, flags):
    seed_model, corpus, _ = small_set
    short = os.path.join(str(tmp_path), 'short.txt')
    with open(short, 'w', encoding='utf-8') as f:
        f.writelines(open(corpus, encoding='utf-8').readlines()[:100])
    r, g, _ = _train_both(ref_tools, emu_trainer, str(tmp_path), seed_model, short, flags, 'b')
    assert open(r, 'rb').read() == open(g, 'rb').read()


def test_csv_corpus_equals_morph_corpus_and_trained_model_loads(emu_trainer, emu_lib, ref_tools, small_set, tmp_path):
    """the one-word-per-line corpus format gives the same model; the written .jppmdl is read back by the reference's
    analyser and by this repo's loader, and both analyse with the trained weights identically"""
    seed_model, corpus, raw = small_set
    lines = open(corpus, encoding='utf-8').read().split('\n')[:25]
    morph = os.path.join(str(tmp_path), 'm.txt')
    csv = os.path.join(str(tmp_path), 'c.csv')
    with open(morph, 'w', encoding='utf-8') as f:
        f.write('\n'.join(l for l in lines if l) + '\n')
    with open(csv, 'w', encoding='utf-8') as f:
        for l in lines:
            if l:
                for tok in l.split(' '):
                    f.write(','.join(tok.split('_')) + '\n')
                f.write('\n')
    outs = []
    for path, fmt in ((morph, 'morph'), (csv, 'csv')):
        o = os.path.join(str(tmp_path), fmt + '.model')
        p = subprocess.run([emu_trainer, '--model-input=' + seed_model, '--model-output=' + o, '--corpus=' + path,
                            '--corpus-format=' + fmt] + GB, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()
        outs.append(o)
    assert open(outs[0], 'rb').read() == open(outs[1], 'rb').read()
    # the trained model through both analysers
    import __graft_entry__ as ge
    cli = ge.build_host_emu()
    sents = os.path.join(str(tmp_path), 's.txt')
    with open(sents, 'w', encoding='utf-8') as f:
        f.writelines(open(raw, encoding='utf-8').readlines()[:40])
    want = subprocess.run([os.path.join(ref_tools, 'jumanpp_v2'), '--model=' + outs[0], sents], stdout=subprocess.PIPE,
                          stderr=subprocess.DEVNULL, check=True).stdout
    got = subprocess.run([cli, '--model=' + outs[0], sents], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert got.returncode == 0, got.stderr.decode()
    assert got.stdout == want


def test_gold_seed_hook_through_the_abi(emu_lib, golden_dir):
    """jppgpu_analyze_batch_seeds by hand: extra seeds land behind the makers' nodes of their start position, get the entry
    row and hash they were given, the last UNK entry pointers, and the rest of the lattice is the one the plain analysis
    builds; a sentence without extra seeds is untouched; path n-grams of the top-1 path equal the top-1 read-out."""
    img = os.path.join(golden_dir, 'mini.img')
    lines = [l.rstrip('\n') for l in open(os.path.join(golden_dir, 'mini.txt'), encoding='utf-8')][:6]
    ctx = J.Context(img, lib_path=emu_lib, dynamic_features=True)
    base = ctx.analyze(lines).fetch(full=True)
    seen = {}

    def hook(view):
        seen.update(view)
        extra = [[] for _ in lines]
        # sentence 0: two nodes starting at codepoint 1 (spans 1-3 and 1-2, in that order), one at the last codepoint;
        # sentence 2: one node over the first codepoint
        n0 = int(view['n_codepoints'][0])
        extra[0] = [(1, 3, -12345, [11, 12, 13, 14, 15, 16, 17, 18]), (1, 2, -777, [1, 2, 3, 4, 5, 6, 7, 8]),
                    (n0 - 1, n0, -99, [21, 22, 23, 24, 25, 26, 27, 28])]
        extra[2] = [(0, 1, -5, [31, 32, 33, 34, 35, 36, 37, 38])]
        return extra
    res = ctx.analyze_with_seeds(lines, hook).fetch(full=True)
    # what the hook saw: the seeds of the plain analysis (nodes without BOS, BOS, EOS), UNK entry pointers still unnumbered
    for s in range(len(lines)):
        nb = int(base.node_base[s])
        N = int(base.nnodes[s])
        assert int(seen['n_seeds'][s]) == N - 3
        sb = int(seen['seed_base'][s])
        sv = seen['seeds'][sb:sb + N - 3]
        bn = base.nodes[nb + 2:nb + N - 1]
        assert (sv['start'] == bn['start']).all() and (sv['end'] == bn['end']).all()
        assert ((sv['eptr'] < 0) == (bn['eptr'] < 0)).all()
        assert (sv['eptr'][bn['eptr'] >= 0] == bn['eptr'][bn['eptr'] >= 0]).all()
    added = {0: 3, 2: 1}
    for s in range(len(lines)):
        nb, N = int(res.node_base[s]), int(res.nnodes[s])
        ob, oN = int(base.node_base[s]), int(base.nnodes[s])
        assert N == oN + added.get(s, 0)
        nodes, unk = res.nodes[nb:nb + N], res.unk[nb:nb + N]
        gold = unk['maker'] == 0xffff
        assert int(gold.sum()) == added.get(s, 0)
        # everything else in the order of the plain analysis
        keep = ~gold
        assert (nodes[keep]['start'] == base.nodes[ob:ob + oN]['start']).all()
        assert (nodes[keep]['end'] == base.nodes[ob:ob + oN]['end']).all()
        old_eptr = base.nodes[ob:ob + oN]['eptr']
        assert (nodes[keep]['eptr'] == old_eptr).all()   # the makers' UNK nodes keep their numbers, dictionary nodes their pointers
        assert (res.entry_rows[nb:nb + N][keep][2:] == base.entry_rows[ob:ob + oN][2:]).all()   # (the BOS rows are not materialised)
        if s not in added:
            assert (res.t0[nb + 2:nb + N] == base.t0[ob + 2:ob + oN]).all()
            continue
        n_old_unk = int((old_eptr[2:oN - 1] < 0).sum())
        gi = np.nonzero(gold)[0]
        # behind the last old node of their start, in the order given; numbered after every maker's UNK node
        for rank, k in enumerate(gi):
            assert nodes[k]['eptr'] == ~(n_old_unk + rank)
            st = nodes[k]['start']
            same = np.nonzero(nodes['start'][2:N - 1] == st)[0] + 2
            n_gold_here = int(gold[same].sum())
            assert k in same[-n_gold_here:]
            assert unk[k]['tmpl'] == 0 and unk[k]['ph0'] == 0 and unk[k]['ph1'] == 0
        if s == 0:
            assert [int(nodes[k]['end']) for k in gi] == [3, 2, int(res.ncp[0])]
            assert [int(unk[k]['hash']) for k in gi] == [-12345, -777, -99]
            assert res.entry_rows[nb + gi[0]].tolist() == [11, 12, 13, 14, 15, 16, 17, 18]
            assert res.entry_rows[nb + gi[2]].tolist() == [21, 22, 23, 24, 25, 26, 27, 28]
        # boundary table and ends lists stay consistent
        bb = int(res.bnd_base[s])
        for b in range(2, int(res.ncp[s]) + 3):
            first, cnt = int(res.bnd_first[bb + b]), int(res.bnd_count[bb + b])
            assert (nodes['start'][first:first + cnt] == b - 2).all() or b == int(res.ncp[s]) + 2
        assert sorted(res.end_nodes[nb:nb + N - 1].tolist()) == list(range(N - 1))   # (EOS ends nowhere)
    # path n-grams: the top-1 path given explicitly (text order, EOS last) reproduces the top-1 read-out (EOS first)
    first, pnodes, feats = res.fetch_top1_ngrams()
    pf = np.zeros(len(lines) + 1, dtype=np.uint64)
    pn = []
    for s in range(len(lines)):
        seg = pnodes[int(first[s]):int(first[s + 1])][::-1]
        pn.extend(seg.tolist())
        pf[s + 1] = len(pn)
    given = res.fetch_path_ngrams(pf, np.array(pn, dtype=np.uint32))
    for s in range(len(lines)):
        a, b = int(first[s]), int(first[s + 1])
        assert (given[a:b] == feats[a:b][::-1]).all()


def test_gold_seeds_repair_a_disconnected_sentence(emu_lib, golden_dir):
    """a sentence the makers cannot connect (JPPGPU_SENT_NO_LATTICE; here: a model stripped of its UNK makers) is
    analysable once the trainer supplies the missing nodes (Trainer::prepare, trainer.cc:16-37: the gold nodes are added
    and connectivity is checked again); without them it keeps its status, and connected sentences are untouched"""
    img = os.path.join(golden_dir, 'mini.img')
    ctx = J.Context(img, lib_path=emu_lib, dynamic_features=True, max_unk_makers=0)
    lines = [l.rstrip('\n') for l in open(os.path.join(golden_dir, 'mini.txt'), encoding='utf-8')][:12]
    plain = ctx.analyze(lines).fetch(full=False)
    broken = [s for s in range(len(lines)) if int(plain.status[s]) == 3]
    fine = [s for s in range(len(lines)) if int(plain.status[s]) == 0]
    assert broken and fine, 'the fixture needs connected and disconnected sentences'
    fix = broken[0]

    def hook(view):
        assert int(view['status'][fix]) == 3 and int(view['n_seeds'][fix]) > 0   # the seeds of a disconnected sentence are shown
        extra = [[] for _ in lines]
        extra[fix] = [(i, i + 1, -1000 - i, [0] * 8) for i in range(int(view['n_codepoints'][fix]))]
        return extra
    res = ctx.analyze_with_seeds(lines, hook).fetch(full=False)
    assert int(res.status[fix]) == 0 and int(res.path_len[fix]) >= 2
    for s in broken[1:]:
        assert int(res.status[s]) == 3
    for s in fine:
        assert int(res.status[s]) == 0 and int(res.path_len[s]) == int(plain.path_len[s])
        a, b = int(res.node_base[s]), int(plain.node_base[s])
        n = int(res.path_len[s])
        assert (res.path_nodes[a:a + n] == plain.path_nodes[b:b + n]).all()


@pytest.mark.gpu
def test_gpu_trained_model_identical_and_batched_training(gpu_lib, ref_tools, tmp_path):
    """on the MI355X: 400 examples with --batch 1 -> the reference trainer's model file byte for byte; the same corpus
    with --batch 128 (one device pass per 128 examples, weights frozen inside a batch) trains a model both analysers load"""
    if ref_tools is None or not os.path.exists(os.path.join(ref_tools, 'jumanpp_v2_train')):
        pytest.skip('oracle/_ref not built')
    import __graft_entry__ as ge
    ge.build_host()
    seed_model, corpus, raw = _training_set(ref_tools, str(tmp_path), 400, dict_entries=30000, seed=9)
    r, g, log = _train_both(ref_tools, ge.TRAIN_CLI, str(tmp_path), seed_model, corpus, ['--size=18', '--max-epochs=2', '--epsilon=0'] + GB, 'gpu')
    assert _gold_added(log) > 200
    assert open(r, 'rb').read() == open(g, 'rb').read()
    # the reference trainer's default configuration: no global beam (full-beam scoring, k_sweep_full<DYN>)
    head = os.path.join(str(tmp_path), 'head.txt')
    with open(head, 'w', encoding='utf-8') as f:
        f.writelines(open(corpus, encoding='utf-8').readlines()[:150])
    r2, g2, _ = _train_both(ref_tools, ge.TRAIN_CLI, str(tmp_path), seed_model, head, ['--size=16'], 'gpu_full')
    assert open(r2, 'rb').read() == open(g2, 'rb').read()
    o = os.path.join(str(tmp_path), 'batched.model')
    p = subprocess.run([ge.TRAIN_CLI, '--model-input=' + seed_model, '--model-output=' + o, '--corpus=' + corpus, '--batch=128',
                        '--size=18', '--max-epochs=3', '--epsilon=0'] + GB, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    want = subprocess.run([os.path.join(ref_tools, 'jumanpp_v2'), '--model=' + o, raw], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                          check=True).stdout
    got = subprocess.run([ge.HOST_CLI, '--model=' + o, raw], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert got.returncode == 0, got.stderr.decode()
    assert got.stdout == want
