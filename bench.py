#!/usr/bin/env python3
"""bench.py -- sentences/sec of the Juman++ analysis hot path on MI355X.

One "step" = one pass of the whole hot path (decode -> seeds -> lattice -> T0
-> global-beam sweep -> RNNLM re-ranking -> top-1 path -> packed result) over
one batch of 65,536 synthetic 40-codepoint sentences that is already resident
in HBM.  Workload = BASELINE.json configs[2]: perceptron + RNNLM scorer, beam 5
(global beam 6, right-check 1, right-beam 5 = the CLI defaults), 1M sentences
batched 64k (`--no-rnn`: configs[1], perceptron only; also reported beside the
headline as `perceptron_only`).

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for the
definition of the roofline, cpu_baseline and parity_sample objects.
"""
import argparse
import hashlib
import json
import glob
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
REF = os.path.join(ROOT, 'oracle', '_ref')


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_workload(args, cache_dir):
    """synthetic dictionary + random perceptron + corpus, built with the
    reference's own offline tools (oracle/_ref) -- untimed setup."""
    os.makedirs(cache_dir, exist_ok=True)
    hom = int(getattr(args, 'homographs', 0) or 0)
    dkey = 'd%d_s%d%s' % (args.dict_entries, args.seed, '_h%d' % hom if hom else '')
    key = '%s_w%d%s' % (dkey, args.weights_exp, '_rnn%d' % args.rnn_hidden if args.rnn else '')
    mdic = os.path.join(cache_dir, dkey + '.mdic')
    seed_model = os.path.join(cache_dir, dkey + '.seed')
    model = os.path.join(cache_dir, key + '.model')
    img = os.path.join(cache_dir, key + '.img')
    if not os.path.exists(seed_model):
        with open(mdic, 'w', encoding='utf-8') as f:
            subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'gen_dict.py'), str(args.dict_entries),
                                   '--seed', str(args.seed)] + (['--homographs', str(hom)] if hom else []), stdout=f)
        subprocess.check_call([os.path.join(REF, 'jpp_jumandic_bootstrap'), mdic, seed_model + '.tmp'],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        os.rename(seed_model + '.tmp', seed_model)
    if not os.path.exists(img):
        subprocess.check_call([os.path.join(REF, 'ref_dump'), 'mkmodel', seed_model, model, str(args.weights_exp),
                               str(args.seed), '0.1'])
        if args.rnn:
            # BASELINE configs[2]: + synthetic faster-rnnlm NCE model embedded by the reference's trainer binary
            rnn = os.path.join(cache_dir, dkey + '_rnn%d.rnnlm' % args.rnn_hidden)
            if not os.path.exists(rnn):
                subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'gen_rnn.py'), mdic, rnn, '--vocab',
                                       str(args.rnn_vocab), '--hidden', str(args.rnn_hidden), '--maxent-size',
                                       str(1 << 22), '--seed', str(args.seed)], stdout=subprocess.DEVNULL)
            pmodel = model + '.perceptron'
            os.rename(model, pmodel)
            subprocess.check_call([os.path.join(REF, 'jumanpp_v2_train'), '--model-input=' + pmodel,
                                   '--model-output=' + model, '--rnn-model=' + rnn, '--rnn-fields=surface,pos',
                                   '--rnn-nce-bias=5.62844432562', '--rnn-unk-constant=-3.4748115191',
                                   '--rnn-unk-length=-2.92994951022', '--feature-weight-perceptron=1',
                                   '--feature-weight-rnn=0.0176'], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        subprocess.check_call([os.path.join(REF, 'ref_dump'), 'export', model, img + '.tmp'], stderr=subprocess.DEVNULL)
        os.rename(img + '.tmp', img)
    return mdic, model, img


def make_corpus(args, mdic, cache_dir, n_lines, seed):
    oov = float(getattr(args, 'oov', 0.05))
    zipf = float(getattr(args, 'zipf', 0.0) or 0.0)
    path = os.path.join(cache_dir, 'corpus_%d_%d_%d%s%s.txt' % (n_lines, args.sent_len, seed, '' if oov == 0.05 else '_oov%g' % oov,
                                                                '_zipf%g' % zipf if zipf else ''))
    if not os.path.exists(path):
        with open(path + '.tmp', 'w', encoding='utf-8') as f:
            subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'gen_corpus.py'), mdic, str(n_lines),
                                   '--seed', str(seed), '--len', str(args.sent_len), '--oov', str(oov)] +
                                  (['--zipf', str(zipf)] if zipf else []), stdout=f)
        os.rename(path + '.tmp', path)
    return path


def load_batches(path, batch, np):
    data = open(path, 'rb').read()
    lines = data.split(b'\n')
    if lines and lines[-1] == b'':
        lines.pop()
    out = []
    for i in range(0, len(lines), batch):
        chunk = lines[i:i + batch]
        if len(chunk) < batch:
            break
        offs = np.zeros(len(chunk) + 1, dtype=np.uint32)
        offs[1:] = np.cumsum([len(c) for c in chunk])
        out.append((b''.join(chunk), offs))
    return out


def algorithmic_bytes(res, cfg_beam, cfg_gbeam, rcheck, rbeam, np):
    """Algorithmic HBM bytes of one batch, per kernel, from the fetched lattice
    (DESIGN.md "Algorithmic bytes").  4 B per weight gather, no sector rounding."""
    ok = res.status == 0
    N = int(res.nnodes[ok].sum())
    # T0 kernel: node info+aux (24) + entry row read (~12 varint bytes) + 32 gathers + writes (entry 32, pat 112, t0 4)
    t0_bytes = N * (24 + 12 + 32 * 4 + 32 + 112 + 4)
    total = 0
    R = res.bnd_count.astype(np.int64)
    L = res.end_count.astype(np.int64)
    ngb = res.gbeam_count.astype(np.int64)
    scored = np.zeros(len(R), dtype=bool)
    for s in np.nonzero(ok)[0]:
        b0 = int(res.bnd_base[s])
        scored[b0 + 2: b0 + int(res.ncp[s]) + 3] = True
    scored &= R > 0
    Rs, Ls, Gs = R[scored], L[scored], ngb[scored]
    gl = res.gbeam_entries['left'][scored]
    U = np.zeros(len(Rs), dtype=np.int64)
    for j in range(cfg_gbeam):
        col = gl[:, j]
        new = Gs > j
        for k in range(j):
            new &= ~((gl[:, k] == col) & (Gs > k))
        U += new
    c = np.minimum(np.minimum(rcheck, Rs), Gs)
    K = np.minimum(rbeam, Rs) if rcheck > 0 else Rs
    slot = 16
    sweep = (Ls * cfg_beam * slot                      # left beams read for the global beam
             + (U + Gs) * 112                           # T1 / T2 pattern rows
             + Rs * (112 + 4)                           # right-node patterns + T0
             + 4 * (c * Rs * 41 + K * (U * 37 + np.maximum(Gs - c, 0) * 4))  # bi/tri weight gathers
             + Rs * cfg_beam * slot                     # beams written
             + (K * Gs + (Rs - K) * c) * 4)             # score cells written
    # gathers actually issued: the bigram weights of (kept right node, T1 row 0) serve prescore and tail
    issued = sweep - 4 * K * 37 * (c > 0)
    gathers = c * Rs * 41 + K * (U * 37 + np.maximum(Gs - c, 0) * 4) - K * 37 * (c > 0)
    return dict(t0=int(t0_bytes), sweep=int(sweep.sum()), sweep_issued=int(issued.sum()), nodes=N, sweep_gathers=int(gathers.sum()))


def kernel_source_id():
    """sha256 over the kernel sources: profiles/traffic.json carries the id of the sources its counters were collected
    with, and roofline.traffic is only reported when it equals the running library's (a counter profile must not
    survive a kernel change silently)"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'jumanpp_amd', 'csrc')
    for name in sorted(os.listdir(d)):
        fp = os.path.join(d, name)
        if os.path.isfile(fp):
            h.update(name.encode())
            h.update(open(fp, 'rb').read())
    return h.hexdigest()[:16]


def front_end_bytes(img, res, lines, np):
    """Algorithmic bytes of the front end (decode -> seeds -> lattice layout) per SURVEY 8(d): 4 B per double-array
    unit touched (B_t) + entry-pointer list bytes (P), counted by walking the model image's trie on the host for a
    sample of sentences, + the sentence bytes + what the front end writes per codepoint / boundary / node.  Returns
    bytes per sentence (mean over the sample)."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import model_image as mi
    secs = mi.read_sections(img)
    units = np.frombuffer([p for t, a, p in secs if t == mi.SEC_TRIE][0], dtype='<u4')
    eptrs = [p for t, a, p in secs if t == mi.SEC_ENTRY_PTRS][0]

    def varint_len(pos):
        n = 1
        while eptrs[pos + n - 1] & 0x80:
            n += 1
        return n

    def list_bytes(pos):
        v, shift, q = 0, 0, pos
        while True:
            b = eptrs[q]
            q += 1
            v |= (b & 0x7f) << shift
            shift += 7
            if b < 0x80:
                break
        for _ in range(v):
            q += varint_len(q)
        return q - pos

    bt = p_bytes = 0
    for line in lines:
        raw = line
        cps = raw.decode('utf-8', errors='ignore')
        starts, off = [], 0
        for ch in cps:
            starts.append(off)
            off += len(ch.encode('utf-8'))
        starts.append(off)
        for i in range(len(cps)):
            idn = 0
            unit = int(units[0])
            bt += 1
            ok = True
            for j in range(i, len(cps)):
                for b in raw[starts[j]:starts[j + 1]]:
                    offs = (unit >> 10) << ((unit & (1 << 9)) >> 6)
                    idn ^= offs ^ b
                    unit = int(units[idn])
                    bt += 1
                    if (unit & ((1 << 31) | 0xff)) != b:
                        ok = False
                        break
                if not ok:
                    break
                if (unit >> 8) & 1:
                    offs = (unit >> 10) << ((unit & (1 << 9)) >> 6)
                    leaf = int(units[idn ^ offs])
                    bt += 1
                    p_bytes += list_bytes(leaf & 0x7fffffff)
    n = max(1, len(lines))
    ok = res.status == 0
    nodes = float(res.nnodes[ok].sum()) / max(1, int(ok.sum()))
    ncp = float(res.ncp[ok].sum()) / max(1, int(ok.sum()))
    nbytes = sum(len(l) for l in lines) / n
    per_cp = 4 + 4 + 2 + 20 + 48 + 6 + 8          # codepoint, class, byte offset, charlattice nodes, walk record, counts, end mask
    per_bnd = 16 + 16 + 16                        # first / count arrays, ends first / count, the packed boundary record
    per_node = 8 + 16 + 4                         # node record, UNK record, ends-list entry
    return {'trie_units': bt / n, 'entry_pointer_bytes': p_bytes / n,
            'bytes_per_sentence': 4 * bt / n + p_bytes / n + nbytes + ncp * per_cp + (ncp + 3) * per_bnd + nodes * per_node}


def _cpu_flags():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('flags'):
                return set(line.split(':', 1)[1].split())
    except OSError:
        pass
    return set()


def usable_cores():
    """hardware threads this process may really use: affinity mask and cgroup CPU quota, not just what is visible"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ('/sys/fs/cgroup/cpu.max',):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != 'max':
                n = min(n, max(1, int(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    try:
        q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        if q > 0 and per > 0:
            n = min(n, max(1, q // per))
    except (OSError, ValueError):
        pass
    return n


def reference_build():
    """oracle/_ref is built for haswell-class CPUs (-O3 -march=haswell); oracle/_ref/v4 is the same reference
    built -O3 -march=x86-64-v4 (AVX-512).  /root/reference is not on the GPU box, so `-march=native` there
    means: the most specific of the two this host can run."""
    v4 = os.path.join(REF, 'v4')
    need = {'avx512f', 'avx512bw', 'avx512vl', 'avx512dq', 'avx512cd'}
    if os.path.exists(os.path.join(v4, 'ref_dump')) and need <= _cpu_flags():
        return v4, 'g++ -O3 -march=x86-64-v4'
    return REF, 'g++ -O3 -march=haswell'


def _ref_time(ref_dir, model, corpus):
    with open(corpus, 'rb') as f:
        out = subprocess.check_output([os.path.join(ref_dir, 'ref_dump'), 'time', model], stdin=f)
    return json.loads(out.decode())


def reference_top1(ref_dir, model, text, offs, np, tmp_dir, procs=None, beams=None):
    """CHECKER: the reference's packed top-1 result (oracle/ref_dump.cc `top1`: Analyzer::analyze per sentence, then
    {EntryPtr, start, end} of the best path in text order) for the sentences text[offs[i]:offs[i+1]], computed by one
    reference process per usable core.  Returns (status[n], offsets[n+1], items[m] as (eptr i32, start u16, end u16))."""
    n = len(offs) - 1
    procs = max(1, min(procs or usable_cores(), (n + 255) // 256))
    per = (n + procs - 1) // procs
    os.makedirs(tmp_dir, exist_ok=True)
    running = []
    for k in range(procs):
        lo, hi = k * per, min(n, (k + 1) * per)
        if lo >= hi:
            break
        part = os.path.join(tmp_dir, 'top1_part%d.txt' % k)
        with open(part, 'wb') as f:
            for i in range(lo, hi):
                f.write(text[offs[i]:offs[i + 1]])
                f.write(b'\n')
        out = part + '.bin'
        fin = open(part, 'rb')
        running.append((subprocess.Popen([os.path.join(ref_dir, 'ref_dump'), 'top1', model, out] +
                                         [str(int(x)) for x in (beams or [])], stdin=fin,
                                         stderr=subprocess.DEVNULL), fin, part, out, hi - lo))
    item_dt = np.dtype([('eptr', '<i4'), ('start', '<u2'), ('end', '<u2')])
    status, counts, items = [], [], []
    for pr, fin, part, out, cnt in running:
        rc = pr.wait()
        fin.close()
        if rc != 0:
            raise RuntimeError('ref_dump top1 failed (rc %d)' % rc)
        raw = open(out, 'rb').read()
        os.remove(part)
        os.remove(out)
        magic, m = np.frombuffer(raw, dtype='<u4', count=2)
        assert magic == 0x31504f54 and m == cnt, (hex(int(magic)), m, cnt)
        pos = 8
        for _ in range(cnt):
            st, c = np.frombuffer(raw, dtype='<u4', count=2, offset=pos)
            pos += 8
            status.append(int(st))
            counts.append(int(c))
            items.append(np.frombuffer(raw, dtype=item_dt, count=int(c), offset=pos))
            pos += 8 * int(c)
    offsets = np.zeros(n + 1, dtype=np.int64)
    offsets[1:] = np.cumsum(counts)
    return np.array(status, dtype=np.int32), offsets, (np.concatenate(items) if items else np.zeros(0, dtype=item_dt))


def compare_packed(dev_offs, dev_items, ref_status, ref_offs, ref_items, np):
    """sentences whose packed top-1 result (EntryPtr raw value incl. the UNK numbering, start, end of every morpheme)
    differs from the reference's; dev_items is the (m, 2) int32 array of jppgpu_result_pack"""
    item_dt = np.dtype([('eptr', '<i4'), ('start', '<u2'), ('end', '<u2')])
    di = np.ascontiguousarray(dev_items).view(item_dt).reshape(-1)
    do = dev_offs.astype(np.int64)
    n = len(ref_offs) - 1
    bad = []
    same_len = (do[1:n + 1] - do[:n]) == (ref_offs[1:] - ref_offs[:-1])
    for s in range(n):
        if ref_status[s] != 0:
            if do[s + 1] != do[s]:
                bad.append(s)
            continue
        if not same_len[s] or not np.array_equal(di[do[s]:do[s + 1]], ref_items[ref_offs[s]:ref_offs[s + 1]]):
            bad.append(s)
    return bad


def parity_sample(ref_dir, model, batches, run_packed, np, tmp_dir, n_batches=2):
    """CHECKER (untimed): the first `n_batches` timed batches once more through the bench path (analyze_device +
    jppgpu_result_pack), every sentence's packed top-1 result against the reference run on this box's cores."""
    t = time.time()
    total, mism, first = 0, 0, []
    for i in range(min(n_batches, len(batches))):
        text, offs = batches[i]
        d_offs, d_items = run_packed(i)
        rs, ro, ri = reference_top1(ref_dir, model, text, offs, np, tmp_dir)
        bad = compare_packed(d_offs, d_items, rs, ro, ri, np)
        total += len(offs) - 1
        mism += len(bad)
        first += [(i, int(b)) for b in bad[:4]]
    return {'sentences': total, 'mismatches': mism, 'first_mismatches': first[:8],
            'what': 'packed top-1 result (EntryPtr incl. UNK numbering, start, end per morpheme) of the first %d timed '
                    'batches, bench path (analyze_device + jppgpu_result_pack), vs the reference Analyzer::analyze '
                    '(oracle/_ref ref_dump top1, %d processes); %.1f s' % (min(n_batches, len(batches)), usable_cores(), time.time() - t)}


# measured ceilings of the chip this bench runs on (profiles/r04_b_gather_policy.txt, r04_c_gather_policy.txt: random 4-byte
# gathers from 64 MB .. 1 GB tables, every cache policy and allocation flavour): an L2 miss is ONE 128-byte request to the
# fabric (TCC_EA0_RDREQ_128B == TCC_MISS), and the chip serves 59-63 G of them per second = 7.5-8.1 TB/s
GATHER_MISS_PEAK = 63.0e9      # lines/s, 64 MB table (the headline's weight table)
SCLK_HZ = 2.4e9                # MI355X peak engine clock; SIMD-cycles available = duration x SCLK x 1024 SIMDs
F32_MFMA_PEAK = 157.3e12       # dense f32 MFMA, MI355X_MICROARCH.md


def load_profile(path, args, src_id):
    """a committed counter profile (traffic.json / counters.json) -- only if it was collected on THIS workload with THESE
    kernel sources; otherwise (None, why)"""
    try:
        tp = json.load(open(path))
    except (OSError, ValueError):
        return None, 'no profile at %s' % os.path.relpath(path, ROOT)
    same = (tp.get('batch') == args.batch and tp.get('rnn') == bool(args.rnn) and tp.get('sent_len') == args.sent_len
            and tp.get('dict_entries') == args.dict_entries and tp.get('weights_exp', 22) == args.weights_exp)
    if not same:
        return None, '%s was collected on another workload' % os.path.basename(path)
    if tp.get('kernel_source_id') != src_id:
        return None, ('%s was collected with other kernel sources (%s, running %s): not reported'
                      % (os.path.basename(path), tp.get('kernel_source_id'), src_id))
    return tp, tp.get('note')


def valu_roofline(counters, kernel_prefix, launch_ms, gathers):
    """the vector-ALU yard-stick of a kernel that is not a stream.  achieved = wave-instructions x 4
    cycles (SQ_INSTS_VALU; the quarter-rate share -- 32-bit multiplies of the hash -- is not counted separately by the
    hardware and is left out, so this is a LOWER bound of the busy cycles), peak = SIMD-cycles of the launch."""
    ent = None
    for k, v in counters.items():
        if k.startswith(kernel_prefix) and 'SQ_INSTS_VALU' in v and (ent is None or v['SQ_INSTS_VALU'] > ent['SQ_INSTS_VALU']):
            ent, name = v, k
    if ent is None:
        return None
    insts = ent['SQ_INSTS_VALU']
    peak = launch_ms * 1e-3 * SCLK_HZ * 1024
    lanes = ent.get('SQ_THREAD_CYCLES_VALU', 0.0) / max(1.0, ent.get('SQ_ACTIVE_INST_VALU', insts))
    out = {'bound': 'valu', 'kernel': name, 'achieved': round(insts * 4 / 1e9, 3), 'peak': round(peak / 1e9, 3),
           'unit': 'G SIMD-cycles per launch', 'frac': round(insts * 4 / peak, 4),
           'wave_instructions_valu': int(insts), 'wave_instructions_salu': int(ent.get('SQ_INSTS_SALU', 0)),
           'active_lanes_per_valu_instruction': round(lanes, 1),
           'what': 'SQ_INSTS_VALU x 4 cycles / (launch duration x %.1f GHz x 1024 SIMDs); quarter-rate multiplies counted '
                   'as full-rate (lower bound)' % (SCLK_HZ / 1e9)}
    if gathers:
        out['lane_instructions_per_gathered_weight'] = round(insts * lanes / gathers, 1)
    return out


def leg_parity(model, result, n_batch, text, offs, n_check, beams, np, torch, dev, tmp_dir, max_len):
    """CHECKER (untimed) of an extra leg: `result` is the leg's OWN full-size batch as the timed loop analysed it; its
    packed top-1 result is compared, for the first `n_check` sentences, with the reference run on this box's cores
    under the leg's beam configuration (ref_dump top1 <beam gbeam rcheck rbeam>).  The device work is the benchmarked
    shape; only the number of sentences the reference re-analyses is bounded."""
    try:
        t = time.time()
        cap = n_batch * (max_len + 1)
        d_offs = torch.zeros(n_batch + 1, dtype=torch.int32, device=dev)
        d_items = torch.zeros((cap, 2), dtype=torch.int32, device=dev)
        result.pack(d_offs.data_ptr(), d_items.data_ptr(), cap)
        torch.cuda.synchronize()
        ho = d_offs.cpu().numpy().view(np.uint32)
        n_check = min(n_check, n_batch)
        hi = d_items[:int(ho[n_check])].cpu().numpy()
        rs, ro, ri = reference_top1(reference_build()[0], model, text, offs[:n_check + 1], np, tmp_dir, beams=beams)
        bad = compare_packed(ho[:n_check + 1], hi, rs, ro, ri, np)
        return {'sentences': n_check, 'mismatches': len(bad), 'first_mismatches': [int(b) for b in bad[:8]],
                'what': 'packed top-1 result of the first %d sentences of the leg\'s own %d-sentence batch vs the reference '
                        '(ref_dump top1, beams %s, %d processes); %.1f s' % (n_check, n_batch, list(beams), usable_cores(), time.time() - t)}
    except Exception as e:  # the checker must never take the line down -- but it must say so
        return {'error': str(e)[:300]}


def cpu_baseline(args, model, mdic, cache_dir):
    """The real reference (Analyzer::analyze in a loop, oracle/ref_dump.cc `time`) on this box's host cores:
    one thread on the timed workload (with the RNN), one thread perceptron-only, and every core at once
    (one process per core -- the reference has no threading)."""
    ref_dir, flags = reference_build()
    corpus = make_corpus(args, mdic, cache_dir, args.cpu_sample, args.seed + 1000)
    t = time.time()
    r = _ref_time(ref_dir, model, corpus)
    out = {
        'value': round(r['sent_per_s_analyze'], 1),
        'unit': 'sentences/s',
        'cores': 1,
        'kind': 'reference',
        'sample': '%d sentences of the same synthetic workload through the reference Analyzer::analyze '
                  '(oracle/_ref, %s, 1 thread, best of 3; %.1f s wall)'
                  % (r['sentences'], flags, time.time() - t),
        'with_juman_format': round(r['sent_per_s_total'], 1),
    }
    pmodel = model + '.perceptron'
    if args.rnn and os.path.exists(pmodel):
        rp = _ref_time(ref_dir, pmodel, corpus)
        out['perceptron_only'] = {'value': round(rp['sent_per_s_analyze'], 1), 'unit': 'sentences/s', 'cores': 1,
                                  'what': 'same sample, same model without the RNN part (BASELINE configs[0]/[1] scorer)'}
    # all cores: one process per hardware thread, each analysing the same sample (identical work per process,
    # so the aggregate is what a split corpus would give); rates are taken while all processes run
    ncore = usable_cores()
    small = make_corpus(args, mdic, cache_dir, max(2000, args.cpu_sample // 4), args.seed + 1001)
    t = time.time()
    procs = []
    for _ in range(ncore):
        f = open(small, 'rb')
        procs.append((subprocess.Popen([os.path.join(ref_dir, 'ref_dump'), 'time', model], stdin=f,
                                       stdout=subprocess.PIPE), f))
    agg, nsent = 0.0, 0
    for pr, f in procs:
        o, _ = pr.communicate()
        f.close()
        if pr.returncode == 0:
            rr = json.loads(o.decode())
            agg += rr['sent_per_s_analyze']
            nsent += rr['sentences'] * 3
    wall = time.time() - t
    out['all_cores'] = {'value': round(agg, 1), 'unit': 'sentences/s', 'cores': ncore, 'visible_cpus': os.cpu_count(),
                        'what': 'one reference process per hardware thread, %d sentences each (3 passes), sum of the '
                                'per-process rates; %.1f s wall incl. %d model loads (wall-clock rate %.0f/s)'
                                % (nsent // (3 * max(1, ncore)), wall, ncore, nsent / wall)}
    return out


def realism_legs(args, cache, local_rank, np, torch, J):
    """Extra legs beside the headline (never `value`): the same step on (i) the headline workload of rounds 1-3 (300 k
    entries, 2^22 weights = 16 MB, cache resident: the friendly end), (ii) a weight table of 2^26 floats (256 MB: the
    size of the Infinity Cache), (iii) a homograph-heavy dictionary.  Every leg carries its own parity_sample."""
    import copy
    legs = {}
    dev = torch.device('cuda', local_rank)
    stream = torch.cuda.current_stream().cuda_stream
    # (iv, round 5) the headline model on a ZIPF corpus (word rank r drawn with weight 1/r): the word statistics of real
    # text -- most weight gathers repeat and hit the caches, the regime where k_sweep is bound by instruction issue
    for name, over in (('dict_300k_weights_2e22', {'dict_entries': 300000, 'weights_exp': 22}), ('weights_2e26', {'weights_exp': 26}),
                       ('zipf_corpus', {'zipf': 1.0})):
        try:
            a = copy.copy(args)
            for k, v in over.items():
                setattr(a, k, v)
            t = time.time()
            mdic, model, img = make_workload(a, cache)
            corpus = make_corpus(a, mdic, cache, args.batch * 2, a.seed + 1 + a.dict_entries % 7)
            batches = load_batches(corpus, args.batch, np)
            setup_s = time.time() - t
            ctx = J.Context(img, beam=5, global_beam=6, right_check=1, right_beam=5, device=local_rank,
                            use_rnn=None if args.rnn else False)
            d = [(torch.frombuffer(bytearray(tx), dtype=torch.uint8).to(dev),
                  torch.from_numpy(of.astype(np.int32)).to(dev), len(of) - 1, len(tx)) for tx, of in batches]

            def run(i):
                tt, oo, n, nbytes = d[i % len(d)]
                return ctx.analyze_device(tt.data_ptr(), oo.data_ptr(), n, nbytes, stream)
            run(0).release()
            run(1).release()   # both batches once: the workspaces have their final size before the clock starts
            torch.cuda.synchronize()
            k = 5
            km = {}
            per_step = []
            for i in range(k):
                t0 = time.perf_counter()
                r = run(1 + i)
                torch.cuda.synchronize()
                per_step.append(time.perf_counter() - t0)
                for kk, v in ctx.timings().items():
                    km[kk] = km.get(kk, 0.0) + v
                r.release()
            # median step: a fresh context stalls once for ~60 ms between two of its first calls (seen in the kernel
            # trace as one idle gap, profiles/r02_p_realism_gaps.txt), which a 5-step mean would carry as +12 ms per step
            el = sorted(per_step)[k // 2]
            r = run(0)
            par = None
            if not args.no_parity and not args.no_cpu_baseline:
                par = leg_parity(model, r, args.batch, batches[0][0], batches[0][1], 16384, [5, 6, 1, 5], np, torch, dev,
                                 os.path.join(cache, 'parity_tmp'), args.sent_len)
            r = r.fetch()
            legs[name] = {'value': round(args.batch / el, 1), 'unit': 'sentences/s', 'steps': k, 'timing': 'median step',
                          'ms_per_step': round(el * 1e3, 3),
                          'nodes_per_sentence': round(float(r.nnodes.sum()) / args.batch, 1),
                          'failed_sentences_in_batch': int((r.status != 0).sum()),
                          'kernel_ms_per_step': {kk: round(v / k, 3) for kk, v in km.items()},
                          'dict_entries': a.dict_entries, 'weights_exp': a.weights_exp, 'setup_s': round(setup_s, 1)}
            if par is not None:
                legs[name]['parity_sample'] = par
            r.release()
            del ctx, d
            torch.cuda.empty_cache()
        except Exception as e:  # an extra leg must never take the main line down
            legs[name] = {'error': str(e)[:200]}
    legs['homographs'] = homograph_leg(args, cache, local_rank, np, torch, J)
    return legs


def trainer_leg(args, mdic, cache, ge):
    """SURVEY 8 row f4 (trainer hook-up), never `value`: one epoch of `jumanpp_gpu_train` (examples analysed by the device in
    batches, gold nodes injected through the seed hook, loss + SCW update on the host) next to the reference's own
    `jumanpp_v2_train`, and -- the parity half -- the model FILE both write for --batch 1 on a prefix of the corpus.
    Corpus: sentences of the headline generator annotated by the reference with the headline dictionary and a
    random-weight teacher model (`jumanpp_v2 --full-morph` = the trainer's input format)."""
    try:
        train_cli = ge.TRAIN_CLI
        ge.build_host()
        dkey = os.path.basename(mdic)[:-5]
        seed_model = os.path.join(cache, dkey + '.seed')
        teacher = os.path.join(cache, dkey + '.teacher')
        if not os.path.exists(teacher):
            subprocess.check_call([os.path.join(REF, 'ref_dump'), 'mkmodel', seed_model, teacher + '.tmp', '20', '11', '0.1'])
            os.rename(teacher + '.tmp', teacher)
        n_ex = 8192
        corpus = os.path.join(cache, dkey + '_train%d.txt' % n_ex)
        if not os.path.exists(corpus):
            gen = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'gen_corpus.py'), mdic, str(n_ex + n_ex // 16), '--seed', '77',
                                  '--len', str(args.sent_len), '--oov', '0.05'], stdout=subprocess.PIPE, check=True).stdout.decode('utf-8')
            # (the Morph corpus format has no quoting: sentences with its separator characters cannot be written in it)
            lines = [l for l in gen.split('\n') if l and not any(c in l for c in ' _"#,')][:n_ex]
            nproc = max(1, min(16, len(os.sched_getaffinity(0))))
            parts = [lines[i::nproc] for i in range(nproc)]
            procs = []
            for i, part in enumerate(parts):
                pth = corpus + '.raw%d' % i
                with open(pth, 'w', encoding='utf-8') as f:
                    f.write('\n'.join(part) + '\n')
                procs.append((pth, subprocess.Popen([os.path.join(REF, 'jumanpp_v2'), '--model=' + teacher, '--full-morph', pth],
                                                    stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)))
            with open(corpus + '.tmp', 'w', encoding='utf-8') as f:
                for pth, pr in procs:
                    o = pr.communicate()[0].decode('utf-8')
                    os.remove(pth)
                    for l in o.split('\n'):
                        if l.strip():
                            f.write(l.rstrip(' ') + '\n')
            os.rename(corpus + '.tmp', corpus)
        gb = ['--gb-left-min=6', '--gb-left-max=6', '--gb-rcheck-min=1', '--gb-rcheck-max=1', '--gb-right-min=5', '--gb-right-max=5',
              '--size=%d' % args.weights_exp]
        tmp = tempfile.mkdtemp(prefix='jpptrain_')

        def run(cmd):
            t0 = time.perf_counter()
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            return time.perf_counter() - t0, p
        batch = 512
        wall, p = run([train_cli, '--model-input=' + seed_model, '--model-output=' + os.path.join(tmp, 'g.model'), '--corpus=' + corpus,
                       '--batch=%d' % batch] + gb)
        if p.returncode != 0:
            return {'error': p.stderr.decode()[-300:]}
        err = p.stderr.decode().strip().splitlines()
        stages = {}
        for part in err[-1].replace('stage ms: ', '').split(', '):
            k, v = part.rsplit(' ', 1)
            stages[k] = round(float(v), 1)
        epoch_ms = sum(stages.values())
        nthreads = max(1, min(16, len(os.sched_getaffinity(0))))
        ref_wall, _ = run([os.path.join(REF, 'jumanpp_v2_train'), '--model-input=' + seed_model, '--model-output=' + os.path.join(tmp, 'r.model'),
                           '--corpus=' + corpus, '--batch=%d' % (4 * nthreads), '--threads=%d' % nthreads] + gb)
        ref1_wall, _ = run([os.path.join(REF, 'jumanpp_v2_train'), '--model-input=' + seed_model, '--model-output=' + os.path.join(tmp, 'r1.model'),
                            '--corpus=' + corpus, '--batch=1', '--threads=1'] + gb)
        # parity: --batch 1 on the first 256 examples, both trainers, same bytes
        head = os.path.join(tmp, 'head.txt')
        with open(head, 'w', encoding='utf-8') as f:
            f.writelines(open(corpus, encoding='utf-8').readlines()[:256])
        run([os.path.join(REF, 'jumanpp_v2_train'), '--model-input=' + seed_model, '--model-output=' + os.path.join(tmp, 'rp.model'),
             '--corpus=' + head, '--batch=1', '--threads=1'] + gb)
        _, pg = run([train_cli, '--model-input=' + seed_model, '--model-output=' + os.path.join(tmp, 'gp.model'), '--corpus=' + head, '--batch=1'] + gb)
        same = pg.returncode == 0 and open(os.path.join(tmp, 'rp.model'), 'rb').read() == open(os.path.join(tmp, 'gp.model'), 'rb').read()
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
        return {'what': 'jumanpp_gpu_train, one epoch over %d annotated %d-codepoint sentences, --batch %d (one device pass per batch, '
                        'weights frozen inside a batch), global beam 6/1/5, 2^%d weights; examples/s over the epoch loop (stage sum), '
                        'process wall time apart' % (n_ex, args.sent_len, batch, args.weights_exp),
                'value': round(n_ex / (epoch_ms * 1e-3), 1), 'unit': 'examples/s', 'epoch_ms': round(epoch_ms, 1), 'stage_ms': stages,
                'device_share_ms': round(sum(v for k, v in stages.items() if k.startswith(('analyse', 'lattice', 'n-gram', 'weights'))), 1),
                'process_wall_s': round(wall, 2),
                'reference': {'what': 'jumanpp_v2_train, same corpus, whole process', 'threads_%d_wall_s' % nthreads: round(ref_wall, 2),
                              'threads_1_batch_1_wall_s': round(ref1_wall, 2)},
                'parity_batch1': {'examples': 256, 'model_files_identical': bool(same)}}
    except Exception as e:  # noqa: BLE001
        return {'error': repr(e)[:300]}


def homograph_leg(args, cache, local_rank, np, torch, J):
    """jumandic-like fan-out (tools/gen_dict.py --homographs 60: every single hiragana has 8..60 dictionary entries,
    150 two-kana surfaces 4..30): which sweep variants run and what they cost, (i) on ordinary text of that dictionary,
    (ii) on a batch without the fan-out surfaces, and (iii) on (ii) with ONE sentence made of the widest surfaces --
    every sentence runs the variant of its own widest boundary, so (iii) must cost what (ii) costs."""
    import copy
    try:
        a = copy.copy(args)
        a.homographs = 60
        mdic, model, img = make_workload(a, cache)
        # surfaces by number of entries; a dictionary view without the fan-out surfaces for corpus (ii)
        counts = {}
        rows = open(mdic, encoding='utf-8').read().split('\n')
        for r in rows[8:]:
            if r:
                counts[r.split(',', 1)[0]] = counts.get(r.split(',', 1)[0], 0) + 1
        # a dictionary view for corpus (ii): no fan-out surfaces, and no hiragana at all (every single hiragana is a
        # fan-out surface, so any hiragana in the text makes a wide boundary)
        def plain(surface):
            return counts[surface] <= 6 and not any('\u3041' <= ch <= '\u309f' for ch in surface)
        base = os.path.join(cache, os.path.basename(mdic) + '.narrow2')
        if not os.path.exists(base):
            with open(base, 'w', encoding='utf-8') as f:
                f.write('\n'.join(rows[:8] + [r for r in rows[8:] if r and plain(r.split(',', 1)[0])]) + '\n')
        wide = sorted(counts, key=lambda k: -counts[k])[:40]
        corpus_all = make_corpus(a, mdic, cache, args.batch, a.seed + 77)
        a_narrow = copy.copy(a)
        a_narrow.oov = 0.0   # (the OOV runs of the generator contain hiragana)
        corpus_narrow = make_corpus(a_narrow, base, cache, args.batch, a.seed + 79)
        dev = torch.device('cuda', local_rank)
        stream = torch.cuda.current_stream().cuda_stream
        ctx = J.Context(img, beam=5, global_beam=6, right_check=1, right_beam=5, device=local_rank,
                        use_rnn=None if args.rnn else False)

        def measure(lines, check=0):
            offs = np.zeros(len(lines) + 1, dtype=np.uint32)
            offs[1:] = np.cumsum([len(l) for l in lines])
            tx = b''.join(lines)
            t = torch.frombuffer(bytearray(tx), dtype=torch.uint8).to(dev)
            o = torch.from_numpy(offs.astype(np.int32)).to(dev)
            ctx.analyze_device(t.data_ptr(), o.data_ptr(), len(lines), len(tx), stream).release()
            torch.cuda.synchronize()
            steps, ms, cls = [], [0.0, 0.0, 0.0], None
            sweep = 0.0
            for _ in range(4):
                t0 = time.perf_counter()
                r = ctx.analyze_device(t.data_ptr(), o.data_ptr(), len(lines), len(tx), stream)
                torch.cuda.synchronize()
                steps.append(time.perf_counter() - t0)
                sc = ctx.sweep_classes()
                cls = sc['sentences']
                ms = [m + x / 4 for m, x in zip(ms, sc['ms'])]
                sweep += ctx.timings()['sweep'] / 4
                r.release()
            rr = ctx.analyze_device(t.data_ptr(), o.data_ptr(), len(lines), len(tx), stream)
            par = None
            if check and not args.no_parity and not args.no_cpu_baseline:
                par = leg_parity(model, rr, len(lines), tx, offs, check, [5, 6, 1, 5], np, torch, dev,
                                 os.path.join(cache, 'parity_tmp'), max(len(l) for l in lines))
            rr = rr.fetch()
            out = {'sentences_per_s': round(len(lines) / sorted(steps)[2], 1), 'sweep_ms': round(sweep, 3),
                   'sweep_ms_by_class_64_512_any': [round(x, 3) for x in ms], 'sentences_by_class': cls,
                   'nodes_per_sentence': round(float(rr.nnodes.sum()) / len(lines), 1),
                   'failed_sentences_in_batch': int((rr.status != 0).sum())}
            rr.release()
            if par is not None:
                out['parity_sample'] = par
            return out
        lines_all = open(corpus_all, 'rb').read().split(b'\n')[:args.batch]
        lines_narrow = open(corpus_narrow, 'rb').read().split(b'\n')[:args.batch]
        one_wide = list(lines_narrow)
        one_wide[len(one_wide) // 2] = ''.join(wide).encode('utf-8')[:3 * args.sent_len]
        return {'dictionary': '%d entries, --homographs 60 (max %d entries on one surface)' % (a.dict_entries, max(counts.values())),
                'ordinary_text': measure(lines_all, 8192), 'narrow_text': measure(lines_narrow),
                'narrow_text_plus_one_wide_sentence': measure(one_wide, len(one_wide) // 2 + 64)}
    except Exception as e:  # an extra leg must never take the main line down
        return {'error': str(e)[:300]}


def _timing_kv(stderr):
    """key=value tokens of every `--timing` line of jumanpp_gpu (the pipeline line, `reserve:`, `batches:`); the reserve
    line's own `ms` is kept as reserve_ms"""
    kv = {}
    for line in (stderr or '').strip().splitlines():
        pre = 'reserve_' if line.startswith('reserve:') else ''
        for tok in line.split():
            if '=' in tok:
                k, v = tok.split('=', 1)
                try:
                    kv[pre + k if pre and k == 'ms' else k] = float(v)
                except ValueError:
                    pass
    return kv


def _settle():
    """Between two child processes of the CLI legs.  A GPU process that starts within a second of another one's exit can
    lose 4-5 s anywhere -- inside its pipeline (a run at 0.3 M sentences/s with ten times the GPU time per batch) or in
    its start-up (a normal run with 5-6 s of process wall): 6 of 30 back-to-back runs in tools/gpu_cli_probe.py spread,
    with and without --clean-exit, no CPU throttling, no memory events (profiles/r06_r_*, r06_s_*); 0 of 6 after a 3 s
    pause.  The driver is still taking the previous process' queues and 40 GB of buffers apart.  A service is one
    process; the legs below are twenty, so they wait."""
    time.sleep(3.0)


def cli_end_to_end(args, model, corpus, n_lines, ge):
    """The product binary end to end, timed by this process: jumanpp_gpu (C++14 host pipeline above the C ABI)
    reads the corpus file, analyses it in 65,536-sentence batches and writes the JUMAN-format text to a file.
    Host input parsing, H2D, result fetch, formatting and output are all inside; model load is reported apart."""
    try:
        cli = ge.build_host()
        out_path = os.path.join(os.path.dirname(corpus), 'cli_out.txt')
        best = None
        main_rates = []
        for _ in range(3):   # a fresh process sometimes stalls for seconds in its first device allocations (seen as 2.5 s in one of
                             # four runs, tools/gpu_cli_loop.sh), and the 2 GB output goes to the box's scratch disk: best of three runs
            _settle()
            t0 = time.perf_counter()
            p = subprocess.run([cli, '--model=' + model, '--batch=%d' % args.batch, '--timing', '-o', out_path, corpus],
                               capture_output=True, text=True)
            wall = time.perf_counter() - t0
            if p.returncode != 0:
                return {'error': (p.stderr or '')[-200:]}
            kv = _timing_kv(p.stderr)
            size = os.path.getsize(out_path)
            os.remove(out_path)
            r = {'what': 'jumanpp_gpu --model=M.jppmdl corpus -o file: %d lines, file in -> JUMAN text out (%.0f MB), '
                         'sharded pipeline (mapped input cut at newlines | per device: split + analyse | JUMAN text printed by the '
                         'device (k_fmt_*), copied into page-locked blocks | pwritev); stage times are summed over the stage '
                         'threads ("format" = format kernels + the copy of the text); best of 3 runs' % (n_lines, size / 1e6),
                 'value': round(kv.get('sent_per_s', 0.0), 1), 'unit': 'sentences/s',
                 'pipeline_wall_ms': round(kv.get('wall_ms', 0.0), 1), 'gpu_busy_ms': round(kv.get('gpu_ms', 0.0), 1),
                 'stage_busy_ms': {k: round(kv.get(k + '_ms', 0.0), 1) for k in ('read', 'analyze', 'format', 'write')},
                 'process_wall_s_incl_model_load': round(wall, 2),
                 'sentences_per_s_incl_model_load': round(n_lines / wall, 1),
                 'batches': {k: int(kv.get(k, -1)) for k in ('one_enqueue', 'rerun', 'sized', 'device_allocations')},
                 'reserve_ms': round(kv.get('reserve_ms', 0.0), 1),
                 'format_kernels_ms': {'count': round(kv.get('format_count_gpu_ms', 0.0), 1), 'write': round(kv.get('format_write_gpu_ms', 0.0), 1)},
                 'process_ms': {'first_analyzers_ready': round(kv.get('first_analyzers_ready_ms', 0.0), 1),
                                'waited_for_page_locking': round(kv.get('waited_ms', 0.0), 1),
                                'before_teardown': round(kv.get('process_ms_before_teardown', 0.0), 1)}}
            main_rates.append(round(r['value']))
            if best is None or r['value'] > best['value']:
                best = r
        best['runs'] = main_rates
        def variant(flags, reps=3):
            """best of `reps` runs of the command with extra flags; every run's rate is kept (a run now and then spends
            seconds in its first device allocations / on the scratch disk, see above)"""
            top, kvt, rates = None, None, []
            for _ in range(reps):
                _settle()
                p = subprocess.run([cli, '--model=' + model, '--batch=%d' % args.batch, '--timing', '-o', out_path, corpus] + flags,
                                   capture_output=True, text=True)
                if p.returncode != 0:
                    return None, None, rates
                kv = _timing_kv(p.stderr)
                for f in [out_path] + glob.glob(out_path + '.part*'):
                    if os.path.exists(f):
                        os.remove(f)
                rates.append(round(kv.get('sent_per_s', 0.0)))
                if top is None or kv.get('sent_per_s', 0.0) > top:
                    top, kvt = kv.get('sent_per_s', 0.0), kv
            return top, kvt, rates
        # the same with the host formatters (rounds 1-3: --threads format workers with a per-entry text cache)
        top, kv, rates = variant(['--host-format'])
        if top is not None:
            best['host_format'] = {'what': 'the same run with --host-format (%d format threads), best of 3 runs' % int(kv.get('threads', 0)),
                                   'value': round(top, 1), 'unit': 'sentences/s', 'runs': rates,
                                   'pipeline_wall_ms': round(kv.get('wall_ms', 0.0), 1),
                                   'stage_busy_ms': {k: round(kv.get(k + '_ms', 0.0), 1) for k in ('read', 'analyze', 'format', 'write')}}
        # the multi-GPU form of the same command on this one-GPU box: the device list names the GPU twice, i.e. two
        # per-device pipelines (line splitter + analyzer pair + format workers + writer each) that share one GPU
        top, kv, rates = variant(['--devices=0,0'])
        if top is not None:
            best['devices_0_0'] = {'what': 'the same run with --devices=0,0 (two per-device pipelines on the one GPU), best of 3 runs',
                                   'value': round(top, 1), 'unit': 'sentences/s', 'runs': rates,
                                   'pipeline_wall_ms': round(kv.get('wall_ms', 0.0), 1), 'gpu_busy_ms': round(kv.get('gpu_ms', 0.0), 1),
                                   'stage_busy_ms': {k: round(kv.get(k + '_ms', 0.0), 1) for k in ('read', 'analyze', 'format', 'write')}}
        # ... and with the output in four files (--output-shards=4: the input in four contiguous parts, `cat OUT.part*` is the
        # one-file output).  Buffered writes to one file are serialised on its inode -- 14 GB/s on this box whatever the
        # number of writer threads, 28 / 55 / 98 GB/s into 2 / 4 / 8 files (tools/host_write_ceiling.py) -- which is what
        # an eight-GPU run of this command would otherwise be bound by.
        top, kv, rates = variant(['--output-shards=4'])
        if top is not None:
            best['output_shards_4'] = {'what': 'the same run with --output-shards=4 (four output files), best of 3 runs',
                                       'value': round(top, 1), 'unit': 'sentences/s', 'runs': rates,
                                       'pipeline_wall_ms': round(kv.get('wall_ms', 0.0), 1), 'gpu_busy_ms': round(kv.get('gpu_ms', 0.0), 1),
                                       'stage_busy_ms': {k: round(kv.get(k + '_ms', 0.0), 1) for k in ('read', 'analyze', 'format', 'write')}}
        # The same command on an input four times as long (the corpus file repeated): a 1 M-line run is 16 batches, of
        # which the first two run on fresh buffers and the pipeline fills and drains once -- this is what the binary
        # sustains (profiles/r03_x_cli_steady_state.txt).
        try:
            big = corpus + '.x4'
            if not os.path.exists(big):
                data = open(corpus, 'rb').read()
                with open(big + '.tmp', 'wb') as f:
                    for _ in range(4):
                        f.write(data)
                os.rename(big + '.tmp', big)
            ss = None
            for _ in range(2):
                _settle()
                p = subprocess.run([cli, '--model=' + model, '--batch=%d' % args.batch, '--timing', '-o', out_path, big], capture_output=True, text=True)
                if p.returncode != 0:
                    break
                kv = _timing_kv(p.stderr)
                size = os.path.getsize(out_path)
                os.remove(out_path)
                r = {'what': 'the same command on %d lines (%.1f GB of JUMAN text written), best of 2 runs' % (4 * n_lines, size / 1e9),
                     'value': round(kv.get('sent_per_s', 0.0), 1), 'unit': 'sentences/s', 'pipeline_wall_ms': round(kv.get('wall_ms', 0.0), 1),
                     'gpu_busy_ms': round(kv.get('gpu_ms', 0.0), 1),
                     'stage_busy_ms': {k: round(kv.get(k + '_ms', 0.0), 1) for k in ('read', 'analyze', 'format', 'write')}}
                if ss is None or r['value'] > ss['value']:
                    ss = r
            if ss is not None:
                best['steady_state'] = ss
        except OSError as e:
            best['steady_state'] = {'error': str(e)[:200]}
        return best
    except Exception as e:  # an extra measurement must never take the main line down
        return {'error': str(e)[:200]}


def config5_cli_lattice(args, cache, ge, np):
    """BASELINE configs[4] as it is stated -- "beam=32 long-sentence stress, RNNLM on, LATTICE-FORMAT output" -- through the
    product binary: jumanpp_gpu --beam=32 --global-beam=32 --right-beam=32 -s 32, file in, lattice text out, on the leg's
    own corpus (2 x 16,384 sentences of 220 codepoints); and the checker: the first 2,048 lattice blocks against the
    reference CLI.  The reference prints, for a node with several connections, the one std::max_element finds in a
    FlatSet hashed by HOST ADDRESS (lattice_format.cc:133-141): on exact score ties two runs of jumanpp_v2 itself print
    different lines, so the reference runs twice and a block counts as a mismatch only if it differs from BOTH runs while
    they agree with each other (tools/gpu_config5.py, rounds 2-4)."""
    import copy
    import re
    try:
        a = copy.copy(args)
        a.sent_len = 220
        batch = int(getattr(args, 'config5_batch', 16384))
        mdic, model, img = make_workload(a, cache)
        corpus = make_corpus(a, mdic, cache, batch * 2, 31)
        cli = ge.build_host()
        out_path = os.path.join(cache, 'c5_lattice_out.txt')
        flags = ['--beam=32', '--global-beam=32', '--right-beam=32', '-s', '32']
        best, rates = None, []
        for _ in range(2):
            # (no --batch: for lattice output the CLI sizes its batches itself -- a batch's text near 100 MB, 2 048
            # sentences here; with the host printer: the gathered N best paths below 1 GB, 3 072 sentences)
            if os.path.exists(out_path):
                # a NEW file every run, like the other leg: the second of two runs into the same path ran at half the
                # rate in every session of round 6 (135 k / 61 k) -- truncating 1.6 GB of dirty pages and rewriting the
                # file makes ext4 write it out synchronously when it is closed (replace-via-truncate), inside the CLI's clock
                os.remove(out_path)
            _settle()
            t0 = time.perf_counter()
            p = subprocess.run([cli, '--model=' + model, '--timing', '-o', out_path] + flags + [corpus],
                               capture_output=True, text=True)
            wall = time.perf_counter() - t0
            if p.returncode != 0:
                return {'error': (p.stderr or '')[-200:]}
            kv = _timing_kv(p.stderr)
            rates.append(round(kv.get('sent_per_s', 0.0)))
            if best is None or kv.get('sent_per_s', 0.0) > best[0]:
                best = (kv.get('sent_per_s', 0.0), kv, wall)
        rate, kv, wall = best
        size = os.path.getsize(out_path)
        device_text = 'device_lattice_format=1' in (p.stderr or '')
        res = {'what': 'jumanpp_gpu %s corpus -o file: %d sentences x 220 codepoints, N-best lattice format written '
                       '(%.0f MB); %s; batches of %d sentences (the CLI\'s choice for lattice output); best of 2 runs'
                       % (' '.join(flags), 2 * batch, size / 1e6,
                          'the lattice text is printed by the device (k_lat_count / k_lat_write: per-node path sets, ids, "%g" scores; '
                          'entry-row columns from the per-model table) and only text crosses PCIe' if device_text else
                          'the 32 best paths are gathered on the device (k_nbest), the lattice text is printed by the host format workers',
                          int(kv.get('batch_lines', 0))),
               'device_text': device_text,
               'format_kernels': {'count_ms': round(kv.get('format_count_gpu_ms', 0.0), 1), 'write_ms': round(kv.get('format_write_gpu_ms', 0.0), 1),
                                  'text_GB_per_s_of_the_write_pass': round(size / 1e9 / (kv.get('format_write_gpu_ms', 0.0) / 1e3), 1) if kv.get('format_write_gpu_ms', 0.0) > 0 else None,
                                  'what': 'HIP events around k_lat_count (+ offset scan) and k_lat_write, summed over the batches of the '
                                          'run (the two pipelines of the GPU overlap: a kernel of one waits for the other\'s sweep)'},
               'value': round(rate, 1), 'unit': 'sentences/s', 'runs': rates, 'pipeline_wall_ms': round(kv.get('wall_ms', 0.0), 1),
               'gpu_busy_ms': round(kv.get('gpu_ms', 0.0), 1),
               'stage_busy_ms': {k: round(kv.get(k + '_ms', 0.0), 1) for k in ('read', 'analyze', 'format', 'write')},
               'process_wall_s_incl_model_load': round(wall, 2),
               'process_ms': {'first_analyzers_ready': round(kv.get('first_analyzers_ready_ms', 0.0), 1),
                                'waited_for_page_locking': round(kv.get('waited_ms', 0.0), 1),
                              'reserve': round(kv.get('reserve_ms', 0.0), 1),
                              'before_teardown': round(kv.get('process_ms_before_teardown', 0.0), 1)},
               'batches': {k: int(kv.get(k, -1)) for k in ('one_enqueue', 'rerun', 'sized', 'device_allocations')}}
        if device_text:   # the round-5 form of the same command: N best paths gathered on the device, text printed by the host workers
            _settle()
            ph = subprocess.run([cli, '--model=' + model, '--timing', '--host-format', '-o', out_path + '.host'] + flags + [corpus],
                                capture_output=True, text=True)
            if ph.returncode == 0:
                kh = _timing_kv(ph.stderr)
                res['host_format'] = {'what': 'the same run with --host-format (k_nbest + %d host format threads)' % int(kh.get('threads', 0)),
                                      'value': round(kh.get('sent_per_s', 0.0), 1), 'unit': 'sentences/s',
                                      'pipeline_wall_ms': round(kh.get('wall_ms', 0.0), 1),
                                      'stage_busy_ms': {k: round(kh.get(k + '_ms', 0.0), 1) for k in ('read', 'analyze', 'format', 'write')},
                                      'same_bytes_as_device_text': os.path.getsize(out_path + '.host') == size and
                                      subprocess.run(['cmp', '-s', out_path, out_path + '.host']).returncode == 0}
            if os.path.exists(out_path + '.host'):
                os.remove(out_path + '.host')
            # the device-text run again with the output in four files: this leg writes 48 KB per sentence and sits on the
            # one-file write ceiling of the host (see cli_end_to_end.output_shards_4)
            _settle()
            ps = subprocess.run([cli, '--model=' + model, '--timing', '--output-shards=4', '-o', out_path + '.sh'] + flags + [corpus],
                                capture_output=True, text=True)
            if ps.returncode == 0:
                ks = _timing_kv(ps.stderr)
                shard_files = sorted(glob.glob(out_path + '.sh.part*'))
                res['output_shards_4'] = {'what': 'the same run with --output-shards=4',
                                          'value': round(ks.get('sent_per_s', 0.0), 1), 'unit': 'sentences/s',
                                          'pipeline_wall_ms': round(ks.get('wall_ms', 0.0), 1),
                                          'stage_busy_ms': {k: round(ks.get(k + '_ms', 0.0), 1) for k in ('read', 'analyze', 'format', 'write')},
                                          'bytes_equal_the_one_file_output': sum(os.path.getsize(f) for f in shard_files) == size}
                for f in shard_files:
                    os.remove(f)
        if not args.no_parity and not args.no_cpu_baseline:
            t = time.time()
            n_check = 2048
            lines = []
            with open(corpus, 'rb') as f:
                for i, line in enumerate(f):
                    if i >= n_check:
                        break
                    lines.append(line)
            ref_dir = reference_build()[0]
            procs = max(1, min(usable_cores(), 16))
            per = (n_check + procs - 1) // procs
            tmp = os.path.join(cache, 'parity_tmp')
            os.makedirs(tmp, exist_ok=True)

            def ref_run(tag):
                running = []
                for k in range(procs):
                    part = os.path.join(tmp, 'c5lat_%s_%d.txt' % (tag, k))
                    with open(part, 'wb') as f:
                        f.writelines(lines[k * per:(k + 1) * per])
                    running.append((subprocess.Popen([os.path.join(ref_dir, 'jumanpp_v2'), '--model=' + model] + flags + [part],
                                                     stdout=subprocess.PIPE, stderr=subprocess.DEVNULL), part))
                blocks = []
                for pr, part in running:
                    out, _ = pr.communicate()
                    os.remove(part)
                    if pr.returncode != 0:
                        raise RuntimeError('jumanpp_v2 failed (rc %d)' % pr.returncode)
                    b = out.split(b'EOS\n')
                    blocks += b[:-1]
                return blocks
            r1, r2 = ref_run('a'), ref_run('b')
            with open(out_path, 'rb') as f:
                ours = []
                buf = b''
                while len(ours) < n_check:
                    chunk = f.read(1 << 24)
                    if not chunk:
                        break
                    buf += chunk
                    parts = buf.split(b'EOS\n')
                    buf = parts.pop()
                    ours += parts
                ours = ours[:n_check]

            # What a lattice line prints for a node is the score triple of ONE of its connections on the N best paths: the
            # reference takes std::max_element over a FlatSet hashed by host address (lattice_format.cc:133-141), so among
            # connections whose weighted totals tie in the comparator the printed one -- and with it the split
            # "特徴量スコア | 言語モデルスコア | 形態素解析スコア", which differs between such connections from the 6th digit
            # on -- changes from run to run of jumanpp_v2 itself.  The check therefore has two halves: every block with
            # those three numbers masked must be byte-identical (ids, predecessor lists, spans, strings, features, rank
            # lists, the "# MA-SCORE" line with the N-best totals), and the masked numbers must agree within 1e-4.
            pat = re.compile('(特徴量スコア:|言語モデルスコア:|形態素解析スコア:)(-?[0-9.e+-]+)'.encode('utf-8'))

            def masked(block):
                return pat.sub(lambda m: m.group(1) + b'#', block)

            def numbers(block):
                return [float(m.group(2)) for m in pat.finditer(block)]
            n = min(len(ours), len(r1), len(r2))
            unstable = [i for i in range(n) if r1[i] != r2[i]]
            differing = [i for i in range(n) if ours[i] != r1[i]]
            structural, worst = [], 0.0
            for i in differing:
                if masked(ours[i]) != masked(r1[i]):
                    structural.append(i)
                    continue
                xa, xb = numbers(ours[i]), numbers(r1[i])
                d = max([abs(x - y) for x, y in zip(xa, xb)] or [0.0]) if len(xa) == len(xb) else 1.0
                worst = max(worst, d)
                if d > 1e-4:
                    structural.append(i)
            ref_self = 0.0
            for i in unstable:
                xa, xb = numbers(r1[i]), numbers(r2[i])
                if len(xa) == len(xb) and masked(r1[i]) == masked(r2[i]):
                    ref_self = max(ref_self, max([abs(x - y) for x, y in zip(xa, xb)] or [0.0]))
            res['parity_sample'] = {
                'blocks': n, 'byte_identical_to_reference_run_1': n - len(differing),
                'identical_with_the_three_score_fields_masked': n - len([i for i in differing if masked(ours[i]) != masked(r1[i])]),
                'max_difference_of_a_masked_score_field': worst,
                'reference_run_1_vs_run_2': {'differing_blocks': len(unstable), 'max_difference_of_a_masked_score_field': ref_self},
                'mismatches': len(structural) + (n_check - n), 'first_mismatches': structural[:8],
                'what': 'lattice (-s 32) blocks of the first %d sentences vs jumanpp_v2 (%d processes; run twice).  A block is a '
                        'mismatch when it differs with the per-connection score triple masked, or a masked number differs by more '
                        'than 1e-4: which of several exactly tied connections of a node lends its triple to the line depends on '
                        'host addresses in the reference (lattice_format.cc:133-141) -- its own two runs differ the same way; %.1f s'
                        % (n_check, procs, time.time() - t)}
        os.remove(out_path)
        return res
    except Exception as e:  # an extra measurement must never take the main line down
        return {'error': str(e)[:300]}


def config5_leg(args, cache, local_rank, np, torch, J):
    """BASELINE configs[4] on one GPU (never `value`): beam = global beam = right beam = 32, right-check 1,
    220-codepoint sentences, RNNLM on, 16,384 sentences per batch (4,096 until round 3: one wavefront per 1 100-node
    sentence needs more sentences than that to fill the chip, profiles/r03_n_config5_batches.txt), same model as the headline.  Reports the
    device-resident rate and the roofline of its dominant kernel (k_sweep<32, *>); the algorithmic bytes come
    from the fully fetched lattice of the first 256 sentences, scaled by the node count."""
    import copy
    try:
        a = copy.copy(args)
        a.sent_len = 220
        batch = int(getattr(args, 'config5_batch', 16384))
        mdic, model, img = make_workload(a, cache)
        corpus = make_corpus(a, mdic, cache, batch * 2, 31)
        batches = load_batches(corpus, batch, np)
        dev = torch.device('cuda', local_rank)
        stream = torch.cuda.current_stream().cuda_stream
        ctx = J.Context(img, beam=32, global_beam=32, right_check=1, right_beam=32, device=local_rank,
                        use_rnn=None if args.rnn else False)
        d = [(torch.frombuffer(bytearray(tx), dtype=torch.uint8).to(dev),
              torch.from_numpy(of.astype(np.int32)).to(dev), len(of) - 1, len(tx)) for tx, of in batches]

        def run(i):
            tt, oo, n, nbytes = d[i % len(d)]
            return ctx.analyze_device(tt.data_ptr(), oo.data_ptr(), n, nbytes, stream)
        run(0).release()
        run(1).release()
        torch.cuda.synchronize()
        k = 4
        km = {}
        t0 = time.perf_counter()
        for i in range(k):
            r = run(i)
            torch.cuda.synchronize()
            for kk, v in ctx.timings().items():
                km[kk] = km.get(kk, 0.0) + v / k
            r.release()
        el = time.perf_counter() - t0
        r = run(0)
        par = None
        if not args.no_parity and not args.no_cpu_baseline:
            par = leg_parity(model, r, batch, batches[0][0], batches[0][1], batch, [32, 32, 1, 32], np, torch, dev,
                             os.path.join(cache, 'parity_tmp'), 220)
        r = r.fetch()
        nodes = float(r.nnodes.sum())
        bad = int((r.status != 0).sum())
        r.release()
        # algorithmic bytes of k_sweep from a fully fetched sub-batch
        sub = 256
        lines = open(corpus, 'rb').read().split(b'\n')[:sub]
        rs = ctx.analyze(lines).fetch(full=True)
        ab = algorithmic_bytes(rs, 32, 32, 1, 32, np)
        rs.release()
        sweep_bytes = ab['sweep'] * nodes / max(1.0, ab['nodes'])
        ach = sweep_bytes / (km['sweep'] * 1e-3) / 1e9
        # the RNN block of this shape: bytes by SURVEY 8(d), flops of the recurrence (2 E^2 per rnn node); the recurrence
        # of these long sentences runs in k_rnn_chain like everyone's (row records), their scores in k_rnn_score_long
        rnn_roof = None
        if args.rnn and km.get('rnn', 0) > 0:
            ctx.analyze_device(d[0][0].data_ptr(), d[0][1].data_ptr(), d[0][2], d[0][3], stream).release()
            st = ctx.rnn_stats()
            n_rnn = max(0, st['rows'] - 2 * batch)
            E = args.rnn_hidden
            rbytes = n_rnn * (2 * E * 4 + 12) + n_rnn * E * 4 * 2
            fl = n_rnn * 2.0 * E * E
            rnn_roof = {'bound': 'hbm', 'kernels': 'k_rnn_paths + k_rnn_prep + k_rnn_dense + k_rnn_order_* + k_rnn_chain + k_rnn_score_long',
                        'achieved': round(rbytes / (km['rnn'] * 1e-3) / 1e9, 2), 'peak': 8000.0, 'unit': 'GB/s',
                        'frac': round(rbytes / (km['rnn'] * 1e-3) / 1e9 / 8000.0, 5), 'algorithmic_bytes_per_step': int(rbytes),
                        'ms_per_step': round(km['rnn'], 3), 'rnn_nodes_per_sentence': round(n_rnn / batch, 1),
                        'recurrence_tflops_if_all_of_it': round(fl / (km['rnn'] * 1e-3) / 1e12, 2)}
        return {'workload': 'BASELINE configs[4] shape, one GPU: beam=gbeam=rbeam=32 rcheck=1, %d sentences x 220 codepoints '
                            'per step, perceptron + RNNLM' % batch,
                'value': round(batch * k / el, 1), 'unit': 'sentences/s', 'steps': k, 'ms_per_step': round(el / k * 1e3, 3),
                'nodes_per_sentence': round(nodes / batch, 1), 'failed_sentences_in_batch': bad,
                'kernel_ms_per_step': {kk: round(v, 3) for kk, v in km.items()},
                'roofline': {'bound': 'hbm', 'kernel': 'k_sweep<32,*>', 'achieved': round(ach, 2), 'peak': 8000.0, 'unit': 'GB/s',
                             'frac': round(ach / 8000.0, 5), 'algorithmic_bytes_per_launch': int(sweep_bytes),
                             'avg_launch_ms': round(km['sweep'], 3)},
                'roofline_rnn': rnn_roof,
                'batches': ctx.stats(),
                'parity_sample': par}
    except Exception as e:  # an extra leg must never take the main line down
        return {'error': str(e)[:200]}


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=16)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=65536)
    ap.add_argument('--sent-len', type=int, default=40)
    ap.add_argument('--dict-entries', type=int, default=1000000,
                    help='rows of the synthetic jumandic-layout dictionary (SURVEY 8(d): 5e5..1e6; 300000 = the headline of rounds 1-3, now the realism leg dict_300k_weights_2e22)')
    ap.add_argument('--weights-exp', type=int, default=24,
                    help='log2 of the perceptron table (24 = 64 MB: beyond the aggregate L2; 22 until round 3)')
    ap.add_argument('--seed', type=int, default=20260925)
    ap.add_argument('--cpu-sample', type=int, default=20000)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity', action='store_true', help='skip parity_sample (the reference run on the host cores)')
    ap.add_argument('--parity-batches', type=int, default=2,
                    help='timed batches checked sentence by sentence against the reference (parity_sample)')
    ap.add_argument('--no-overlap', action='store_true', help='skip the extra two-batches-in-flight measurement')
    ap.add_argument('--no-config5', action='store_true',
                    help="skip the BASELINE configs[4] leg (beam 32, 220-codepoint sentences, one GPU's share)")
    ap.add_argument('--config5-batch', type=int, default=16384, help='sentences per step of the configs[4]-shape leg')
    ap.add_argument('--no-trainer', action='store_true', help='skip the trainer leg (jumanpp_gpu_train vs jumanpp_v2_train)')
    ap.add_argument('--no-cli', action='store_true', help='skip the end-to-end jumanpp_gpu run (file in, JUMAN text out)')
    ap.add_argument('--no-realism', action='store_true',
                    help='skip the extra workload legs (1M-entry dictionary, 2^24 and 2^26 weights; SURVEY 8(d))')
    ap.add_argument('--rnn', dest='rnn', action='store_true', default=True,
                    help='BASELINE configs[2] (default; the metric is quoted on jumandic+RNNLM): perceptron + RNNLM re-ranker')
    ap.add_argument('--no-rnn', dest='rnn', action='store_false', help='BASELINE configs[1]: perceptron only')
    ap.add_argument('--traffic-profile', default=os.path.join(ROOT, 'profiles', 'traffic.json'),
                    help='per-kernel HBM bytes per launch from the committed rocprofv3 --pmc passes of this command')
    ap.add_argument('--counters-profile', default=os.path.join(ROOT, 'profiles', 'counters.json'),
                    help='per-kernel SQ counters per launch from the committed rocprofv3 --pmc passes (tools/gpu_session.sh valu)')
    ap.add_argument('--rnn-hidden', type=int, default=128)
    ap.add_argument('--rnn-vocab', type=int, default=30000)
    ap.add_argument('--cache', default=os.path.join(tempfile.gettempdir(), 'jppgpu_bench_cache'))
    return ap


def main():
    args = build_parser().parse_args()

    import numpy as np
    import torch
    import __graft_entry__ as ge
    if os.environ.get('JPPGPU_BENCH_EMU') != '1':
        ge.build_native()
    import jumanpp_amd as J

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    # TEST ONLY (tests/test_dist_cpu.py): JPPGPU_BENCH_EMU=1 walks this file's control flow -- the rank-0 model build
    # behind barriers, the sharded corpus, the gather inside the timed loop, the max over ranks, rank 0 certifying and timing
    # the CPU while the others wait -- on the kernel emulator with the gloo backend, at toy sizes.  The line it prints says
    # so and measures nothing; without the variable there is no CPU path.
    emu = os.environ.get('JPPGPU_BENCH_EMU') == '1'
    if not emu and not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback)')
    lib_path = ge.build_emu() if emu else None
    if not emu:
        torch.cuda.set_device(local_rank)

    def _sync():
        if not emu:
            torch.cuda.synchronize()
    dist = None
    # (JPPGPU_BENCH_DIST1=1, tests/test_dist_gpu.py: the N > 1 control flow -- barriers, the gather inside the timed loop,
    # the max over ranks -- through the RCCL backend with ONE rank, which is all a one-GPU box can run of it)
    if world > 1 or os.environ.get('JPPGPU_BENCH_DIST1') == '1':
        import torch.distributed as dist
        os.environ.setdefault('NCCL_DEBUG_FILE', '/dev/stderr')   # (RCCL's version banner: not on stdout, see the end of main)
        dist.init_process_group('gloo' if emu else 'nccl')
    cache = args.cache
    # one model for all ranks: rank 0 builds it (bootstrap + embedding + export on the host cores), the others wait
    if dist is not None and rank != 0:
        dist.barrier()
    mdic, model, img = make_workload(args, cache)
    if dist is not None and rank == 0:
        dist.barrier()
    n_batches = min(16, args.steps + args.warmup)
    # sentences shard embarrassingly: every rank analyses its own distinct lines (weak scaling)
    corpus = make_corpus(args, mdic, cache, args.batch * n_batches, args.seed + 1 + rank)
    batches = load_batches(corpus, args.batch, np)
    log('[rank %d] workload ready: %d batches of %d sentences' % (rank, len(batches), args.batch))

    # The CLI leg comes FIRST, before this process has allocated anything on the device.  `jumanpp_gpu` is a program of
    # its own, and what it does next to a process that holds -- or has just released -- tens of GB of device memory is
    # not its speed: the first child after such a change spends 2-4 s more on the GPU side, and after the legs below have
    # freed their contexts eight runs in a row do (profiles/r04_ah_cli_probe2.txt, the `runs` lists of profiles/r04_t_*).
    cli_result = None
    if not args.no_cli and world == 1 and not emu:
        cli_result = cli_end_to_end(args, model, corpus, args.batch * len(batches), ge)
    c5_cli = None
    if not args.no_config5 and world == 1 and not emu:
        c5_cli = config5_cli_lattice(args, cache, ge, np)

    ctx = J.Context(img, beam=5, global_beam=6, right_check=1, right_beam=5, device=0 if emu else local_rank, lib_path=lib_path)
    dev = torch.device('cpu') if emu else torch.device('cuda', local_rank)
    d_batches = []
    for text, offs in batches:
        t = torch.frombuffer(bytearray(text), dtype=torch.uint8).to(dev)
        o = torch.from_numpy(offs.astype(np.int32).view(np.int32)).to(dev)
        d_batches.append((t, o, len(offs) - 1, len(text)))
    stream = None if emu else torch.cuda.current_stream().cuda_stream
    # packed top-1 output (8 B / morpheme) -- what a CLI would format, and what is gathered across GPUs
    cap_items = args.batch * (args.sent_len + 1)
    d_offs = torch.zeros(args.batch + 1, dtype=torch.int32, device=dev)
    d_items = torch.zeros((cap_items, 2), dtype=torch.int32, device=dev)
    from jumanpp_amd.dist import gather_packed_fixed

    def step(i):
        t, o, n, nbytes = d_batches[i % len(d_batches)]
        r = ctx.analyze_device(t.data_ptr(), o.data_ptr(), n, nbytes, stream)
        return r

    # What is gathered per step is a fixed shape on every rank (no size crosses to the host first): the offsets and the
    # items array up to the largest morpheme count any batch of any rank packs -- found here, untimed, by packing every
    # batch once (the bound sent_len + 1 per sentence is 1.6 x that: 21 MB instead of 13 MB per rank and step).
    gather_items = cap_items
    if dist is not None:
        most = 0
        for i in range(len(d_batches)):
            r = step(i)
            r.pack(d_offs.data_ptr(), d_items.data_ptr(), cap_items)
            most = max(most, int(d_offs[-1].item()))
            r.release()
        tm = torch.tensor([most], dtype=torch.int64, device=dev)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        gather_items = min(cap_items, (int(tm.item()) + 4095) // 4096 * 4096)
    for i in range(args.warmup):
        r = step(i)
        if dist is not None:
            # the gather's first call sets up the point-to-point connections between the ranks (RCCL does that lazily,
            # hundreds of milliseconds): part of the warm-up, like the kernels' first launches
            r.pack(d_offs.data_ptr(), d_items.data_ptr(), cap_items)
            gather_packed_fixed(d_offs, d_items[:gather_items], dst=0)
        r.release()
    _sync()
    if dist is not None:
        dist.barrier()
    _sync()
    t0 = time.perf_counter()
    kernel_ms = {}
    total_path = 0
    for i in range(args.steps):
        r = step(args.warmup + i)
        r.pack(d_offs.data_ptr(), d_items.data_ptr(), cap_items)
        if dist is not None:
            got = gather_packed_fixed(d_offs, d_items[:gather_items], dst=0)   # RCCL: one fixed-shape gather to rank 0, no host sync before it
            if rank == 0:
                total_path += int(torch.stack([g[0][-1] for g in got]).sum().item())  # one read-back (observes completion)
        else:
            total_path += int(d_offs[-1].item())          # forces completion of the batch
        for k, v in ctx.timings().items():
            kernel_ms[k] = kernel_ms.get(k, 0.0) + v
        r.release()
    _sync()
    if dist is not None:
        dist.barrier()
    _sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    sentences = args.batch * args.steps * world
    value = sentences / elapsed

    out = None
    if rank == 0:
        # algorithmic bytes of one representative batch (untimed)
        r = step(args.warmup).fetch(full=True)
        ab = algorithmic_bytes(r, 5, 6, 1, 5, np)
        bad = int((r.status != 0).sum())
        r_front = r
        # self-certification of the line (untimed; the reference is the checker, never the thing measured)
        parity = None
        if not args.no_cpu_baseline and not args.no_parity:   # (world > 1: rank 0 certifies its own shard)
            try:
                def run_packed(i):
                    rr = step(i)
                    rr.pack(d_offs.data_ptr(), d_items.data_ptr(), cap_items)
                    _sync()
                    ho = d_offs.cpu().numpy().view(np.uint32)
                    hi = d_items[:int(ho[-1])].cpu().numpy()
                    rr.release()
                    return ho, hi
                # (the timed loop starts at batch index `warmup`)
                order = [(args.warmup + j) % len(batches) for j in range(len(batches))]
                parity = parity_sample(reference_build()[0], model, [batches[j] for j in order],
                                       lambda i: run_packed(order[i]), np, os.path.join(cache, 'parity_tmp'),
                                       n_batches=args.parity_batches)
            except Exception as e:  # the checker must never take the main line down -- but it must say so
                parity = {'error': str(e)[:300]}
        avg = {k: v / args.steps for k, v in kernel_ms.items()}
        pipe_stats = ctx.stats()   # (of the timed context: warm-up batch 0 sized the buffers, the rest are one enqueue each)
        dom = 'sweep' if avg['sweep'] >= avg['t0'] else 't0'  # k_rnn has its own line in kernel_ms_per_step
        achieved = ab[dom] / (avg[dom] * 1e-3) / 1e9 if avg[dom] > 0 else 0.0
        # HBM traffic of the dominant kernel: measured offline (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
        # of this same command, tools/gpu_session_*.sh), committed under profiles/ -- reported only when it was
        # collected with the kernel sources that are running now
        traffic, traffic_note = None, None
        src_id = kernel_source_id()
        tp, traffic_note = load_profile(args.traffic_profile, args, src_id)
        if tp is not None:
            ent = tp['kernels'].get('k_' + dom) or tp['kernels'].get('k_' + dom + '_memo')
            if ent:
                traffic = ent['hbm_bytes_per_launch']
        else:
            tp = {}
        # what actually bounds k_sweep on this workload: the rate of L2 MISSES (every one a 128-byte request to the fabric)
        gather_roof = None
        ent = tp.get('kernels', {}).get('k_sweep')
        if ent and avg['sweep'] > 0:
            lines = 2 * ent['FETCH_SIZE_KB'] * 1024 / 128.0
            rate = lines / (avg['sweep'] * 1e-3)
            gather_roof = {'bound': 'l2-miss', 'kernel': 'k_sweep', 'achieved': round(rate / 1e9, 2), 'peak': GATHER_MISS_PEAK / 1e9,
                           'unit': 'G 128-byte lines/s', 'frac': round(rate / GATHER_MISS_PEAK, 4), 'lines_per_launch': int(lines),
                           'read_bytes_per_launch': int(lines * 128), 'read_tb_per_s': round(lines * 128 / (avg['sweep'] * 1e-3) / 1e12, 2),
                           'what': 'L2 misses of the launch (FETCH_SIZE: TCC_EA0_RDREQ, all of them 128-byte requests) / launch time; '
                                   'peak = the chip\'s measured rate of random 4-byte gather misses from a 64 MB table, '
                                   'profiles/r04_c_gather_policy.txt (no cache policy or allocation flavour changes it)'}
        cp, counters_note = load_profile(args.counters_profile, args, src_id)
        valu_roof = valu_roofline(cp['kernels'], 'k_sweep<8, 64', avg['sweep'], ab.get('sweep_gathers')) if cp else None
        if valu_roof is None:
            valu_roof = {'note': counters_note}
        # RNN block: bytes by SURVEY 8(d) (n_rnn (2 E 4 + 12) embedding rows and ids + n_ctx E 4 x 2 hidden states written and
        # read), flops of the recurrence (2 E^2 per rnn node) against the dense f32 MFMA peak for k_rnn_chain alone
        rnn_roof = None
        if args.rnn and avg.get('rnn', 0) > 0:
            st = ctx.rnn_stats()
            n_rnn = max(0, st['rows'] - 2 * args.batch)
            E = args.rnn_hidden
            rbytes = n_rnn * (2 * E * 4 + 12) + n_rnn * E * 4 * 2
            rnn_roof = {'bound': 'hbm', 'kernels': 'k_rnn_paths + k_rnn_prep + k_rnn_dense + k_rnn_order_* + k_rnn_chain + k_rnn_score',
                        'achieved': round(rbytes / (avg['rnn'] * 1e-3) / 1e9, 2), 'peak': 8000.0, 'unit': 'GB/s',
                        'frac': round(rbytes / (avg['rnn'] * 1e-3) / 1e9 / 8000.0, 5), 'algorithmic_bytes_per_step': int(rbytes),
                        'ms_per_step': round(avg['rnn'], 3), 'rnn_nodes_per_sentence': round(n_rnn / args.batch, 2)}
            if st['chain_ms'] > 0:
                fl = n_rnn * 2.0 * E * E
                rnn_roof['chain_mfma'] = {'bound': 'mfma', 'kernel': 'k_rnn_chain', 'achieved': round(fl / (st['chain_ms'] * 1e-3) / 1e12, 2),
                                          'peak': F32_MFMA_PEAK / 1e12, 'unit': 'TFLOP/s',
                                          'frac': round(fl / (st['chain_ms'] * 1e-3) / F32_MFMA_PEAK, 4), 'ms': round(st['chain_ms'], 3),
                                          'what': 'v_mfma_f32_16x16x4_f32, 2 E^2 flops per rnn node; 16 of the 16 columns of a tile are '
                                                  'sentences in lock step'}
        # the contract's byte count charges the 37 bigram gathers of (kept right node, T1 row 0) twice, as the reference
        # performs them (prescore + tail); the kernel gathers them once and forms both sums from them
        issued = ab.get('sweep_issued', ab['sweep'])
        # front end (decode + seeds + normalize + layout: everything before T0)
        front = None
        try:
            sample = [batches[args.warmup % len(batches)][0][int(o0):int(o1)] for o0, o1 in
                      zip(batches[args.warmup % len(batches)][1][:384], batches[args.warmup % len(batches)][1][1:385])]
            fb = front_end_bytes(img, r_front, sample, np)
            front_ms = avg['decode'] + avg['seeds'] + avg['layout']
            fbytes = fb['bytes_per_sentence'] * args.batch
            front = {'bound': 'hbm', 'kernels': 'k_decode + k_seeds<0/1/2> + k_norm<0/1/2> + k_layout / k_scan / k_connect / k_relocate / k_ends',
                     'achieved': round(fbytes / (front_ms * 1e-3) / 1e9, 2), 'peak': 8000.0, 'unit': 'GB/s',
                     'frac': round(fbytes / (front_ms * 1e-3) / 1e9 / 8000.0, 5), 'traffic': None,
                     'algorithmic_bytes_per_step': int(fbytes), 'ms_per_step': round(front_ms, 3),
                     'trie_units_per_sentence': round(fb['trie_units'], 1),
                     'entry_pointer_bytes_per_sentence': round(fb['entry_pointer_bytes'], 1),
                     'what': 'SURVEY 8(d): 4 B per double-array unit touched + entry-pointer list bytes (host walk of the '
                             "model's trie over 384 sentences of the timed batch) + sentence bytes + the arrays the front end "
                             'writes; since round 5 the phase has no host wait inside (one enqueue per batch, the '
                             'capacity guards run on the device)'}
            try:
                tpk = tp['kernels'] if tp.get('kernel_source_id') == src_id else {}
                fk = [k for k in tpk if k.startswith(('k_decode', 'k_seeds', 'k_norm', 'k_layout', 'k_connect', 'k_ends', 'k_relocate',
                                                      'k_scan', 'k_cap_guard', 'k_cls_guard', 'k_sweep_classify'))]
                if fk:
                    front['traffic'] = int(sum(tpk[k]['hbm_bytes_per_launch'] for k in fk))
            except Exception:
                pass
        except Exception as e:
            front = {'error': str(e)[:200]}
        r_front.release()
        # configs[1] (perceptron only) on the same model and batches, outside the timed region
        perceptron_only = None
        ctx2 = None
        if args.rnn and world == 1 and not emu:
            ctx2 = J.Context(img, beam=5, global_beam=6, right_check=1, right_beam=5, device=local_rank, use_rnn=False)
            def step2(i):
                t, o, n, nbytes = d_batches[i % len(d_batches)]
                return ctx2.analyze_device(t.data_ptr(), o.data_ptr(), n, nbytes, stream)
            step2(0).release()
            _sync()
            k2 = min(args.steps, 8)
            t1 = time.perf_counter()
            km2 = {}
            for i in range(k2):
                r2 = step2(1 + i)
                r2.pack(d_offs.data_ptr(), d_items.data_ptr(), cap_items)
                int(d_offs[-1].item())
                for k, v in ctx2.timings().items():
                    km2[k] = km2.get(k, 0.0) + v
                r2.release()
            _sync()
            e2 = time.perf_counter() - t1
            perceptron_only = {'workload': 'BASELINE configs[1]: same model and batches, RNN off', 'value': round(args.batch * k2 / e2, 1),
                               'unit': 'sentences/s', 'steps': k2, 'ms_per_step': round(e2 / k2 * 1e3, 3),
                               'kernel_ms_per_step': {k: round(v / k2, 3) for k, v in km2.items()}}
        # the same workload with two batches in flight (two contexts on two HIP streams): the front of a batch
        # (seeds/layout/T0: latency- and bandwidth-bound) runs beside the back of the previous one (k_sweep /
        # k_rnn_score: VALU-bound).  Reported beside `value`, not as it: per-kernel durations under overlap are
        # inflated, so the roofline figures stay those of the serial timed region above.
        overlapped = None
        if world == 1 and not args.no_overlap and not emu:
            try:
                ctxB = J.Context(img, beam=5, global_beam=6, right_check=1, right_beam=5, device=local_rank,
                                 use_rnn=None if args.rnn else False, share_with=ctx)   # (one copy of the model in HBM)
                ctxs = [ctx, ctxB]
                streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
                offsB = [torch.zeros(args.batch + 1, dtype=torch.int32, device=dev) for _ in range(2)]
                itemsB = [torch.zeros((cap_items, 2), dtype=torch.int32, device=dev) for _ in range(2)]
                # (a batch is one enqueue followed by one wait inside the library since round 5: two batches are in
                # flight when two host threads drive the two contexts, which is how jumanpp_gpu runs a device)
                import threading
                k3 = min(args.steps, 8)
                def worker(k, count):
                    torch.cuda.set_device(local_rank)
                    for j in range(count):
                        t, o, n, nbytes = d_batches[(2 * j + k) % len(d_batches)]
                        r = ctxs[k].analyze_device(t.data_ptr(), o.data_ptr(), n, nbytes, streams[k].cuda_stream)
                        r.pack(offsB[k].data_ptr(), itemsB[k].data_ptr(), cap_items)
                        streams[k].synchronize()      # (its packed result is ready)
                        r.release()
                for k in range(2):
                    worker(k, 1)
                _sync()
                t2 = time.perf_counter()
                th = [threading.Thread(target=worker, args=(k, k3 // 2)) for k in range(2)]
                for x in th:
                    x.start()
                for x in th:
                    x.join()
                _sync()
                e3 = time.perf_counter() - t2
                k3 = 2 * (k3 // 2)
                overlapped = {'what': 'two batches in flight: two contexts on two HIP streams, each driven by its own host thread, same workload',
                              'value': round(args.batch * k3 / e3, 1), 'unit': 'sentences/s', 'steps': k3,
                              'ms_per_step': round(e3 / k3 * 1e3, 3)}
                del ctxs, ctxB, offsB, itemsB, th   # (the list kept both contexts alive into the CLI leg)
            except Exception as e:  # the extra measurement must never take the main line down
                overlapped = {'error': str(e)[:200]}
        # The same batches through the host entry point of the C ABI: text and offsets in host memory in
        # (jppgpu_analyze_batch uploads them), the packed top-1 result in page-locked host memory out.  The PCIe-inclusive
        # rate of the boundary; reported beside `value`, never as it (the contract: inputs resident in HBM).
        host_buffers = None
        if world == 1 and not emu:
            try:
                import ctypes as C
                h_offs = torch.empty(args.batch + 1, dtype=torch.int32).pin_memory()
                h_items = torch.empty((cap_items, 2), dtype=torch.int32).pin_memory()
                k4 = min(args.steps, 8)

                def host_step(i):
                    text, offs = batches[i % len(batches)]
                    h = C.c_void_p()
                    if ctx.lib.jppgpu_analyze_batch(ctx.handle, text, offs.ctypes.data, len(offs) - 1, C.byref(h)) != 0:
                        raise RuntimeError(ctx.lib.jppgpu_last_error().decode())
                    res = J.Result(ctx, h)
                    res.pack(d_offs.data_ptr(), d_items.data_ptr(), cap_items)
                    _sync()                                  # (the pack kernels run on the library's stream)
                    h_offs.copy_(d_offs)
                    m = int(h_offs[-1])
                    h_items[:m].copy_(d_items[:m])
                    res.release()
                    return m
                host_step(0)
                _sync()
                t4 = time.perf_counter()
                got = sum(host_step(1 + j) for j in range(k4))
                _sync()
                e4 = time.perf_counter() - t4
                host_buffers = {'what': 'host buffers in (jppgpu_analyze_batch: text + offsets uploaded by the library), packed top-1 result '
                                        'copied to page-locked host memory; PCIe inclusive, one batch at a time',
                                'value': round(args.batch * k4 / e4, 1), 'unit': 'sentences/s', 'steps': k4,
                                'ms_per_step': round(e4 / k4 * 1e3, 3), 'morphemes_per_step': got // k4,
                                'bytes_over_pcie_per_step': int(sum(len(b[0]) + 4 * len(b[1]) for b in batches) // len(batches) + 8 * (got // k4))}
                del h_offs, h_items
            except Exception as e:  # the extra measurement must never take the main line down
                host_buffers = {'error': str(e)[:200]}
        out = {
            'metric': 'sentences/sec whole-node, beam=5 jumandic %s; achieved HBM GB/s' % ('+RNNLM' if args.rnn else 'perceptron (RNN off)'),
            'value': round(value, 1),
            'unit': 'sentences/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3),
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'u64+f32',  # 64-bit integer hashing, f32 score sums
            'data': 'synthetic' if not emu else 'synthetic -- EMULATOR CONTROL-FLOW TEST (JPPGPU_BENCH_EMU=1): not a measurement',
            'config': {
                'workload': (('BASELINE configs[2]: 1xMI355X per rank, perceptron + RNNLM (E=%d), ' % args.rnn_hidden)
                             if args.rnn else
                             'BASELINE configs[1]: 1xMI355X per rank, linear perceptron scorer only (RNN off), ')
                            + ('beam=5 gbeam=6 rcheck=1 rbeam=5, synthetic %d-codepoint UTF-8 sentences batched %d, '
                               '%d-entry synthetic jumandic-layout dictionary, 2^%d random weights'
                               % (args.sent_len, args.batch, args.dict_entries, args.weights_exp)),
                'sentences_per_step_per_gpu': args.batch,
                'nodes_per_sentence': round(ab['nodes'] / args.batch, 1),
                'failed_sentences_in_batch': bad,
                'morphemes_per_sentence': round(total_path / max(1, sentences), 2),
                'parallelism': 'sentence-sharded x%d, no data-path collective' % world,
                'result_gather': ('packed top-1 results gathered to rank 0 inside the timed region (%s backend, %d bytes per rank and step)'
                                  % (dist.get_backend(), 4 * (args.batch + 1) + 8 * gather_items))
                                 if dist is not None else 'none (one rank)',
            },
            'kernel_ms_per_step': {k: round(v, 3) for k, v in avg.items()},
            'batches': dict(pipe_stats, what='one_enqueue_batches: enqueued against the held capacity, ONE host wait (at the end); '
                                             'sized_batches: the three-wait path (the first batch of the context); '
                                             'device_allocations: hipMalloc calls of the process up to the read-out'),
            'roofline': {
                'bound': 'hbm',
                'kernel': 'k_' + dom,
                'achieved': round(achieved, 2),
                'peak': 8000.0,
                'unit': 'GB/s',
                'frac': round(achieved / 8000.0, 5),
                'traffic': traffic,
                'traffic_source': traffic_note,
                'algorithmic_bytes_per_launch': ab[dom],
                'avg_launch_ms': round(avg[dom], 3),
                'kernel_source_id': src_id,
                # the same with the gathers the kernel actually issues
                'frac_gathers_issued': round(issued / (avg['sweep'] * 1e-3) / 1e9 / 8000.0, 5) if dom == 'sweep' and avg['sweep'] > 0 else None,
                'algorithmic_bytes_gathers_issued': issued if dom == 'sweep' else None,
            },
            'roofline_front': front,
            'roofline_gather': gather_roof,
            'roofline_valu': valu_roof,
            'roofline_rnn': rnn_roof,
        }
        if parity is not None:
            out['parity_sample'] = parity
        if perceptron_only is not None:
            out['perceptron_only'] = perceptron_only
        if overlapped is not None:
            out['overlapped_two_streams'] = overlapped
        if host_buffers is not None:
            out['host_buffers'] = host_buffers
        # (this process' contexts go before the legs that run child processes or make contexts of their own)
        del ctx
        ctx2 = None
        import gc
        gc.collect()
        if not emu:
            torch.cuda.empty_cache()
        if emu:
            args.no_config5 = args.no_realism = args.no_trainer = True
        if not args.no_config5 and world == 1:
            out['config5'] = config5_leg(args, cache, local_rank, np, torch, J)
            if c5_cli is not None and isinstance(out['config5'], dict):
                out['config5']['cli_lattice'] = c5_cli   # (measured before the first device allocation of this process)
        if not args.no_realism and world == 1:
            out['realism'] = realism_legs(args, cache, local_rank, np, torch, J)
        if not args.no_trainer and world == 1:
            out['trainer'] = trainer_leg(args, mdic, cache, ge)
        if not args.no_cpu_baseline:   # rank 0 only, at every world size (the other ranks wait at the final barrier)
            out['cpu_baseline'] = cpu_baseline(args, model, mdic, cache)
        if cli_result is not None:
            out['cli_end_to_end'] = cli_result   # (measured before the first device allocation of this process, see above)
    # The JSON line is the LAST thing on stdout.  RCCL prints a version banner to the C library's stdout (found on the
    # MI355X with one rank, tests/test_dist_gpu.py): behind a pipe that buffer is flushed when the process ends, i.e.
    # after a line printed here -- so the process group goes first, the C streams are flushed, then the line.
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
    if rank == 0:
        print(json.dumps(out, ensure_ascii=False), flush=True)


if __name__ == '__main__':
    main()
