"""Parser for the golden vectors written by `oracle/_ref/ref_dump dump`
(format documented in oracle/ref_dump.cc) and the lattice comparison used by
both the emulator (CPU) and the GPU parity tests."""
import struct

import numpy as np

EOS = -2147483646
BOS = -2147483648


class GoldSentence:
    pass


def read_gold(path):
    data = open(path, 'rb').read()
    assert data[:8] == b'JPPGOLD1'
    pos = 8
    hdr = struct.unpack_from('<9I', data, pos)
    pos += 36
    meta = dict(beam=hdr[0], gbeam=hdr[1], rcheck=hdr[2], rbeam=hdr[3], nscorers=hdr[4], npat=hdr[5],
                entry=hdr[6], nph=hdr[7], nsent=hdr[8])
    beam, npat, esz, nsc = meta['beam'], meta['npat'], meta['entry'], meta['nscorers']
    sents = []
    for _ in range(meta['nsent']):
        g = GoldSentence()
        g.status, g.ncp = struct.unpack_from('<II', data, pos)
        pos += 8
        if g.status != 0:
            sents.append(g)
            continue
        (nb,) = struct.unpack_from('<I', data, pos)
        pos += 4
        g.bnds = []
        for b in range(nb):
            R, L = struct.unpack_from('<II', data, pos)
            pos += 8
            ends = np.frombuffer(data, dtype='<u2', count=2 * L, offset=pos).reshape(-1, 2).copy()
            pos += 4 * L
            (ngb,) = struct.unpack_from('<I', data, pos)
            pos += 4
            gb = np.frombuffer(data, dtype=np.dtype([('left', '<u2'), ('beam', '<u2'), ('score', '<f4')]),
                               count=ngb, offset=pos).copy()
            pos += 8 * ngb
            node_dt = np.dtype([('eptr', '<i4'), ('start', '<u2'), ('end', '<u2'), ('unk', '<i4', (4,)),
                                ('entry', '<i4', (esz,)), ('pat', '<u8', (npat,)), ('t0', '<f4'),
                                ('kept', '<u4'),
                                ('beam', np.dtype([('cp', '<u2', (4,)), ('prev', '<u2', (4,)), ('total', '<f4'),
                                                   ('valid', '<u4')]), (beam,)),
                                ('cells', '<f4', (ngb * nsc,))])
            nodes = np.frombuffer(data, dtype=node_dt, count=R, offset=pos).copy()
            pos += node_dt.itemsize * R
            g.bnds.append(dict(R=R, L=L, ends=ends, gbeam=gb, nodes=nodes))
        (npath,) = struct.unpack_from('<I', data, pos)
        pos += 4
        g.path = np.frombuffer(data, dtype='<u2', count=2 * npath, offset=pos).reshape(-1, 2).copy()
        pos += 4 * npath
        (tlen,) = struct.unpack_from('<I', data, pos)
        pos += 4
        g.text = data[pos:pos + tlen].decode('utf-8')
        pos += tlen
        pos = (pos + 7) & ~7
        sents.append(g)
    assert pos == len(data), (pos, len(data))
    return meta, sents


def stored_pattern_slots(image_path):
    """the patterns a table-driven context stores per node (those a bigram / trigram feature reads, in pattern
    order: DevSpec::Pattern::slot) -- slot -> pattern index, from the model image's feature descriptors"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import model_image as mi
    blob = [p for t, a, p in mi.read_sections(image_path) if t == mi.SEC_FEATURES][0]
    f = mi.parse_features(blob)
    used = sorted({r for _, refs in f['ngrams'] if len(refs) >= 2 for r in refs})
    return used


def compare_sentence(res, s, g, meta, check_scores=True, tol=0.0, verbose=True, rnn_exact=True, pat_slots=None):
    """Compare sentence `s` of a fully fetched jumanpp_amd Result with golden `g`.
    Returns a list of mismatch strings (empty == parity).

    rnn_exact (default): RNN score cells, RNN-adjusted totals, the re-made EOS beam (order included) and the
    top-1 path must equal the reference's bit for bit -- the device evaluates the RNN with the oracle build's
    arithmetic (k-ascending fused multiply-adds, glibc's expf, the sequential NCE dot product, the fused
    weighted sums of adjustBeamScores).  rnn_exact=False is the 1e-4 contract of north_star, kept for
    reference builds whose Eigen is not the loop stand-in."""
    errs = []

    def bad(msg):
        errs.append('sent %d: %s' % (s, msg))
        return errs

    if g.status != 0:
        if res.status[s] == 0:
            bad('reference failed but device status is OK')
        return errs
    if res.status[s] != 0:
        return bad('device status %d, reference OK' % res.status[s])
    if res.ncp[s] != g.ncp:
        return bad('codepoints %d vs %d' % (res.ncp[s], g.ncp))
    nb = len(g.bnds)
    # with the RNN scorer only the elements on the surviving EOS paths carry defined totals
    # (adjustBeamScores touches every global-beam element, but RNN cells exist only on those paths)
    on_path = set()
    if meta['nscorers'] == 2 and nb > 3:
        eos = g.bnds[nb - 1]['nodes'][0]
        stack = [(nb - 1, 0, q) for q in range(meta['beam']) if eos['beam'][q]['valid']]
        while stack:
            key = stack.pop()
            if key in on_path or key[0] < 2:
                continue
            on_path.add(key)
            sl = g.bnds[key[0]]['nodes'][key[1]]['beam'][key[2]]
            stack.append((int(sl['prev'][0]), int(sl['prev'][1]), int(sl['prev'][2])))
    rtol = 0.0 if rnn_exact else 1e-4

    def close(a, e):
        if rnn_exact:
            return np.float32(a).view('<u4') == np.float32(e).view('<u4')
        return abs(float(a) - float(e)) <= rtol * max(1.0, abs(float(e)))
    nbase = int(res.node_base[s])
    bbase = int(res.bnd_base[s])
    beam = meta['beam']
    N = int(res.nnodes[s])
    total_ref = sum(b['R'] for b in g.bnds)
    if total_ref != N:
        bad('node count %d vs reference %d' % (N, total_ref))
    for b in range(nb):
        gb = g.bnds[b]
        R = int(res.bnd_count[bbase + b])
        first = int(res.bnd_first[bbase + b])
        if R != gb['R']:
            bad('boundary %d: R %d vs %d' % (b, R, gb['R']))
            continue
        # ends
        if b >= 1:
            L = int(res.end_count[bbase + b])
            if L != gb['L']:
                bad('boundary %d: L %d vs %d' % (b, L, gb['L']))
            else:
                ef = int(res.end_first[bbase + b])
                for l in range(L):
                    node = int(res.end_nodes[nbase + ef + l])
                    rb, rp = int(gb['ends'][l][0]), int(gb['ends'][l][1])
                    exp = int(res.bnd_first[bbase + rb]) + rp
                    if node != exp:
                        bad('boundary %d: end %d is node %d, reference node %d' % (b, l, node, exp))
        scored = b >= 2 and R > 0 and nb > 3
        if scored:
            ngb = int(res.gbeam_count[bbase + b])
            if ngb != len(gb['gbeam']):
                bad('boundary %d: gbeam size %d vs %d' % (b, ngb, len(gb['gbeam'])))
            else:
                for i in range(ngb):
                    e = res.gbeam_entries[bbase + b][i]
                    r = gb['gbeam'][i]
                    if e['left'] != r['left'] or e['beam'] != r['beam']:
                        bad('boundary %d: gbeam[%d] (%d,%d) vs (%d,%d)' % (b, i, e['left'], e['beam'], r['left'], r['beam']))
        for r in range(R):
            k = nbase + first + r
            gn = gb['nodes'][r]
            nd = res.nodes[k]
            if nd['start'] != gn['start'] or nd['end'] != gn['end']:
                bad('b%d n%d: span (%d,%d) vs (%d,%d)' % (b, r, nd['start'], nd['end'], gn['start'], gn['end']))
                continue
            ge = int(gn['eptr'])
            if ge >= 0 or ge in (BOS, EOS):
                if int(nd['eptr']) != ge:
                    bad('b%d n%d: eptr %d vs %d' % (b, r, nd['eptr'], ge))
            else:
                u = res.unk[k]
                # UNK entry pointers are numbered in creation order like the reference's (extra_nodes.cc:41-52)
                if int(nd['eptr']) != ge:
                    bad('b%d n%d: UNK eptr %d vs %d' % (b, r, nd['eptr'], ge))
                if (int(u['tmpl']), int(u['hash']), int(u['ph0']), int(u['ph1'])) != tuple(int(x) for x in gn['unk']):
                    bad('b%d n%d: unk (%d,%d,%d,%d) vs %s' % (b, r, u['tmpl'], u['hash'], u['ph0'], u['ph1'], gn['unk']))
            if not scored:
                continue
            if not np.array_equal(res.entry_rows[k][:len(gn['entry'])], gn['entry']):   # (rows are padded to 8 / 16 columns)
                bad('b%d n%d: entry row %s vs %s' % (b, r, res.entry_rows[k], gn['entry']))
            if pat_slots is None:
                if not np.array_equal(res.patterns[k], gn['pat']):
                    bad('b%d n%d: patterns differ' % (b, r))
            elif len(gb['gbeam']) > 0:
                # a spec outside the built-in tables: the reference's dynamic lattice keeps every pattern (of the
                # boundaries it can reach), the device the ones bigrams / trigrams read
                ours = res.patterns[k][:len(pat_slots)]
                if not np.array_equal(ours, np.asarray(gn['pat'])[pat_slots]):
                    bad('b%d n%d: stored patterns differ' % (b, r))
            # (dynamic spec: the reference never made the patterns of a boundary it cannot reach, its T0 there is noise)
            if check_scores and not (pat_slots is not None and len(gb['gbeam']) == 0):
                a, e = np.float32(res.t0[k]), np.float32(gn['t0'])
                if (tol == 0.0 and a.view('<u4') != e.view('<u4')) or (tol > 0 and abs(float(a) - float(e)) > tol):
                    bad('b%d n%d: T0 %r vs %r' % (b, r, float(a), float(e)))
            if int(res.kept[k]) != int(gn['kept']) and len(gb['gbeam']) > 0:
                bad('b%d n%d: kept %d vs %d' % (b, r, res.kept[k], gn['kept']))
            eos_rnn = meta['nscorers'] == 2 and b == nb - 1 and not rnn_exact
            if eos_rnn:
                # RNN totals carry a 1e-4 tolerance, so candidates whose totals differ by less may swap
                # ranks; require the same candidate set, matching totals, and a non-increasing order.
                dev = [(int(x['left']), int(x['beam']), float(x['total'])) for x in res.beams[k]
                       if not (x['left'] == 0xffff and x['beam'] == 0xffff)]
                ref = [(int(x['cp'][1]), int(x['cp'][3]), float(x['total'])) for x in gn['beam'] if x['valid']]
                if len(dev) != len(ref):
                    bad('EOS beam size %d vs %d' % (len(dev), len(ref)))
                if any(dev[i][2] < dev[i + 1][2] for i in range(len(dev) - 1)):
                    bad('EOS beam not sorted: %s' % (dev,))
                cutoff = min([x[2] for x in ref]) if ref else 0.0
                for (l, bm, t) in ref:
                    m = [d for d in dev if d[0] == l and d[1] == bm]
                    if m:
                        if abs(m[0][2] - t) > rtol * max(1.0, abs(t)):
                            bad('EOS beam (%d,%d): total %r vs %r' % (l, bm, m[0][2], t))
                    elif abs(t - cutoff) > rtol * max(1.0, abs(t)):
                        bad('EOS beam candidate (%d,%d) total %r missing on device' % (l, bm, t))
                continue
            for q in range(beam):
                sl = res.beams[k][q]
                gs = gn['beam'][q]
                fake = sl['left'] == 0xffff and sl['beam'] == 0xffff
                if gs['valid'] == 0:
                    if not fake:
                        bad('b%d n%d slot %d: live where reference is fake' % (b, r, q))
                    continue
                if fake:
                    bad('b%d n%d slot %d: fake where reference is live' % (b, r, q))
                    continue
                if sl['left'] != gs['cp'][1] or sl['beam'] != gs['cp'][3]:
                    bad('b%d n%d slot %d: (left,beam)=(%d,%d) vs (%d,%d)' % (b, r, q, sl['left'], sl['beam'], gs['cp'][1], gs['cp'][3]))
                pb, pr = int(gs['prev'][0]), int(gs['prev'][1])
                exp_prev = int(res.bnd_first[bbase + pb]) + pr
                if int(sl['prev_node']) != exp_prev:
                    bad('b%d n%d slot %d: prev node %d vs %d' % (b, r, q, sl['prev_node'], exp_prev))
                if check_scores and meta['nscorers'] == 1:
                    a, e = np.float32(sl['total']), np.float32(gs['total'])
                    if (tol == 0.0 and a.view('<u4') != e.view('<u4')) or (tol > 0 and abs(float(a) - float(e)) > tol):
                        bad('b%d n%d slot %d: total %r vs %r' % (b, r, q, float(a), float(e)))
                if check_scores and meta['nscorers'] == 2 and ((b, r, q) in on_path or b == nb - 1):
                    a, e = float(sl['total']), float(gs['total'])
                    if not close(a, e):
                        bad('b%d n%d slot %d: rnn-adjusted total %r vs %r' % (b, r, q, a, e))
                    # score cells of this connection: [perceptron, rnn]
                    gi = None
                    for i, ge in enumerate(gb['gbeam']):
                        if ge['left'] == gs['cp'][1] and ge['beam'] == gs['cp'][3]:
                            gi = i
                    if gi is not None and b < nb - 1:
                        c_ref = gn['cells'][gi * 2:gi * 2 + 2]
                        c_dev = res.cells[k][gi]
                        if np.float32(c_ref[0]).view('<u4') != np.float32(c_dev[0]).view('<u4'):
                            bad('b%d n%d slot %d: perceptron cell %r vs %r' % (b, r, q, float(c_dev[0]), float(c_ref[0])))
                        if not close(c_dev[1], c_ref[1]):
                            bad('b%d n%d slot %d: rnn cell %r vs %r' % (b, r, q, float(c_dev[1]), float(c_ref[1])))
    # top-1 path
    if meta['nscorers'] == 2 and nb > 3:
        eb = [float(x['total']) for x in g.bnds[nb - 1]['nodes'][0]['beam'] if x['valid']]
        if not rnn_exact and len(eb) > 1 and abs(eb[0] - eb[1]) <= rtol * max(1.0, abs(eb[0])):
            return errs  # best two paths tie within the RNN tolerance: the top-1 choice is not defined
    plen = int(res.path_len[s])
    if plen != len(g.path):
        bad('path length %d vs %d' % (plen, len(g.path)))
    else:
        for i in range(plen):
            node = int(res.path_nodes[nbase + i])
            pb, pr = int(g.path[i][0]), int(g.path[i][1])
            exp = int(res.bnd_first[bbase + pb]) + pr
            if node != exp:
                bad('path[%d]: node %d vs %d' % (i, node, exp))
    return errs
