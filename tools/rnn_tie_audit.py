#!/usr/bin/env python3
"""RNN tie audit: for every sentence whose top-1 path differs between the
reference and the device on a perceptron + RNNLM workload, dump the EOS-beam totals of both as float
bits and classify the flip:

  (a) reference totals of the two contenders are BIT-EQUAL (an exact tie, decided by the stable order of
      makeT0Beam, score_processor.cc:426-469) and the device totals are not -> a device nondeterminism
      between twin paths; must be fixed.
  (b) reference totals differ, by less than the 1e-4 float contract -> inherent to the tolerance.
  (c) anything else -> a bug.

Also reports, over ALL sentences, how many reference-exact EOS ties the device keeps exact.

  python tools/rnn_tie_audit.py --img w.img --gold w.gold --text w.txt [--lib tests/emu/libjppgpu_emu.so]

Test/analysis infrastructure: reads goldens written by oracle/_ref/ref_dump."""
import argparse
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def bits(x):
    return struct.unpack('<I', struct.pack('<f', float(x)))[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--img')
    ap.add_argument('--gold')
    ap.add_argument('--text')
    ap.add_argument('--bench-workload', type=int, default=0, metavar='N',
                    help="audit N sentences of bench.py's workload (model and corpus are generated, the reference "
                         "goldens come from oracle/_ref/ref_dump run here)")
    ap.add_argument('--lib', default=None)
    ap.add_argument('--beams', default='5,6,1,5')
    ap.add_argument('--verbose', type=int, default=20, help='print this many flipped sentences in detail')
    args = ap.parse_args()
    import numpy as np
    import golden_io as G
    import jumanpp_amd as J

    beam, gbeam, rcheck, rbeam = [int(x) for x in args.beams.split(',')]
    if args.bench_workload:
        import subprocess
        import tempfile
        import bench
        wa = argparse.Namespace(dict_entries=300000, weights_exp=22, seed=20260925, sent_len=40, rnn=True,
                                rnn_hidden=128, rnn_vocab=30000)
        cache = os.path.join(tempfile.gettempdir(), 'jppgpu_bench_cache')
        mdic, model, args.img = bench.make_workload(wa, cache)
        args.text = bench.make_corpus(wa, mdic, cache, args.bench_workload, wa.seed + 77)
        args.gold = os.path.join(cache, 'audit_%d.gold' % args.bench_workload)
        with open(args.text, 'rb') as f:
            subprocess.check_call([os.path.join(bench.REF, 'ref_dump'), 'dump', model, args.gold, str(beam), str(gbeam),
                                   str(rcheck), str(rbeam)], stdin=f, stderr=subprocess.DEVNULL)
    lines = [l.rstrip('\n') for l in open(args.text, encoding='utf-8')]
    meta, gold = G.read_gold(args.gold)
    assert meta['nscorers'] == 2, 'the golden file must come from a model with the RNN scorer'
    ctx = J.Context(args.img, lib_path=args.lib, beam=beam, global_beam=gbeam, right_check=rcheck, right_beam=rbeam)
    res = ctx.analyze(lines).fetch(full=True)

    n = len(lines)
    stats = dict(sentences=n, top1_differs=0, class_a=0, class_b=0, class_c=0,
                 ref_exact_tie_pairs=0, ref_exact_tie_pairs_kept_exact=0, ref_top2_exact_ties=0,
                 eos_totals_bit_equal=0, eos_totals_compared=0, max_abs_total_diff=0.0)
    shown = 0
    for s in range(n):
        g = gold[s]
        if g.status != 0 or res.status[s] != 0:
            continue
        nb = len(g.bnds)
        if nb <= 3:
            continue
        nbase, bbase = int(res.node_base[s]), int(res.bnd_base[s])
        N = int(res.nnodes[s])
        eos_ref = [(int(x['cp'][1]), int(x['cp'][3]), float(x['total'])) for x in g.bnds[nb - 1]['nodes'][0]['beam'] if x['valid']]
        eos_dev = [(int(x['left']), int(x['beam']), float(x['total'])) for x in res.beams[nbase + N - 1]
                   if not (x['left'] == 0xffff and x['beam'] == 0xffff)]
        dmap = {(l, b): t for (l, b, t) in eos_dev}
        for (l, b, t) in eos_ref:
            if (l, b) in dmap:
                stats['eos_totals_compared'] += 1
                stats['eos_totals_bit_equal'] += int(bits(t) == bits(dmap[(l, b)]))
                stats['max_abs_total_diff'] = max(stats['max_abs_total_diff'], abs(t - dmap[(l, b)]))
        # exact ties in the reference between any two EOS candidates, and whether the device keeps them exact
        for i in range(len(eos_ref)):
            for j in range(i + 1, len(eos_ref)):
                if bits(eos_ref[i][2]) == bits(eos_ref[j][2]):
                    stats['ref_exact_tie_pairs'] += 1
                    a, b2 = dmap.get(eos_ref[i][:2]), dmap.get(eos_ref[j][:2])
                    if a is not None and b2 is not None and bits(a) == bits(b2):
                        stats['ref_exact_tie_pairs_kept_exact'] += 1
        if len(eos_ref) > 1 and bits(eos_ref[0][2]) == bits(eos_ref[1][2]):
            stats['ref_top2_exact_ties'] += 1
        # top-1 path
        plen = int(res.path_len[s])
        same = plen == len(g.path)
        if same:
            for i in range(plen):
                node = int(res.path_nodes[nbase + i])
                pb, pr = int(g.path[i][0]), int(g.path[i][1])
                if node != int(res.bnd_first[bbase + pb]) + pr:
                    same = False
                    break
        if same:
            continue
        stats['top1_differs'] += 1
        # contenders: reference winner and device winner
        rw = eos_ref[0]
        dw = eos_dev[0] if eos_dev else None
        cls = 'c'
        ref_of_dw = None
        if dw is not None:
            m = [x for x in eos_ref if x[:2] == dw[:2]]
            if m:
                ref_of_dw = m[0][2]
                if bits(ref_of_dw) == bits(rw[2]):
                    cls = 'a'
                elif abs(ref_of_dw - rw[2]) <= 1e-4 * max(1.0, abs(rw[2])):
                    cls = 'b'
        stats['class_' + cls] += 1
        if shown < args.verbose:
            shown += 1
            print('sentence %d: class (%s)  %s' % (s, cls, lines[s]))
            print('   reference EOS beam: ' + ' '.join('(%d,%d) %.7g [%08x]' % (l, b, t, bits(t)) for l, b, t in eos_ref))
            print('   device    EOS beam: ' + ' '.join('(%d,%d) %.7g [%08x]' % (l, b, t, bits(t)) for l, b, t in eos_dev))
            if ref_of_dw is not None:
                print('   reference total of the device winner: %.9g vs reference winner %.9g (diff %.3g)' %
                      (ref_of_dw, rw[2], rw[2] - ref_of_dw))
    print('SUMMARY ' + ' '.join('%s=%s' % (k, v) for k, v in stats.items()))


if __name__ == '__main__':
    main()
