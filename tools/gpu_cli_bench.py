#!/usr/bin/env python3
"""End-to-end throughput of the jumanpp_gpu CLI (read file -> analyse on the MI355X -> JUMAN text
-> write file) on bench.py's workload, next to the reference CLI on a sample of the same lines.

  gpurun -- 'python tools/gpu_cli_bench.py > gpurun_out/cli_bench.txt 2>&1'

Developer tool; the numbers go to DESIGN.md section 7.
"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lines', type=int, default=262144)
    ap.add_argument('--ref-sample', type=int, default=5000)
    ap.add_argument('--no-rnn', dest='rnn', action='store_false', default=True)
    a = ap.parse_args()
    args = argparse.Namespace(dict_entries=300000, weights_exp=22, seed=20260925, rnn=a.rnn, rnn_hidden=128,
                              rnn_vocab=30000, sent_len=40)
    cache = '/tmp/jppgpu_bench_cache'
    mdic, model, img = bench.make_workload(args, cache)
    corpus = bench.make_corpus(args, mdic, cache, a.lines, 7)
    import __graft_entry__ as ge
    cli = ge.build_host()
    out = '/tmp/cli_out.txt'
    print('host cores:', os.cpu_count(), flush=True)
    runs = [('serial: 1 format thread, no pipeline', ['--threads=1', '--no-pipeline']),
            ('format threads only', ['--no-pipeline']),
            ('pipeline + format threads (default)', []),
            ('pipeline, 64 format threads', ['--threads=64']),
            ('pipeline, 16 format threads', ['--threads=16']),
            ('default, native .jppmdl model', ['MODEL'])]
    first = None
    for name, extra in runs:
        m = img
        if extra == ['MODEL']:
            m, extra = model, []
        t0 = time.time()
        p = subprocess.run([cli, '--model=' + m, '--timing', '-o', out] + extra + [corpus], capture_output=True, text=True)
        wall = time.time() - t0
        data = open(out, 'rb').read()
        if first is None:
            first = data
        print('%-42s rc=%d process_wall=%.2fs same_output=%s\n    %s' % (name, p.returncode, wall, data == first,
                                                                      p.stderr.strip().splitlines()[-1]), flush=True)
    # the reference CLI on the first lines of the same corpus
    sample = '/tmp/cli_sample.txt'
    with open(corpus, 'rb') as f, open(sample, 'wb') as g:
        for i, line in enumerate(f):
            if i >= a.ref_sample:
                break
            g.write(line)
    t0 = time.time()
    ref = subprocess.run([os.path.join(bench.REF, 'jumanpp_v2'), '--model=' + model, sample], capture_output=True)
    dt = time.time() - t0
    print('reference jumanpp_v2, 1 thread: %d lines in %.2fs = %.0f sent/s' % (a.ref_sample, dt, a.ref_sample / dt))
    ours = first.split(b'EOS\n')
    refs = ref.stdout.split(b'EOS\n')
    same = sum(1 for x, y in zip(ours, refs[:-1]) if x == y)
    print('identical sentence blocks vs reference on the sample: %d of %d' % (same, len(refs) - 1))


if __name__ == '__main__':
    main()
