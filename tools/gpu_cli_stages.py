#!/usr/bin/env python3
"""(GPU box, developer tool) where the time of `jumanpp_gpu corpus -o file` goes: the bench workload through the CLI in
its configurations (device format / --host-format / --devices=0,0 / --devices=0,0,0,0), on 1 M and 4 M lines, with the
per-batch host stage times of GpuAnalyzer::runBatch (JPPGPU_HOST_TIMING=1) condensed to medians.

  gpurun -- 'bash tools/gpu_session.sh TAG run=tools/gpu_cli_stages.py'
"""
import os
import re
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    args = bench.build_parser().parse_args([])
    cache = args.cache
    mdic, model, img = bench.make_workload(args, cache)
    corpus = bench.make_corpus(args, mdic, cache, args.batch * 16, args.seed + 1)
    big = corpus + '.x4'
    if not os.path.exists(big):
        data = open(corpus, 'rb').read()
        with open(big, 'wb') as f:
            for _ in range(4):
                f.write(data)
    import __graft_entry__ as ge
    cli = ge.build_host()
    out = os.path.join(cache, 'cli_stage_out.txt')
    runs = [('default (device format)', [], corpus), ('--host-format', ['--host-format'], corpus),
            ('--devices=0,0', ['--devices=0,0'], corpus), ('default, 4 M lines', [], big),
            ('--devices=0,0, 4 M lines', ['--devices=0,0'], big), ('--devices=0,0,0,0, 4 M lines', ['--devices=0,0,0,0'], big),
            ('--host-format, 4 M lines', ['--host-format'], big)]
    if len(sys.argv) > 1:
        runs = [r for r in runs if any(a in r[0] for a in sys.argv[1:])]
    for name, flags, path in runs:
        env = dict(os.environ, JPPGPU_HOST_TIMING='1')
        best = None
        for rep in range(2):
            t0 = time.perf_counter()
            p = subprocess.run([cli, '--model=' + model, '--batch=%d' % args.batch, '--timing', '-o', out, path] + flags,
                               capture_output=True, text=True, env=env)
            wall = time.perf_counter() - t0
            lines = p.stderr.strip().splitlines()
            last = ([l for l in lines if 'sent_per_s=' in l] or [''])[-1]   # (the pipeline line; `reserve:` / `batches:` lines follow it)
            m = re.search(r'sent_per_s=([0-9.e+]+)', last)
            rate = float(m.group(1)) if m else 0.0
            if best is None or rate > best[0]:
                best = (rate, last, lines, wall)
            if os.path.exists(out):
                os.remove(out)
        rate, last, lines, wall = best
        print('== %s: %.0f sentences/s (process wall %.2f s)' % (name, rate, wall))
        print('   ' + last[:400])
        for l in lines:
            if l.startswith(('reserve:', 'batches:')):
                print('   ' + l[:300])
        # the first three batches against the steady ones (nothing above 1.5 x)
        rb0 = [l for l in lines if l.startswith('runBatch')]
        tot = [float(m.group(1)) for m in (re.search(r'total[= ]([0-9.]+)', l) for l in rb0) if m]
        if len(tot) > 6:
            steady = statistics.median(tot[3:])
            print('   batch totals (ms): first three %s, steady median %.2f, worst first-three / steady %.2f' % (
                [round(x, 1) for x in tot[:3]], steady, max(tot[:3]) / steady if steady > 0 else 0.0))
        rb = [l for l in lines if l.startswith('runBatch')]
        if rb:
            cols = {}
            for l in rb[2:]:   # (the first batches run on fresh buffers)
                for k, v in re.findall(r'(total|prepare|analyze|fetch|offsets)[= ]([0-9.]+)', l):
                    cols.setdefault(k, []).append(float(v))
            print('   runBatch medians over %d steady batches (ms): %s' % (len(rb) - 2, {k: round(statistics.median(v), 2) for k, v in cols.items() if v}))
            print('   first batches: ' + ' | '.join(l[9:90] for l in rb[:3]))
        ft = [l for l in lines if l.startswith('fetchText')]
        if ft:
            vals = {}
            for l in ft[2:]:
                for k, v in re.findall(r'(total|kernels|copy)[= ]([0-9.]+)', l):
                    vals.setdefault(k, []).append(float(v))
            print('   fetchText medians (ms): %s' % {k: round(statistics.median(v), 2) for k, v in vals.items() if v})


if __name__ == '__main__':
    main()
