#!/usr/bin/env python3
"""(GPU box, developer experiment, round 6) are k_lat_count / k_lat_write held by a few slow sentences?  The configs[4]
command (one pipeline, stages one after the other) on 16 384 copies of ONE sentence, under a kernel trace: no imbalance
between wavefronts is possible there."""
import glob
import os
import sqlite3
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
import __graft_entry__ as ge

a = bench.build_parser().parse_args([])
a.sent_len = 220
cache = os.path.join(tempfile.gettempdir(), 'jppgpu_bench_cache')
mdic, model, img = bench.make_workload(a, cache)
corpus = bench.make_corpus(a, mdic, cache, 16384, 31)
lines = open(corpus, 'rb').read().split(b'\n')
cli = ge.build_host()
out = os.path.join(cache, 'latu_out.txt')
BATCH = os.environ.get('LATU_BATCH', '8192')
for tag, which in (('the corpus itself', None), ('16 384 copies of its line 0', 0), ('16 384 copies of its line 7', 7)):
    path = corpus
    if which is not None:
        path = os.path.join(cache, 'latu_%d.txt' % which)
        open(path, 'wb').write((lines[which] + b'\n') * 16384)
    d = os.path.join(ROOT, 'gpurun_out', 'prof_latu')
    subprocess.run(['rm', '-rf', d])
    subprocess.run(['rocprofv3', '--kernel-trace', '--stats', '-d', d, '-o', 'latu', '--', cli, '--model=' + model, '--batch=' + BATCH, '--no-pipeline',
                    '--pipelines-per-device=1', '--clean-exit', '--beam=32', '--global-beam=32', '--right-beam=32', '-s', '32', '-o', out, path],
                   capture_output=True, text=True, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'))
    print('==', tag, '--batch=' + BATCH, '(%d bytes of text)' % (os.path.getsize(out) if os.path.exists(out) else -1))
    for db in glob.glob(os.path.join(d, '**', '*.db'), recursive=True):
        con = sqlite3.connect(db)
        for n, c, t, av, _ in con.execute('select name,total_calls,total_duration,average,percentage from top_kernels'):
            if 'k_lat_' in n or 'k_sweep<32, 64' in n:
                print('   %-40s %3d calls  avg %9.1f us' % (n[:40], c, av))
    subprocess.run(['rm', '-rf', d])
