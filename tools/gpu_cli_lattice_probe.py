#!/usr/bin/env python3
"""(GPU box, developer tool) the configs[4] command through the CLI -- beam 32, -s 32 lattice output -- on a corpus
four times the bench leg's (65 536 sentences x 220 codepoints, 3.2 GB of text) at several --batch sizes: what the
binary sustains once both analyzers of a pipeline have warm result buffers (the bench leg's two batches are both
cold)."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import __graft_entry__ as ge

args = bench.build_parser().parse_args([])
args.sent_len = 220
cache = os.path.join(tempfile.gettempdir(), 'jppgpu_bench_cache')
mdic, model, img = bench.make_workload(args, cache)
corpus = bench.make_corpus(args, mdic, cache, 65536, 31)
cli = ge.build_host()
out = os.path.join(cache, 'probe_out.txt')
flags = ['--beam=32', '--global-beam=32', '--right-beam=32', '-s', '32'] + os.environ.get('PROBE_FLAGS', '').split()   # (e.g. PROBE_FLAGS='--no-pipeline --pipelines-per-device=1': the kernels one after the other)
argv = [x for x in sys.argv[1:] if not x.startswith('--')]
trace = '--trace' in sys.argv[1:]
for batch in [int(x) for x in (argv or ['16384', '8192', '4096'])]:
  for rep in range(2):
    time.sleep(3.0)   # (a process started right behind another one's exit runs with gaps between its launches: DESIGN section 8)
    p = subprocess.run([cli, '--model=' + model, '--batch=%d' % batch, '--timing', '-o', out] + flags + [corpus],
                       capture_output=True, text=True)
    kv = bench._timing_kv(p.stderr)
    print('device_text=%d ' % int('device_lattice_format=1' in p.stderr), end='')
    if os.environ.get('PROBE_VERBOSE'):
        print('\n'.join(ln[:240] for ln in p.stderr.strip().splitlines() if ln.startswith(('batches:', 'reserve:', 'startup:', 'devices='))))
    print('batch %6d: %8.0f sentences/s  wall %7.1f ms  gpu %7.1f  analyze %7.1f  format %7.1f  write %7.1f  reserve %7.1f  rc %d' % (
        batch, kv.get('sent_per_s', 0), kv.get('wall_ms', 0), kv.get('gpu_ms', 0), kv.get('analyze_ms', 0), kv.get('format_ms', 0),
        kv.get('write_ms', 0), kv.get('reserve_ms', 0), p.returncode))
    if os.path.exists(out):
        os.remove(out)
if trace:
    # (round 6) the kernels of one run -- where the device formatter's time goes next to the analysis
    import glob
    import sqlite3
    d = os.path.join(ROOT, 'gpurun_out', 'prof_lat')
    subprocess.run(['rm', '-rf', d])
    batch = int((argv or ['4096'])[-1])
    pr = subprocess.run(['rocprofv3', '--kernel-trace', '--stats', '-d', d, '-o', 'lat', '--', cli, '--model=' + model, '--batch=%d' % batch,
                    '--clean-exit', '-o', out] + flags + [corpus], capture_output=True, text=True, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'))
    found = glob.glob(os.path.join(d, '**', '*.db'), recursive=True)
    if not found:
        print('no trace database; rocprofv3 said:', (pr.stdout or '')[-600:], (pr.stderr or '')[-1200:])
    for db in found:
        con = sqlite3.connect(db)
        try:   # (the view summarize_prof.py reads)
            rows = [(n, c, t, a) for n, c, t, a, _ in
                    con.execute('select name,total_calls,total_duration,average,percentage from top_kernels')]
        except Exception as e:
            print('trace db:', e)
            rows = []
        total = sum(r[2] for r in rows) or 1.0
        print('== kernel trace of one run at --batch=%d: name, calls, total us, avg us, %%' % batch)
        for r in rows[:16]:
            print('  %-70s %6d %12.1f %10.1f %6.2f' % (r[0][:70], r[1], r[2], r[3], 100.0 * r[2] / total))
    subprocess.run(['rm', '-rf', d])
    if os.path.exists(out):
        os.remove(out)
