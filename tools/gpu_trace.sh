#!/bin/bash
# (GPU box) kernel trace only: per-kernel average durations of `bench.py ${BENCH_ARGS}`
export TMPDIR=/tmp
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"; rm -rf "$OUT/prof_trace"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/prof_trace" -o trace -- python "$REPO/bench.py" --steps 4 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > "$OUT/prof_trace.log" 2>&1
python - <<'PY'
import glob, sqlite3
for db in glob.glob('/root/repo/gpurun_out/prof_trace/**/*.db', recursive=True):
    con = sqlite3.connect(db)
    for name, calls, total, avg, pct in con.execute('select name,total_calls,total_duration,average,percentage from top_kernels'):
        print('%-70s %5d %10.1f %6.2f' % (name[:70], calls, avg / 1e3 if avg > 1e5 else avg, pct))
PY
