#!/bin/bash
# (GPU box) per-kernel averages (rocprofv3 kernel trace) for every build/libjppgpu_*.so variant; KERNELS = grep pattern
export TMPDIR=/tmp
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
for so in build/libjppgpu_*.so; do
  echo "== $so"
  rm -rf "$OUT/vtrace"
  (cd /tmp && JPPGPU_LIB=$REPO/$so rocprofv3 --kernel-trace --stats -d "$OUT/vtrace" -o t -- python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > /dev/null 2>&1)
  python - <<'PY' | grep -E "${KERNELS:-k_}"
import glob, sqlite3
for db in glob.glob('/root/repo/gpurun_out/vtrace/**/*.db', recursive=True):
    con = sqlite3.connect(db)
    for name, calls, total, avg, pct in con.execute('select name,total_calls,total_duration,average,percentage from top_kernels'):
        print('%-60s %5d %10.1f' % (name[:60], calls, avg))
PY
done
