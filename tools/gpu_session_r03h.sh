#!/bin/bash
# (GPU box, round 3 session H) front end: wave-per-sentence k_decode, LDS-staged sentence view in k_seeds / k_norm,
# seeds compiled for 8 / 6 waves; RNN hidden states as dense rows (third sync); kernel trace of the bench command
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
A="--no-cpu-baseline --no-overlap --no-realism --no-cli --no-config5"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/r03h_pytest.log" 2>&1; tail -4 "$OUT/r03h_pytest.log"
rm -f "$OUT/r03h_variants.txt"
for w in 8 6; do
  JPPGPU_DEV_SEEDS_WAVES=$w timeout 300 python bench.py --steps 8 --warmup 2 $A 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('seeds compiled for $w waves: step %.3f ms, value %.0f; kernels %s' % (j['ms_per_step'], j['value'], j['kernel_ms_per_step']))" | tee -a "$OUT/r03h_variants.txt"
done
cd /tmp
rm -rf "$OUT/prof_trace"
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/prof_trace" -o trace -- python "$REPO/bench.py" --steps 8 --warmup 2 $A > "$OUT/r03h_trace_bench.json" 2> "$OUT/prof_trace.log"
python - <<'PY' > "$OUT/r03h_rocprof_summary.txt" 2>&1
import glob, sqlite3
for db in sorted(glob.glob('/root/repo/gpurun_out/prof_trace/**/*.db', recursive=True)):
    con = sqlite3.connect(db)
    print('== rocprofv3 --kernel-trace --stats of `python bench.py --steps 8 --warmup 2`: name, calls, total_us, avg_us, pct')
    for name, calls, total, avg, pct in con.execute('select name,total_calls,total_duration,average,percentage from top_kernels'):
        print('  %-78s %5d %12.1f %10.1f %6.2f' % (name[:78], calls, total, avg, pct))
PY
head -32 "$OUT/r03h_rocprof_summary.txt"
rm -rf "$OUT/prof_trace"
