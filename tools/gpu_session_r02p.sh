#!/bin/bash
# (GPU box, round 2 session P) final profile of the round: kernel trace + HBM counter passes of the bench command,
# configs[4] trace, the default bench line (all legs), the GPU test log
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
rm -rf "$OUT/prof_trace" "$OUT/prof_fetch" "$OUT/prof_write" "$OUT/c5_trace"
cd /tmp
A="--no-cpu-baseline --no-overlap --no-realism --no-cli --no-config5"
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/prof_trace" -o trace -- python "$REPO/bench.py" --steps 8 --warmup 2 $A > "$OUT/r02p_trace_bench.json" 2> "$OUT/prof_trace.log"
timeout 400 rocprofv3 --pmc FETCH_SIZE -d "$OUT/prof_fetch" -o fetch -- python "$REPO/bench.py" --steps 2 --warmup 1 $A > "$OUT/prof_fetch.log" 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d "$OUT/prof_write" -o write -- python "$REPO/bench.py" --steps 2 --warmup 1 $A > "$OUT/prof_write.log" 2>&1
python "$REPO/tools/summarize_prof.py" "$OUT" 65536 40 300000 1 > "$OUT/r02p_rocprof_summary.txt" 2>&1
head -40 "$OUT/r02p_rocprof_summary.txt"
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/c5_trace" -o trace -- python "$REPO/tools/gpu_config5.py" --device-only > "$OUT/r02p_config5.txt" 2>&1
python - <<'PY' > "$OUT/r02p_config5_rocprof_summary.txt" 2>&1
import glob, sqlite3
for db in sorted(glob.glob('/root/repo/gpurun_out/c5_trace/**/*.db', recursive=True)):
    con = sqlite3.connect(db)
    print('== rocprofv3 --kernel-trace --stats of tools/gpu_config5.py --device-only: name, calls, total_us, avg_us, pct')
    for name, calls, total, avg, pct in con.execute('select name,total_calls,total_duration,average,percentage from top_kernels'):
        print('  %-78s %5d %12.1f %10.1f %6.2f' % (name[:78], calls, total, avg, pct))
PY
tail -5 "$OUT/r02p_config5.txt"; head -12 "$OUT/r02p_config5_rocprof_summary.txt"
rm -rf "$OUT/c5_trace"
cd "$REPO"
cp "$OUT/traffic.json" "$REPO/profiles/traffic.json"   # (so that the bench line below carries this session's counters)
( time timeout 900 python bench.py > "$OUT/r02p_bench.json" 2> "$OUT/r02p_bench.err" ) 2> "$OUT/r02p_bench_time.txt"; tail -3 "$OUT/r02p_bench_time.txt"; cat "$OUT/r02p_bench.json"
timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/r02p_pytest.log" 2>&1; tail -3 "$OUT/r02p_pytest.log"
