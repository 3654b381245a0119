#!/bin/bash
# (GPU box, round 3 session R = final state) all GPU tests, kernel trace + HBM counter passes of the bench command
# (traffic.json stamped with the kernel source id), the complete default bench line, trainer stage times
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
A="--no-cpu-baseline --no-overlap --no-realism --no-cli --no-config5 --no-trainer"
timeout 1700 python -m pytest tests -m gpu -x -q > "$OUT/r03r_pytest.log" 2>&1; tail -4 "$OUT/r03r_pytest.log"
cd /tmp
rm -rf "$OUT/prof_trace" "$OUT/prof_fetch" "$OUT/prof_write"
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/prof_trace" -o trace -- python "$REPO/bench.py" --steps 8 --warmup 2 $A > "$OUT/r03r_trace_bench.json" 2> "$OUT/prof_trace.log"
timeout 400 rocprofv3 --pmc FETCH_SIZE -d "$OUT/prof_fetch" -o fetch -- python "$REPO/bench.py" --steps 2 --warmup 1 $A > "$OUT/prof_fetch.log" 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d "$OUT/prof_write" -o write -- python "$REPO/bench.py" --steps 2 --warmup 1 $A > "$OUT/prof_write.log" 2>&1
python "$REPO/tools/summarize_prof.py" "$OUT" 65536 40 300000 1 > "$OUT/r03r_rocprof_summary.txt" 2>&1
head -12 "$OUT/r03r_rocprof_summary.txt"
rm -rf "$OUT/prof_trace" "$OUT/prof_fetch" "$OUT/prof_write"
cd "$REPO"
cp "$OUT/traffic.json" "$REPO/profiles/traffic.json"
( time timeout 1500 python bench.py > "$OUT/r03r_bench.json" 2> "$OUT/r03r_bench.err" ) 2> "$OUT/r03r_bench_time.txt"; tail -3 "$OUT/r03r_bench_time.txt"; cut -c1-1500 "$OUT/r03r_bench.json"
