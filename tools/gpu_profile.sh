#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): bench line, failing-sentence report,
# rocprofv3 kernel trace + HBM counters.  Outputs under gpurun_out/.
set -u
REPO="$(pwd)"
OUT="$REPO/gpurun_out"
mkdir -p "$OUT"
export TMPDIR=/tmp
STEPS="${STEPS:-8}"
python bench.py --steps "$STEPS" --warmup 2 > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -3 "$OUT/bench.err"; cat "$OUT/bench.json"
python tools/report_failures.py > "$OUT/failures.txt" 2>&1; tail -5 "$OUT/failures.txt"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/prof_trace" -o trace -- python "$REPO/bench.py" --steps 4 --warmup 1 --no-cpu-baseline --no-overlap > "$OUT/prof_trace.log" 2>&1
ls -R "$OUT/prof_trace" | head -20
rocprofv3 --pmc FETCH_SIZE -d "$OUT/prof_fetch" -o fetch -- python "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-overlap > "$OUT/prof_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$OUT/prof_write" -o write -- python "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-overlap > "$OUT/prof_write.log" 2>&1
python "$REPO/tools/summarize_prof.py" "$OUT" 65536 40 300000 1 > "$OUT/prof_summary.txt" 2>&1
cat "$OUT/prof_summary.txt"
