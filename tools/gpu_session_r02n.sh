#!/bin/bash
# (GPU box, round 2 session N) kernel trace of the bench command with and without chain-length grouping
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for v in order; do
  rm -rf "$OUT/prof_$v"
  if [ $v = noorder ]; then export JPPGPU_RNN_NOORDER=1; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$v" -o t -- python "$REPO/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-overlap --no-realism --no-cli --no-config5 > /dev/null 2> "$OUT/prof_$v.log"
  f=$(find "$OUT/prof_$v" -name "*kernel_stats.csv" | head -1)
  echo "== $v"; head -12 "$f" | cut -d, -f1-5 | cut -c1-150
done
