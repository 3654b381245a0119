#!/bin/bash
# (GPU box, round 2 session Q) the default bench line and the GPU test log of the final tree
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
( time timeout 900 python bench.py > "$OUT/r02q_bench.json" 2> "$OUT/r02q_bench.err" ) 2> "$OUT/r02q_bench_time.txt"; tail -3 "$OUT/r02q_bench_time.txt"; cat "$OUT/r02q_bench.json"
timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/r02q_pytest.log" 2>&1; tail -3 "$OUT/r02q_pytest.log"
