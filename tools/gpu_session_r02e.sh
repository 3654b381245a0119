#!/bin/bash
# (GPU box, round 2 session E) wide variant: partition + parallel stable rank, threshold-search global beam; full default bench
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/r02e_pytest.log" 2>&1; tail -5 "$OUT/r02e_pytest.log"
timeout 300 python tools/gpu_sweep_phases.py --rnn --config5 > "$OUT/r02e_phases_config5.txt" 2>&1; tail -9 "$OUT/r02e_phases_config5.txt"
timeout 600 python tools/gpu_config5.py > "$OUT/r02e_config5.txt" 2>&1; cat "$OUT/r02e_config5.txt"
( time timeout 900 python bench.py > "$OUT/r02e_bench.json" 2> "$OUT/r02e_bench.err" ) 2> "$OUT/r02e_bench_time.txt"; tail -2 "$OUT/r02e_bench.err"; cat "$OUT/r02e_bench.json"; cat "$OUT/r02e_bench_time.txt"
