#!/bin/bash
# (GPU box, round 2 session H) overlapped gathers in k_sweep (trigram requests before the tail bigrams, T1/T2 rows behind the first-stage hashes)
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/r02h_pytest.log" 2>&1; tail -3 "$OUT/r02h_pytest.log"
timeout 300 python tools/gpu_sweep_phases.py --rnn > "$OUT/r02h_phases_default.txt" 2>&1; tail -9 "$OUT/r02h_phases_default.txt"
timeout 900 python bench.py --no-realism --no-cpu-baseline > "$OUT/r02h_bench.json" 2> "$OUT/r02h_bench.err"; tail -2 "$OUT/r02h_bench.err"; cat "$OUT/r02h_bench.json"
