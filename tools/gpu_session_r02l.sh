#!/bin/bash
# (GPU box, round 2 session L) lock-step RNN rounds with the next round's loads in flight behind the matrix work
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rnn" > "$OUT/r02l_pytest.log" 2>&1; tail -3 "$OUT/r02l_pytest.log"
timeout 900 python bench.py --no-realism --no-cpu-baseline --no-cli --no-config5 --no-overlap > "$OUT/r02l_bench.json" 2> "$OUT/r02l_bench.err"; tail -2 "$OUT/r02l_bench.err"; cat "$OUT/r02l_bench.json"
timeout 300 python tools/gpu_sweep_phases.py --rnn > "$OUT/r02l_phases_default.txt" 2>&1; tail -16 "$OUT/r02l_phases_default.txt"
