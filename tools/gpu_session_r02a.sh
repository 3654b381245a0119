#!/bin/bash
# (GPU box, round 2 session A) parity after the FMA change, tie audit on the bench workload, phase shares of
# k_sweep<8,64> and k_sweep<32,512>, configs[4] timing with the tie-free beam path, bench line.
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/r02a_pytest.log" 2>&1; tail -3 "$OUT/r02a_pytest.log"
timeout 600 python tools/rnn_tie_audit.py --bench-workload 5000 --verbose 40 > "$OUT/r02a_tie_audit.txt" 2>&1; tail -2 "$OUT/r02a_tie_audit.txt"
timeout 300 python tools/gpu_sweep_phases.py --rnn > "$OUT/r02a_phases_default.txt" 2>&1; cat "$OUT/r02a_phases_default.txt"
timeout 300 python tools/gpu_sweep_phases.py --rnn --config5 > "$OUT/r02a_phases_config5.txt" 2>&1; cat "$OUT/r02a_phases_config5.txt"
timeout 600 python tools/gpu_config5.py > "$OUT/r02a_config5.txt" 2>&1; cat "$OUT/r02a_config5.txt"
timeout 600 python bench.py > "$OUT/r02a_bench.json" 2> "$OUT/r02a_bench.err"; cat "$OUT/r02a_bench.json"
