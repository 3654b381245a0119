#!/bin/bash
# (GPU box, round 2 session K) k_rnn_score with the recurrence on the matrix cores (lock step over the 16 sentences of a workgroup)
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/r02k_pytest.log" 2>&1; tail -3 "$OUT/r02k_pytest.log"
timeout 900 python bench.py --no-realism --no-cpu-baseline --no-cli --no-config5 > "$OUT/r02k_bench.json" 2> "$OUT/r02k_bench.err"; tail -2 "$OUT/r02k_bench.err"; cat "$OUT/r02k_bench.json"
JPPGPU_RNN_LDSW=1 timeout 900 python bench.py --no-realism --no-cpu-baseline --no-cli --no-config5 --no-overlap > "$OUT/r02k_bench_ldsw.json" 2> "$OUT/r02k_bench_ldsw.err"; cat "$OUT/r02k_bench_ldsw.json"
timeout 600 python tools/rnn_tie_audit.py --bench-workload 5000 > "$OUT/r02k_tie_audit.txt" 2>&1; tail -3 "$OUT/r02k_tie_audit.txt"
