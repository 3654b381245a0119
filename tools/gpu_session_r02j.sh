#!/bin/bash
# (GPU box, round 2 session J) k_rnn_score_reg: one sentence per workgroup with W^T in registers, k_rnn_score for the rest
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/r02j_pytest.log" 2>&1; tail -3 "$OUT/r02j_pytest.log"
timeout 900 python bench.py --no-realism --no-cpu-baseline --no-cli --no-config5 > "$OUT/r02j_bench.json" 2> "$OUT/r02j_bench.err"; tail -2 "$OUT/r02j_bench.err"; cat "$OUT/r02j_bench.json"
timeout 600 python tools/rnn_tie_audit.py --bench-workload 5000 > "$OUT/r02j_tie_audit.txt" 2>&1; tail -5 "$OUT/r02j_tie_audit.txt"
