#!/bin/bash
# (GPU box, round 2 session O) all GPU tests + bench after the RNN split (k_rnn_paths / k_rnn_chain / k_rnn_score<.., 2>) and k_seeds at 8 waves/SIMD
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/r02o_pytest.log" 2>&1; tail -3 "$OUT/r02o_pytest.log"
timeout 900 python bench.py --no-realism --no-cpu-baseline --no-cli --no-config5 > "$OUT/r02o_bench.json" 2> "$OUT/r02o_bench.err"; tail -2 "$OUT/r02o_bench.err"; cat "$OUT/r02o_bench.json"
timeout 600 python tools/rnn_tie_audit.py --bench-workload 5000 > "$OUT/r02o_tie_audit.txt" 2>&1; tail -1 "$OUT/r02o_tie_audit.txt"
