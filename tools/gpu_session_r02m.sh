#!/bin/bash
# (GPU box, round 2 session M) lock-step RNN: sentences grouped by chain length, two row tiles per computing wavefront
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 python tools/gpu_sweep_phases.py --rnn > "$OUT/r02m_phases.txt" 2>&1; grep "lock step" "$OUT/r02m_phases.txt" | tail -1
K='import json,sys; print(json.loads(sys.stdin.read())["kernel_ms_per_step"])'
timeout 600 python bench.py --no-realism --no-cpu-baseline --no-cli --no-config5 --no-overlap 2>/dev/null | python -c "$K"
JPPGPU_RNN_NOORDER=1 timeout 600 python bench.py --no-realism --no-cpu-baseline --no-cli --no-config5 --no-overlap 2>/dev/null | python -c "$K"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k rnn 2>&1 | tail -2
timeout 600 python tools/rnn_tie_audit.py --bench-workload 5000 > "$OUT/r02m_tie_audit.txt" 2>&1; tail -1 "$OUT/r02m_tie_audit.txt"
