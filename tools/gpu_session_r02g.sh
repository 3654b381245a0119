#!/bin/bash
# (GPU box, round 2 session G) k_sweep2 (two sentences per wavefront): parity, bench, occupancy variants, phases
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/r02g_pytest.log" 2>&1; tail -5 "$OUT/r02g_pytest.log"
for so in build/libjppgpu_w2.so build/libjppgpu_w3.so; do echo "== $so"; JPPGPU_LIB=$PWD/$so python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-realism --no-overlap --no-cli --no-config5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms_per_step'])"; done > "$OUT/r02g_variants.txt" 2>&1; cat "$OUT/r02g_variants.txt"
timeout 300 python tools/gpu_sweep_phases.py --rnn > "$OUT/r02g_phases_default.txt" 2>&1; tail -9 "$OUT/r02g_phases_default.txt"
timeout 300 python tools/rnn_tie_audit.py --bench-workload 5000 --verbose 5 > "$OUT/r02g_tie_audit.txt" 2>&1; tail -1 "$OUT/r02g_tie_audit.txt"
timeout 900 python bench.py --no-realism > "$OUT/r02g_bench.json" 2> "$OUT/r02g_bench.err"; tail -2 "$OUT/r02g_bench.err"; cat "$OUT/r02g_bench.json"
