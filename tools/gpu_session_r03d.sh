#!/bin/bash
# (GPU box, round 3 session D) after compute_s1 as lane = feature, packed T1 indices, reciprocal small divisions, global-typed
# (9.8 KB LDS, 4 waves/SIMD) and in the lean layout (6.6 KB) compiled for 4 / 5 / 6 wavefronts per SIMD
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
A="--no-cpu-baseline --no-overlap --no-realism --no-cli --no-config5"
rm -f "$OUT/r03d_waves.txt"
for w in 0 4 5 6; do
  JPPGPU_DEV_SWEEP_WAVES=$w timeout 300 python bench.py --steps 8 --warmup 2 $A 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('k_sweep variant $w (0 = round-2 layout, else lean layout compiled for that many waves/SIMD): sweep %.3f ms, step %.3f ms, value %.0f; kernels %s' % (j['kernel_ms_per_step']['sweep'], j['ms_per_step'], j['value'], j['kernel_ms_per_step']))" | tee -a "$OUT/r03d_waves.txt"
done
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/r03d_pytest.log" 2>&1; tail -4 "$OUT/r03d_pytest.log"
