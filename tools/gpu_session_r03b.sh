#!/bin/bash
# (GPU box, round 3 session B) per-sentence routing + the lean default-configuration sweep (6.6 KB LDS) compiled
# for 4 / 5 / 6 wavefronts per SIMD: GPU tests, then the sweep time of each
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
A="--no-cpu-baseline --no-overlap --no-realism --no-cli --no-config5"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/r03b_pytest.log" 2>&1; tail -4 "$OUT/r03b_pytest.log"
for w in 4 5 6; do
  JPPGPU_DEV_SWEEP_WAVES=$w timeout 300 python bench.py --steps 8 --warmup 2 $A 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('lean k_sweep compiled for $w waves/SIMD: sweep %.3f ms, step %.3f ms, value %.0f' % (j['kernel_ms_per_step']['sweep'], j['ms_per_step'], j['value']))" | tee -a "$OUT/r03b_waves.txt"
done
