#!/bin/bash
# (GPU box, round 3 session F) with lane_now(): the lane index is taken afresh in every boundary iteration, so nothing derived from it is hoisted
# paths as calls): 0 = round-2 layout, 4/5 = lean two-row compiled for 4/5 waves, 15/6 = one-row for 5/6 waves
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
A="--no-cpu-baseline --no-overlap --no-realism --no-cli --no-config5"
rm -f "$OUT/r03f_waves.txt"
for w in 0 4 5 15 6; do
  JPPGPU_DEV_SWEEP_WAVES=$w timeout 300 python bench.py --steps 8 --warmup 2 $A 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('k_sweep variant $w: sweep %.3f ms, step %.3f ms, value %.0f; kernels %s' % (j['kernel_ms_per_step']['sweep'], j['ms_per_step'], j['value'], j['kernel_ms_per_step']))" | tee -a "$OUT/r03f_waves.txt"
done
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/r03f_pytest.log" 2>&1; tail -4 "$OUT/r03f_pytest.log"
