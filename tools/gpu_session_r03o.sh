#!/bin/bash
# (GPU box, round 3 session O) configs[4]-shape leg with the 256-slot candidate buffer (8 instead of 6 wavefronts per CU) + its GPU tests
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "config5 or wide or beam or routing or full_beam" > "$OUT/r03o_pytest.log" 2>&1; tail -3 "$OUT/r03o_pytest.log"
A="--no-cpu-baseline --no-overlap --no-realism --no-cli --no-parity --steps 2 --warmup 1"
for b in 4096 16384; do
  timeout 600 python bench.py $A --config5-batch $b > "$OUT/r03o_c5_$b.json" 2> "$OUT/r03o_c5_$b.err"
  python - "$OUT/r03o_c5_$b.json" $b <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d.get('config5', {})
print('config5 batch %s: %s' % (sys.argv[2], json.dumps({k: c.get(k) for k in ('value', 'ms_per_step', 'kernel_ms_per_step', 'error')})))
print('   roofline frac %s' % (c.get('roofline') or {}).get('frac'))
PY
done 2>&1 | tee "$OUT/r03o_config5.txt"
