#!/bin/bash
# (GPU box, round 3 session N) the configs[4]-shape leg at larger batches: does one wavefront per 1 100-node sentence fill the chip when there are more sentences?
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
A="--no-cpu-baseline --no-overlap --no-realism --no-cli --no-parity --steps 2 --warmup 1"
for b in 4096 16384 32768; do
  timeout 600 python bench.py $A --config5-batch $b > "$OUT/r03n_c5_$b.json" 2> "$OUT/r03n_c5_$b.err"
  python - "$OUT/r03n_c5_$b.json" $b <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d.get('config5', {})
print('config5 batch %s: %s' % (sys.argv[2], json.dumps({k: c.get(k) for k in ('value', 'ms_per_step', 'kernel_ms_per_step', 'error')})))
r = c.get('roofline') or {}
print('   roofline frac %s' % r.get('frac'))
PY
done 2>&1 | tee "$OUT/r03n_config5_batches.txt"
