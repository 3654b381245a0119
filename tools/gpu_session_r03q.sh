#!/bin/bash
# (GPU box, round 3 session Q) jumanpp_gpu file to file with the submit / collect split (three analyzers, two writers per device)
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
echo "(CLI tests: see the previous run)"
timeout 900 python bench.py --no-cpu-baseline --no-overlap --no-realism --no-config5 --no-parity --no-trainer > "$OUT/r03q_cli.json" 2> "$OUT/r03q_cli.err"
python - <<'PY'
import json
d = json.loads(open('/root/repo/gpurun_out/r03q_cli.json').read().strip().splitlines()[-1])
print(json.dumps(d.get('cli_end_to_end'), indent=1))
PY
grep "sharded=1" "$OUT/r03q_cli.err" | tail -4
