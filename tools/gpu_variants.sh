#!/bin/bash
# (GPU box) A/B of prebuilt library variants under build/: prints kernel ms per variant
for so in build/libjppgpu_*.so; do
  echo "== $so"
  JPPGPU_LIB=$PWD/$so python bench.py --steps 6 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms_per_step'])"
done
