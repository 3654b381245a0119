#!/bin/bash
# (GPU box, developer tool) the CLI on the bench corpus several times in a row: run-to-run variance of a fresh
# process; with build/old/libjppgpu.so present, alternating with that library (A/B on one box)
export TMPDIR=/tmp
python bench.py --steps 16 --warmup 1 --no-realism --no-config5 --no-overlap --no-cpu-baseline --no-cli > /dev/null 2>&1   # builds the cached workload
C=/tmp/jppgpu_bench_cache
M=$(ls $C/*_rnn128.model | head -1); T=$(ls -S $C/corpus_*.txt | head -1)
echo model $M corpus $T $(wc -l < $T)
for i in 1 2 3 4; do
  echo -n "new: "; jumanpp_amd/bin/jumanpp_gpu --model=$M --batch=65536 --timing -o /dev/null $T 2>&1 | tail -1 | cut -c1-200
  if [ -f build/old/libjppgpu.so ]; then echo -n "old: "; LD_LIBRARY_PATH=$PWD/build/old jumanpp_amd/bin/jumanpp_gpu --model=$M --batch=65536 --timing -o /dev/null $T 2>&1 | tail -1 | cut -c1-200; fi
done
JPPGPU_HOST_TIMING=1 jumanpp_amd/bin/jumanpp_gpu --model=$M --batch=65536 --timing -o /dev/null $T 2>&1 | tail -20 | cut -c1-200
