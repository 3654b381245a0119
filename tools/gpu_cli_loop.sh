export TMPDIR=/tmp
python bench.py --steps 1 --warmup 1 --no-realism --no-config5 --no-overlap --no-cpu-baseline --no-cli > /dev/null 2>&1   # builds the cached workload
C=/tmp/jppgpu_bench_cache
M=$(ls $C/*_rnn128.model | head -1); T=$(ls -S $C/corpus_*.txt | head -1)
echo model $M corpus $T $(wc -l < $T)
for i in 1 2 3 4 5; do jumanpp_amd/bin/jumanpp_gpu --model=$M --batch=65536 --timing -o /tmp/cli_out.txt $T 2>&1 | tail -1 | cut -c1-260; rm -f /tmp/cli_out.txt; done
for i in 1 2; do jumanpp_amd/bin/jumanpp_gpu --model=$M --batch=65536 --timing -o /dev/null $T 2>&1 | tail -1 | cut -c1-260; done
JPPGPU_HOST_TIMING=1 jumanpp_amd/bin/jumanpp_gpu --model=$M --batch=65536 --timing -o /dev/null $T 2>&1 | tail -24 | cut -c1-220
