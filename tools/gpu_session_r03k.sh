#!/bin/bash
# (GPU box, round 3 session K) GPU tests (incl. the trainer), the default bench line, k_t0 with / without the per-entry memo,
# the kernel trace and the HBM counter passes of the bench command (traffic.json stamped with the kernel source id),
# trainer throughput
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
A="--no-cpu-baseline --no-overlap --no-realism --no-cli --no-config5"
timeout 1700 python -m pytest tests -m gpu -x -q > "$OUT/r03k_pytest.log" 2>&1; tail -4 "$OUT/r03k_pytest.log"
for memo in 1 0; do
  JPPGPU_DEV_T0_MEMO=$memo timeout 300 python bench.py --steps 8 --warmup 2 $A --no-parity > "$OUT/r03k_memo$memo.json" 2> "$OUT/r03k_memo$memo.err"
  python - "$OUT/r03k_memo$memo.json" $memo <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('T0 memo %s: value %.0f, ms/step %.3f, kernels %s' % (sys.argv[2], d['value'], d['ms_per_step'], d['kernel_ms_per_step']))
PY
  grep "T0 memo" "$OUT/r03k_memo$memo.err" | head -2
done 2>&1 | tee "$OUT/r03k_t0_memo.txt"
cd /tmp
rm -rf "$OUT/prof_trace" "$OUT/prof_fetch" "$OUT/prof_write"
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/prof_trace" -o trace -- python "$REPO/bench.py" --steps 8 --warmup 2 $A > "$OUT/r03k_trace_bench.json" 2> "$OUT/prof_trace.log"
timeout 400 rocprofv3 --pmc FETCH_SIZE -d "$OUT/prof_fetch" -o fetch -- python "$REPO/bench.py" --steps 2 --warmup 1 $A > "$OUT/prof_fetch.log" 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d "$OUT/prof_write" -o write -- python "$REPO/bench.py" --steps 2 --warmup 1 $A > "$OUT/prof_write.log" 2>&1
python "$REPO/tools/summarize_prof.py" "$OUT" 65536 40 300000 1 > "$OUT/r03k_rocprof_summary.txt" 2>&1
head -40 "$OUT/r03k_rocprof_summary.txt"
rm -rf "$OUT/prof_trace" "$OUT/prof_fetch" "$OUT/prof_write"
cd "$REPO"
cp "$OUT/traffic.json" "$REPO/profiles/traffic.json"
( time timeout 1200 python bench.py > "$OUT/r03k_bench.json" 2> "$OUT/r03k_bench.err" ) 2> "$OUT/r03k_bench_time.txt"; tail -3 "$OUT/r03k_bench_time.txt"; cut -c1-3500 "$OUT/r03k_bench.json"
# trainer throughput: 20 000 examples, reference trainer (1 thread / 16 threads) vs jumanpp_gpu_train
{
  R="$REPO/oracle/_ref"; T=/tmp/trainbench; rm -rf $T; mkdir -p $T
  python tools/gen_dict.py 100000 --seed 3 > $T/d.mdic
  $R/jpp_jumandic_bootstrap $T/d.mdic $T/seed.model > /dev/null 2>&1
  $R/ref_dump mkmodel $T/seed.model $T/teacher.model 20 11 0.1
  python tools/gen_corpus.py $T/d.mdic 20000 --seed 5 --len 40 --oov 0.05 > $T/raw.txt
  split -n l/16 $T/raw.txt $T/part_
  for f in $T/part_*; do $R/jumanpp_v2 --model=$T/teacher.model --full-morph $f 2>/dev/null | sed 's/ *$//' > $f.out & done; wait
  cat $T/part_*.out > $T/train.txt; wc -l $T/train.txt
  GB="--gb-left-min=6 --gb-left-max=6 --gb-rcheck-min=1 --gb-rcheck-max=1 --gb-right-min=5 --gb-right-max=5 --size=22"
  echo "== reference jumanpp_v2_train, 1 epoch, 20 000 examples of 40 codepoints, 100 k-entry dictionary, 2^22 weights"
  for th in 1 16; do
    /usr/bin/time -f "reference --batch=$((th*4)) --threads=$th: %e s wall" $R/jumanpp_v2_train --model-input=$T/seed.model --model-output=$T/ref$th.model --corpus=$T/train.txt --batch=$((th*4)) --threads=$th $GB > /dev/null 2> $T/ref$th.log; tail -1 $T/ref$th.log
  done
  echo "== jumanpp_gpu_train (MI355X)"
  for b in 1 256 4096; do
    /usr/bin/time -f "jumanpp_gpu_train --batch=$b: %e s wall" jumanpp_amd/bin/jumanpp_gpu_train --model-input=$T/seed.model --model-output=$T/gpu$b.model --corpus=$T/train.txt --batch=$b $GB 2> $T/gpu$b.log; tail -2 $T/gpu$b.log
  done
  /usr/bin/time -f "reference --batch=1 --threads=1: %e s wall" $R/jumanpp_v2_train --model-input=$T/seed.model --model-output=$T/refb1.model --corpus=$T/train.txt --batch=1 --threads=1 $GB > /dev/null 2> $T/refb1.log; tail -1 $T/refb1.log
  cmp $T/refb1.model $T/gpu1.model && echo "batch 1: model files identical (20 000 examples)"
} > "$OUT/r03k_train_throughput.txt" 2>&1
cat "$OUT/r03k_train_throughput.txt"
