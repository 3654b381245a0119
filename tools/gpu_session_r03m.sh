#!/bin/bash
# (GPU box, round 3 session M) where the device-driven trainer spends its time (stage timers of jumanpp_gpu_train)
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
{
  R="$REPO/oracle/_ref"; T=/tmp/trainbench; rm -rf $T; mkdir -p $T
  python -c "import __graft_entry__ as g; g.build_host()"
  python tools/gen_dict.py 100000 --seed 3 > $T/d.mdic
  $R/jpp_jumandic_bootstrap $T/d.mdic $T/seed.model > /dev/null 2>&1
  $R/ref_dump mkmodel $T/seed.model $T/teacher.model 20 11 0.1
  python tools/gen_corpus.py $T/d.mdic 20400 --seed 5 --len 40 --oov 0.05 2>/dev/null | grep -v '[ _"#,]' | head -20000 > $T/raw.txt
  split -n l/16 $T/raw.txt $T/part_
  for f in $T/part_*; do $R/jumanpp_v2 --model=$T/teacher.model --full-morph $f 2>/dev/null | sed 's/ *$//' > $f.out & done; wait
  cat $T/part_*.out > $T/train.txt; wc -l $T/train.txt
  GB="--gb-left-min=6 --gb-left-max=6 --gb-rcheck-min=1 --gb-rcheck-max=1 --gb-right-min=5 --gb-right-max=5 --size=22"
  for b in 64 256 1024 4096; do
    echo "jumanpp_gpu_train --batch=$b"
    t0=$(date +%s.%N); jumanpp_amd/bin/jumanpp_gpu_train --model-input=$T/seed.model --model-output=$T/gpu$b.model --corpus=$T/train.txt --batch=$b $GB 2>&1 | tail -2; t1=$(date +%s.%N)
    python -c "print('   wall %.2f s' % ($t1 - $t0))"
  done
} > "$OUT/r03m_train_stages.txt" 2>&1
cat "$OUT/r03m_train_stages.txt"
