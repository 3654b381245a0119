#!/bin/bash
# (GPU box, round 3 session L) trainer throughput: 20 000 examples, reference trainer (1 / 16 threads) vs jumanpp_gpu_train
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
{
  R="$REPO/oracle/_ref"; T=/tmp/trainbench; rm -rf $T; mkdir -p $T
  python -c "import __graft_entry__ as g; g.build_host()"
  python tools/gen_dict.py 100000 --seed 3 > $T/d.mdic
  $R/jpp_jumandic_bootstrap $T/d.mdic $T/seed.model > /dev/null 2>&1
  $R/ref_dump mkmodel $T/seed.model $T/teacher.model 20 11 0.1
  python tools/gen_corpus.py $T/d.mdic 20400 --seed 5 --len 40 --oov 0.05 | grep -v '[ _"#,]' | head -20000 > $T/raw.txt   # (the Morph corpus format has no quoting for these)
  split -n l/16 $T/raw.txt $T/part_
  for f in $T/part_*; do $R/jumanpp_v2 --model=$T/teacher.model --full-morph $f 2>/dev/null | sed 's/ *$//' > $f.out & done; wait
  cat $T/part_*.out > $T/train.txt; wc -l $T/train.txt
  GB="--gb-left-min=6 --gb-left-max=6 --gb-rcheck-min=1 --gb-rcheck-max=1 --gb-right-min=5 --gb-right-max=5 --size=22"
  wall() { local t0=$(date +%s.%N); eval "$1"; local t1=$(date +%s.%N); python -c "print('   wall %.2f s' % ($t1 - $t0))"; }
  echo "== 1 epoch, 20 000 examples of 40 codepoints, 100 k-entry dictionary, 2^22 weights, global beam 6/1/5, beam 5"
  for th in 1 16; do
    echo "reference jumanpp_v2_train --batch=$((th*4)) --threads=$th"
    wall "$R/jumanpp_v2_train --model-input=$T/seed.model --model-output=$T/ref$th.model --corpus=$T/train.txt --batch=$((th*4)) --threads=$th $GB > /dev/null 2> $T/ref$th.log"; grep "finished" $T/ref$th.log | tail -1
  done
  echo "reference jumanpp_v2_train --batch=1 --threads=1"
  wall "$R/jumanpp_v2_train --model-input=$T/seed.model --model-output=$T/refb1.model --corpus=$T/train.txt --batch=1 --threads=1 $GB > /dev/null 2> $T/refb1.log"; grep "finished" $T/refb1.log | tail -1
  for b in 1 256 4096 20000; do
    echo "jumanpp_gpu_train --batch=$b (MI355X)"
    wall "jumanpp_amd/bin/jumanpp_gpu_train --model-input=$T/seed.model --model-output=$T/gpu$b.model --corpus=$T/train.txt --batch=$b $GB 2> $T/gpu$b.log"; tail -1 $T/gpu$b.log
  done
  cmp $T/refb1.model $T/gpu1.model && echo "batch 1: model files identical (20 000 examples)"
  echo "== quality of the batched models: the training sentences re-analysed with each model, words equal to the gold analysis"
  for mdl in refb1 ref16 gpu1 gpu256 gpu4096 gpu20000; do
    $R/jumanpp_v2 --model=$T/$mdl.model --full-morph $T/raw.txt 2>/dev/null | sed 's/ *$//' > $T/$mdl.out
    python - $T/train.txt $T/$mdl.out $mdl <<'PY'
import sys
g = open(sys.argv[1], encoding='utf-8').read().split('\n'); a = open(sys.argv[2], encoding='utf-8').read().split('\n')
tot = ok = sent = 0
for x, y in zip(g, a):
    if not x: continue
    gx, gy = x.split(' '), y.split(' ')
    # words by (start offset, fields)
    def spans(ws):
        out, p = set(), 0
        for w in ws:
            s = w.split('_')[0]; out.add((p, w)); p += len(s)
        return out
    sg, sa = spans(gx), spans(gy)
    tot += len(sg); ok += len(sg & sa); sent += sg == sa
print('%-10s word recall %.4f, sentences fully equal %d / %d' % (sys.argv[3], ok / tot, sent, sum(1 for x in g if x)))
PY
  done
} > "$OUT/r03l_train_throughput.txt" 2>&1
cat "$OUT/r03l_train_throughput.txt"
