#!/bin/bash
# Regenerates the committed golden fixtures from the REAL reference (oracle/_ref,
# built by oracle/Makefile from /root/reference).  Run in the build container:
#   bash tests/golden/make_golden.sh
# Inputs are fully synthetic and seeded (tools/gen_dict.py, tools/gen_corpus.py).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
REF="$ROOT/oracle/_ref"
TMP="$(mktemp -d)"
python3 "$ROOT/tools/gen_dict.py" 2500 --seed 11 > "$TMP/mini.mdic"
"$REF/jpp_jumandic_bootstrap" "$TMP/mini.mdic" "$TMP/mini.seed" > /dev/null 2>&1
"$REF/ref_dump" mkmodel "$TMP/mini.seed" "$TMP/mini.model" 14 20260925 0.1
"$REF/ref_dump" export "$TMP/mini.model" "$HERE/mini.img"
cp "$TMP/mini.model" "$HERE/mini.jppmdl"   # the reference's own container, read natively by jumanpp_amd/host
{
  python3 "$ROOT/tools/gen_corpus.py" "$TMP/mini.mdic" 20 --seed 21 --oov 0.15 --len 40
  python3 "$ROOT/tools/gen_corpus.py" "$TMP/mini.mdic" 2 --seed 22 --oov 0.2 --len 90
  # edge cases: empty line, single char, ASCII only, digits with separators, prolong/small kana, onomatopoeia
  printf '\n'
  printf 'あ\n'
  printf 'hello, world (test) [x]\n'
  printf '１２，３４５．６７キロ数十何百分の一ぶんの３\n'
  printf 'すごーーい〜かぁっこいいねぇっッ！ケーキとヶ月\n'
  printf 'どきどきドキドキわんわんわんわんぱたぱたた\n'
} > "$HERE/mini.txt"
"$REF/ref_dump" dump "$TMP/mini.model" "$HERE/mini.gold" < "$HERE/mini.txt"
"$REF/ref_dump" dump "$TMP/mini.model" "$HERE/mini_b3.gold" 3 4 2 3 < "$HERE/mini.txt"
# beam 4 / global beam 12: makeT0Beam's quickselect branch (more candidates than beam*4/3), first 8 lines
head -8 "$HERE/mini.txt" | "$REF/ref_dump" dump "$TMP/mini.model" "$HERE/mini_b4g12.gold" 4 12 1 4
"$REF/jumanpp_v2" --model="$TMP/mini.model" "$HERE/mini.txt" > "$HERE/mini.juman.txt"
# perceptron + synthetic RNNLM (faster-rnnlm NCE format), embedded by the reference's own trainer binary
python3 "$ROOT/tools/gen_rnn.py" "$TMP/mini.mdic" "$TMP/mini_rnn" --vocab 600 --hidden 32 --maxent-size 16384 --seed 31
"$REF/jumanpp_v2_train" --model-input="$TMP/mini.model" --model-output="$TMP/mini_rnn.model" \
    --rnn-model="$TMP/mini_rnn" --rnn-fields=surface,pos --rnn-nce-bias=5.6 --rnn-unk-constant=-3.47 \
    --rnn-unk-length=-2.93 --feature-weight-perceptron=1 --feature-weight-rnn=0.0176 > /dev/null 2>&1
"$REF/ref_dump" export "$TMP/mini_rnn.model" "$HERE/mini_rnn.img"
cp "$TMP/mini_rnn.model" "$HERE/mini_rnn.jppmdl"
"$REF/ref_dump" dump "$TMP/mini_rnn.model" "$HERE/mini_rnn.gold" < "$HERE/mini.txt" 2> /dev/null
head -8 "$HERE/mini.txt" | "$REF/ref_dump" dump "$TMP/mini_rnn.model" "$HERE/mini_rnn_b4g12.gold" 4 12 1 4 2> /dev/null
"$REF/jumanpp_v2" --model="$TMP/mini_rnn.model" "$HERE/mini.txt" > "$HERE/mini_rnn.juman.txt"
rm -rf "$TMP"
ls -la "$HERE"
