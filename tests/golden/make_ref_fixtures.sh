#!/bin/bash
# Golden vectors over the REFERENCE'S OWN test fixtures.  Run in the build container:
#   bash tests/golden/make_ref_fixtures.sh
# Reads /root/reference/test/jumandic/{jumanpp_minimal,codegen,bug-28-lattice,bug950111-003}.mdic, the sentences the
# reference's tests run over them (jumandic_codegen_test.cc:52-63, bug_28_lattice.cc, bug_950111-003_test.cc,
# train_mini_01.txt, partial_01.data) and writes, for every dictionary, into tests/golden/ref/:
#   <dic>.jppmdl          jpp_jumandic_bootstrap + 2^14 random weights (ref_dump mkmodel) -- a model FILE is a build
#                         product of the reference's tools, not reference source
#   minimal_trained.jppmdl  the jumanpp_minimal dictionary trained by jumanpp_v2_train on train_mini_01.txt
#   <dic>.txt             input sentences (the surfaces of the fixture's analysed lines, one per line)
#   <dic>.<mode>.out      stdout of oracle/_ref/jumanpp_v2 in that mode
#   minimal.img           ref_dump export of minimal.jppmdl (flat image for the ctypes harness)
#   minimal.gold          ref_dump dump: the whole lattice (nodes, rows, patterns, T0, global beams, beams, cells)
# /root/reference does not exist on the GPU box; these files do.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
REF="$ROOT/oracle/_ref"
FIX=/root/reference/test/jumandic
OUT="$HERE/ref"
TMP="$(mktemp -d)"
mkdir -p "$OUT"

# surfaces of a Morph-format training file: "surf_read_base_pos_..._... surf_..." -> concatenated surfaces;
# a trailing "# comment" is dropped, lines that are only comments or empty vanish
surfaces() {
  python3 - "$1" <<'PY'
import sys
seen = []
for line in open(sys.argv[1], encoding='utf-8'):
    line = line.split(' # ')[0].strip()
    if not line or line.startswith('#'):
        continue
    s = ''.join(tok.split('_')[0] for tok in line.split(' ') if tok)
    if s and s not in seen:
        seen.append(s)
print('\n'.join(seen))
PY
}

# the sentence of jumandic_codegen_test.cc:52-63 is analysed over codegen.mdic in that test
CODEGEN_SENT='５５１年もガラフケマペが兵をつの〜ってたな！'

declare -A DIC=( [minimal]=jumanpp_minimal [codegen]=codegen [bug28]=bug-28-lattice [bug950111]=bug950111-003 )
for key in minimal codegen bug28 bug950111; do
  src="$FIX/${DIC[$key]}.mdic"
  "$REF/jpp_jumandic_bootstrap" "$src" "$TMP/$key.seed" > /dev/null 2>&1
  "$REF/ref_dump" mkmodel "$TMP/$key.seed" "$OUT/$key.jppmdl" 14 20260926 0.1
  {
    surfaces "$FIX/train_mini_01.txt"
    surfaces "$FIX/bug-28-lattice.in"
    surfaces "$FIX/bug950111-003.in"
    surfaces "$FIX/unk_ex.data"
    echo "$CODEGEN_SENT"
    # the lines of partial_01.data, glued (the plain text of its examples)
    python3 - "$FIX/partial_01.data" <<'PY'
import sys
cur = []
for line in open(sys.argv[1], encoding='utf-8'):
    line = line.rstrip('\n')
    if line.startswith('#'):
        continue
    if not line:
        if cur:
            print(''.join(cur))
        cur = []
    else:
        cur.append(line.replace('\t', ''))
if cur:
    print(''.join(cur))
PY
  } | awk '!seen[$0]++' > "$OUT/$key.txt"
done

# a trained model over the reference's own mini corpus (perceptron only; --batch 1 is the deterministic case)
"$REF/jumanpp_v2_train" --model-input="$TMP/minimal.seed" --model-output="$OUT/minimal_trained.jppmdl" \
    --corpus="$FIX/train_mini_01.txt" --size=14 --max-epochs=3 --epsilon=0 --batch=1 --threads=1 > /dev/null 2>&1
cp "$OUT/minimal.txt" "$OUT/minimal_trained.txt"
cp "$FIX/partial_01.data" "$OUT/partial_01.data"

run() {  # run <model-key> <mode-name> <args...>: stdout of the reference CLI
  local key="$1" mode="$2"; shift 2
  "$REF/jumanpp_v2" --model="$OUT/$key.jppmdl" "$@" "$OUT/$key.txt" > "$OUT/$key.$mode.out" 2> /dev/null || true
}
for key in minimal minimal_trained codegen bug28 bug950111; do
  run "$key" juman
  run "$key" s5 -s 5
  run "$key" gbeam0 --global-beam=0
  run "$key" b32 --beam=32 --global-beam=32 --right-beam=32 -s 32
  run "$key" b3g10 --beam=3 --global-beam=10 --right-check=2 --right-beam=4
  run "$key" morph -M
  run "$key" segment --segment
done
for key in minimal minimal_trained; do
  "$REF/jumanpp_v2" --model="$OUT/$key.jppmdl" --partial-input "$OUT/partial_01.data" > "$OUT/$key.partial.out" 2> /dev/null || true
done
# the whole lattice of the jumanpp_minimal sentences, default beams and beam 32
"$REF/ref_dump" dump "$OUT/minimal.jppmdl" "$OUT/minimal.gold" < "$OUT/minimal.txt" 2> /dev/null
"$REF/ref_dump" dump "$OUT/minimal_trained.jppmdl" "$OUT/minimal_trained.gold" < "$OUT/minimal.txt" 2> /dev/null
"$REF/ref_dump" dump "$OUT/minimal.jppmdl" "$OUT/minimal_b32.gold" 32 32 1 32 < "$OUT/minimal.txt" 2> /dev/null
# the flat image of the same models for the ctypes harness (tests/golden_io.py compares the fetched lattice with *.gold)
"$REF/ref_dump" export "$OUT/minimal.jppmdl" "$OUT/minimal.img" 2> /dev/null
"$REF/ref_dump" export "$OUT/minimal_trained.jppmdl" "$OUT/minimal_trained.img" 2> /dev/null
rm -rf "$TMP"
ls -la "$OUT"
du -sh "$OUT"
