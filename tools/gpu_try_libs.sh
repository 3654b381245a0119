#!/bin/bash
# (GPU box, developer tool) run the mini goldens through every build/libjppgpu_*.so in its own process
for so in build/libjppgpu_*.so; do
  echo "== $so"
  JPPGPU_LIB=$PWD/$so timeout 120 python - <<'PY' 2>&1 | grep -v "^W2026\|amdgpu.ids" | tail -3
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import jumanpp_amd as J, golden_io as G
lib = os.environ['JPPGPU_LIB']
for name in ('mini', 'mini_rnn'):
    ctx = J.Context('tests/golden/%s.img' % name, lib_path=lib)
    lines = [l.rstrip('\n') for l in open('tests/golden/mini.txt', encoding='utf-8')]
    meta, gold = G.read_gold('tests/golden/%s.gold' % name)
    res = ctx.analyze(lines).fetch(full=True)
    bad = sum(1 for s in range(len(lines)) if G.compare_sentence(res, s, gold[s], meta, verbose=False))
    print(name, 'sentences with mismatches:', bad)
PY
done
