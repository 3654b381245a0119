#!/bin/bash
# (GPU box, round 3 session U) jumanpp_gpu file to file with ordinary vs page-locked (recycled) result blocks
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build_host()"
python - <<'PY'
import argparse, os, sys
sys.path.insert(0, os.getcwd())
import bench
args = argparse.Namespace(dict_entries=300000, weights_exp=22, seed=20260925, rnn=True, rnn_hidden=128, rnn_vocab=30000, sent_len=40)
cache = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'jppgpu_bench_cache')
mdic, model, img = bench.make_workload(args, cache)
corpus = bench.make_corpus(args, mdic, cache, 16 * 65536, args.seed + 1)
open('/tmp/cli_paths.txt', 'w').write(model + '\n' + corpus + '\n')
PY
MODEL=$(sed -n 1p /tmp/cli_paths.txt); CORPUS=$(sed -n 2p /tmp/cli_paths.txt)
{
for pin in 0 1 0 1; do
  for rep in 1 2 3; do
    JPPGPU_DEV_PINNED=$pin jumanpp_amd/bin/jumanpp_gpu --model=$MODEL --batch=65536 --timing -o /tmp/cli_out.txt $CORPUS 2>&1 | tail -1 | sed "s/^/pinned=$pin /"
  done
done
JPPGPU_HOST_TIMING=1 JPPGPU_DEV_PINNED=0 jumanpp_amd/bin/jumanpp_gpu --model=$MODEL --batch=65536 -o /tmp/cli_out.txt $CORPUS 2>&1 | grep runBatch | tail -4 | sed "s/^/pinned=0 /"
JPPGPU_HOST_TIMING=1 JPPGPU_DEV_PINNED=1 jumanpp_amd/bin/jumanpp_gpu --model=$MODEL --batch=65536 -o /tmp/cli_out.txt $CORPUS 2>&1 | grep runBatch | tail -4 | sed "s/^/pinned=1 /"
} > "$OUT/r03u_pinned_results.txt" 2>&1
cat "$OUT/r03u_pinned_results.txt"
