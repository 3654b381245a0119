#!/bin/bash
# (GPU box, round 3 session V) jumanpp_gpu file to file: host-side stage times of every batch of a 1 M-line run (where does the warm-up go)
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
for rep in 1 2; do
  rm -f /tmp/cli_out.txt
  JPPGPU_HOST_TIMING=1 jumanpp_amd/bin/jumanpp_gpu --model=$MODEL --batch=65536 --timing -o /tmp/cli_out.txt $CORPUS 2>&1 | grep "runBatch\|sharded=1"
  echo
done
} > "$OUT/r03v_cli_batches.txt" 2>&1
cat "$OUT/r03v_cli_batches.txt"
