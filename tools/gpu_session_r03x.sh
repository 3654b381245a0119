#!/bin/bash
# (GPU box, round 3 session X) jumanpp_gpu file to file at steady state: 4 M lines (64 batches), output to /dev/null and to a file
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
big = os.path.join(cache, 'corpus_4m.txt')
with open(big, 'wb') as f:
    data = open(corpus, 'rb').read()
    for _ in range(4):
        f.write(data)
open('/tmp/cli_paths.txt', 'w').write(model + '\n' + corpus + '\n' + big + '\n')
PY
MODEL=$(sed -n 1p /tmp/cli_paths.txt); CORPUS=$(sed -n 2p /tmp/cli_paths.txt); BIG=$(sed -n 3p /tmp/cli_paths.txt)
{
df -h /tmp | tail -1
for rep in 1 2; do
  jumanpp_amd/bin/jumanpp_gpu --model=$MODEL --batch=65536 --timing -o /dev/null $BIG 2>&1 | tail -1 | sed "s/^/4 M lines -> \/dev\/null: /"
done
rm -f /tmp/cli_out.txt
jumanpp_amd/bin/jumanpp_gpu --model=$MODEL --batch=65536 --timing -o /tmp/cli_out.txt $BIG 2>&1 | tail -1 | sed "s/^/4 M lines -> file (9 GB): /"
ls -la /tmp/cli_out.txt | awk '{print $5}'
rm -f /tmp/cli_out.txt
jumanpp_amd/bin/jumanpp_gpu --model=$MODEL --batch=65536 --timing -o /tmp/cli_out.txt $CORPUS 2>&1 | tail -1 | sed "s/^/1 M lines -> file: /"
rm -f /tmp/cli_out.txt
} > "$OUT/r03x_cli_steady_state.txt" 2>&1
cat "$OUT/r03x_cli_steady_state.txt"
