#!/bin/bash
# (GPU box, round 3 session W) jumanpp_gpu file to file: default worker count (cores granted) against 32 and 8 workers
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
print('usable cores (bench.usable_cores):', bench.usable_cores())
PY
MODEL=$(sed -n 1p /tmp/cli_paths.txt); CORPUS=$(sed -n 2p /tmp/cli_paths.txt)
{
cat /sys/fs/cgroup/cpu.max 2>/dev/null
for th in 0 32 0 32 12 8; do
  rm -f /tmp/cli_out.txt
  jumanpp_amd/bin/jumanpp_gpu --model=$MODEL --batch=65536 --threads=$th --timing -o /tmp/cli_out.txt $CORPUS 2>&1 | tail -1 | sed "s/^/--threads=$th: /"
done
} > "$OUT/r03w_cli_threads.txt" 2>&1
cat "$OUT/r03w_cli_threads.txt"
