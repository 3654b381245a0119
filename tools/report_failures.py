#!/usr/bin/env python3
"""(GPU box) list sentences of the bench workload whose device status != OK."""
import argparse
import collections
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
import jumanpp_amd as J  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=65536)
    a = ap.parse_args()
    args = argparse.Namespace(dict_entries=300000, weights_exp=22, seed=20260925, sent_len=40, batch=a.batch, rnn=True, rnn_hidden=128, rnn_vocab=30000)
    cache = os.path.join(tempfile.gettempdir(), 'jppgpu_bench_cache')
    mdic, model, img = bench.make_workload(args, cache)
    corpus = bench.make_corpus(args, mdic, cache, args.batch * 2, args.seed + 1)
    lines = open(corpus, encoding='utf-8').read().split('\n')[:args.batch]
    ctx = J.Context(img)
    res = ctx.analyze(lines).fetch(full=True)
    print('status histogram', collections.Counter(int(x) for x in res.status))
    R = res.bnd_count
    print('max R', int(R.max()), 'max L', int(res.end_count.max()), 'nodes/sent', float(res.nnodes.mean()))
    for s in np.nonzero(res.status)[0][:40]:
        print(int(s), int(res.status[s]), lines[s])


if __name__ == '__main__':
    main()
