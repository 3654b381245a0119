#!/usr/bin/env python3
"""(GPU box, developer tool) one realism leg of bench.py (--weights-exp N) on its own, for a kernel trace:
   rocprofv3 --kernel-trace -d DIR -o t -- python tools/gpu_realism_gaps.py 24 ; python tools/gpu_realism_gaps.py --gaps DIR
   prints the idle gaps between consecutive kernels of the timed steps."""
import glob
import os
import sqlite3
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def gaps(d):
    db = glob.glob(os.path.join(d, '**', '*.db'), recursive=True)[0]
    con = sqlite3.connect(db)
    rows = list(con.execute('select name, start, end from kernels order by start'))
    last = None
    out = []
    for name, st, en in rows:
        if last is not None and st - last[2] > 300000:
            out.append((st - last[2], last[0][:50], name[:50]))
        last = (name, st, en)
    print('kernels', len(rows), 'span ms', (rows[-1][2] - rows[0][1]) / 1e6)
    for g, a, b in out[-40:]:
        print('gap %.2f ms after %-50s before %s' % (g / 1e6, a, b))


def main():
    if sys.argv[1] == '--gaps':
        return gaps(sys.argv[2])
    import argparse
    import time
    import numpy as np
    import torch
    import bench
    import jumanpp_amd as J
    a = argparse.Namespace(dict_entries=300000, weights_exp=int(sys.argv[1]), seed=20260925, sent_len=40, batch=65536, rnn=True,
                           rnn_hidden=128, rnn_vocab=30000)
    cache = os.path.join(tempfile.gettempdir(), 'jppgpu_bench_cache')
    mdic, model, img = bench.make_workload(a, cache)
    corpus = bench.make_corpus(a, mdic, cache, a.batch * 2, a.seed + 1 + a.dict_entries % 7)
    batches = bench.load_batches(corpus, a.batch, np)
    dev = torch.device('cuda', 0)
    stream = torch.cuda.current_stream().cuda_stream
    ctx = J.Context(img, beam=5, global_beam=6, right_check=1, right_beam=5, device=0)
    d = [(torch.frombuffer(bytearray(tx), dtype=torch.uint8).to(dev), torch.from_numpy(of.astype(np.int32)).to(dev), len(of) - 1, len(tx))
         for tx, of in batches]
    for i in range(6):
        tt, oo, n, nbytes = d[i % 2]
        t0 = time.perf_counter()
        r = ctx.analyze_device(tt.data_ptr(), oo.data_ptr(), n, nbytes, stream)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        km = ctx.timings()
        r.release()
        t3 = time.perf_counter()
        print('step %d: enqueue %.2f ms, wait %.2f ms, release %.2f ms, kernel total %.2f' % (i, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, km['total']))


if __name__ == '__main__':
    main()
