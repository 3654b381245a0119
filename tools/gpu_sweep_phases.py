#!/usr/bin/env python3
"""(GPU box, developer tool) phase shares of k_sweep from a -DJPP_DEV_PROF build:
   hipcc ... -DJPP_DEV_PROF ... -o build/libjppgpu_prof.so ; python tools/gpu_sweep_phases.py"""
import ctypes
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import argparse
import numpy as np
import torch
import bench
import jumanpp_amd as J

lib_path = os.path.join(ROOT, 'build', 'libjppgpu_prof.so')
C5 = '--config5' in sys.argv   # BASELINE configs[4] shape: beam 32, 220-codepoint sentences, 4096 per batch
args = bench.build_parser().parse_args([])   # bench.py's default workload
if C5:
    args.sent_len, args.batch = 220, 4096
cache = os.path.join(tempfile.gettempdir(), 'jppgpu_bench_cache')
mdic, model, img = bench.make_workload(args, cache)
corpus = bench.make_corpus(args, mdic, cache, args.batch * 2, 31 if C5 else args.seed + 1)
batches = bench.load_batches(corpus, args.batch, np)
cfg = dict(beam=32, global_beam=32, right_check=1, right_beam=32) if C5 else {}
ctx = J.Context(img, lib_path=lib_path, use_rnn=('--rnn' in sys.argv), **cfg)
lib = ctypes.CDLL(lib_path)
dev = torch.device('cuda', 0)
text, offs = batches[0]
t = torch.frombuffer(bytearray(text), dtype=torch.uint8).to(dev)
o = torch.from_numpy(offs.astype(np.int32)).to(dev)
buf = (ctypes.c_ulonglong * 16)()
for it in range(3):
    r = ctx.analyze_device(t.data_ptr(), o.data_ptr(), len(offs) - 1, len(text), None)
    ms = ctx.timings()
    r.release()
    lib.jppgpu_debug_sweep_prof(buf)
vals = [buf[i] for i in range(8)]
tot = sum(vals)
names = ['loop head + prefetch issue', '1 candidates + global beam', '2 T1 dedup + T1/T2 rows', '3 prescores',
         '4 cutoff', '5a tail bigrams', '5b cells (trigrams)', '5c beams']
print('k_sweep ms', ms['sweep'], 'rnn ms', ms['rnn'])
if '--rnn' in sys.argv:
    rv = [buf[i] for i in range(8, 14)]
    rt = sum(rv)
    rn = ['stage W + bookkeeping', 'maxent', 'ctx/nce loads, dot, score', 'matvec + sigmoid + store', 'boundary tail (cells, totals)', 'EOS beam']
    for n_, v in zip(rn, rv):
        print('rnn %-30s %6.2f %%' % (n_, 100.0 * v / max(1, rt)))
    print('rnn cycles per sentence (lane 0): %.0f' % (buf[14] / max(1, buf[15])))
if '--fine' in sys.argv:   # finer marks inside the phases (k_sweep<32,*>: 5c = A ranks | B1 half-wave replays | B2 serial replays | C)
    fine = [buf[i] for i in range(8, 14)]
    tot += sum(fine)
    names += ['fine %d' % i for i in range(8, 14)]
    vals += fine
for n_, v in zip(names, vals):
    print('%-30s %6.2f %%' % (n_, 100.0 * v / max(1, tot)))
