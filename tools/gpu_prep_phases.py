#!/usr/bin/env python3
"""(GPU box, developer tool) phase shares of k_rnn_prep on the configs[4] shape from a
-DJPP_DEV_PROF build:  hipcc ... -DJPP_DEV_PROF ... -o build/libjppgpu_prof.so ; python tools/gpu_prep_phases.py"""
import ctypes
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import jumanpp_amd as J

lib_path = os.path.join(ROOT, 'build', 'libjppgpu_prof.so')
HEAD = '--headline' in sys.argv   # bench.py's default workload instead of the configs[4] shape
args = bench.build_parser().parse_args([])
if not HEAD:
    args.sent_len, args.batch = 220, 16384
cache = os.path.join(tempfile.gettempdir(), 'jppgpu_bench_cache')
mdic, model, img = bench.make_workload(args, cache)
corpus = bench.make_corpus(args, mdic, cache, args.batch * 2, args.seed + 1 if HEAD else 31)
batches = bench.load_batches(corpus, args.batch, np)
cfg = {} if HEAD else dict(beam=32, global_beam=32, right_check=1, right_beam=32)
ctx = J.Context(img, lib_path=lib_path, use_rnn=True, **cfg)
lib = ctypes.CDLL(lib_path)
dev = torch.device('cuda', 0)
text, offs = batches[0]
t = torch.frombuffer(bytearray(text), dtype=torch.uint8).to(dev)
o = torch.from_numpy(offs.astype(np.int32)).to(dev)
buf = (ctypes.c_ulonglong * 8)()
for it in range(3):
    r = ctx.analyze_device(t.data_ptr(), o.data_ptr(), len(offs) - 1, len(text), None)
    ms = ctx.timings()
    r.release()
    lib.jppgpu_debug_prep_prof(buf)
vals = [buf[i] for i in range(6)]
names = ['A first occurrences (list)', 'A resolve vocabulary ids', 'B round: rows arrive', 'B round: groups + nodes', 'B round: stores', 'dense row offsets']
print('rnn ms', ms['rnn'])
for n_, v in zip(names, vals):
    print('%-30s %6.2f %%  %8.0f cycles per sentence' % (n_, 100.0 * v / max(1, sum(vals)), v / float(args.batch)))
