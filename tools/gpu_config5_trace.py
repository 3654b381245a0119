#!/usr/bin/env python3
"""(GPU box, developer tool) the BASELINE configs[4] shape alone (beam = gbeam = rbeam = 32, 220-codepoint sentences,
perceptron + RNNLM), a few steps of the shipped library: the command to put under
   rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c5 -o trace -- python tools/gpu_config5_trace.py
(tools/gpu_session.sh TAG trace5 does that and prints the per-kernel table)."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import jumanpp_amd as J

args = bench.build_parser().parse_args([])
args.sent_len, args.batch = 220, 16384
cache = os.path.join(tempfile.gettempdir(), 'jppgpu_bench_cache')
mdic, model, img = bench.make_workload(args, cache)
corpus = bench.make_corpus(args, mdic, cache, args.batch * 2, 31)
batches = bench.load_batches(corpus, args.batch, np)
ctx = J.Context(img, use_rnn=True, beam=32, global_beam=32, right_check=1, right_beam=32)
dev = torch.device('cuda', 0)
text, offs = batches[0]
t = torch.frombuffer(bytearray(text), dtype=torch.uint8).to(dev)
o = torch.from_numpy(offs.astype(np.int32)).to(dev)
for it in range(4):
    r = ctx.analyze_device(t.data_ptr(), o.data_ptr(), len(offs) - 1, len(text), None)
    ms = ctx.timings()
    r.release()
print({k: round(v, 3) for k, v in ms.items() if isinstance(v, float)})
