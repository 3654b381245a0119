"""world_size=2 gloo test (CPU) of the sentence-sharded multi-GPU path: each
rank analyses its contiguous shard (emulator library), packs the top-1
morphemes, rank 0 gathers and the concatenation must equal the single-process
result."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
import jumanpp_amd as J
from jumanpp_amd.dist import gather_packed, gather_packed_fixed, shard_range
dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
lines = [l.rstrip('\n') for l in open(%(txt)r, encoding='utf-8')]
ctx = J.Context(%(img)r, lib_path=%(lib)r)

def packed(sub):
    res = ctx.analyze(sub)
    offs = np.zeros(len(sub) + 1, dtype=np.int32)
    items = np.zeros((4096, 2), dtype=np.int32)
    res.pack(offs.ctypes.data, items.ctypes.data, items.shape[0])
    res.fetch()   # emulator is synchronous; a fetch orders after the pack on a GPU
    return torch.from_numpy(offs), torch.from_numpy(items)

lo, hi = shard_range(len(lines), rank, world)
offs, items = packed(lines[lo:hi])
got = gather_packed(offs, items, dst=0)
if rank == 0:
    full_offs, full_items = packed(lines)
    cat_items = torch.cat([g[1] for g in got])
    cat_counts = torch.cat([g[0][1:] - g[0][:-1] for g in got])
    assert torch.equal(cat_counts, full_offs[1:] - full_offs[:-1]), 'per-sentence morpheme counts differ'
    m = int(full_offs[-1])
    assert torch.equal(cat_items, full_items[:m]), 'gathered morphemes differ'
    assert m > 100
    print('DIST_OK', m)
# the sync-free variant (equal shapes on every rank: the shards are padded to the same number of sentences)
per = (len(lines) + world - 1) // world
sub = lines[lo:hi] + [''] * (per - (hi - lo))
offs2, items2 = packed(sub)
got2 = gather_packed_fixed(offs2, items2, dst=0)
if rank == 0:
    cat2 = torch.cat([g[1][:int(g[0][-1])] for g in got2])
    assert torch.equal(cat2, full_items[:m]), 'fixed-capacity gather differs'
    print('DIST_FIXED_OK')
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_sharded_gather_matches_single_process(emu_lib, golden_dir, tmp_path):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % dict(root=ROOT, txt=os.path.join(golden_dir, 'mini.txt'),
                                    img=os.path.join(golden_dir, 'mini.img'), lib=emu_lib))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert 'DIST_OK' in outs[0] and 'DIST_FIXED_OK' in outs[0], outs


def test_bench_world_2_control_flow_on_the_emulator(emu_lib, ref_tools, tmp_path):
    """bench.py exactly as the driver launches it for N = 2 (torch.distributed.run, one rank per device), walked on the
    kernel emulator with gloo (JPPGPU_BENCH_EMU=1, toy sizes): rank 0 builds the model while rank 1 waits behind the
    barrier, both analyse their own shard, the packed results are gathered to rank 0 inside the timed loop, the elapsed
    time is the maximum over the ranks, rank 0 certifies its shard against the reference and times the CPU baseline while
    rank 1 waits at the final barrier -- and ONE JSON line comes out, carrying n_gpus = 2, roofline, cpu_baseline and a
    parity_sample without mismatches.  (It measures nothing: the line says so in `data`.)"""
    import json
    import pytest
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cache = str(tmp_path / 'cache')
    env = dict(os.environ, JPPGPU_BENCH_EMU='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--batch', '48', '--dict-entries', '3000', '--weights-exp', '16', '--rnn-hidden', '32', '--rnn-vocab', '500',
           '--cpu-sample', '48', '--parity-batches', '1', '--cache', cache]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, (lines, p.stderr.decode()[-1000:])
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 2 and d['warmup'] == 1 and d['scaling'] == 'weak' and d['value'] > 0
    assert 'EMULATOR' in d['data']
    assert d['config']['parallelism'].startswith('sentence-sharded x2')
    assert d['roofline']['bound'] == 'hbm' and d['roofline']['algorithmic_bytes_per_launch'] > 0   # (the emulator has no clock: achieved = 0)
    assert d['cpu_baseline']['kind'] == 'reference' and d['cpu_baseline']['value'] > 0
    assert d['parity_sample']['mismatches'] == 0 and d['parity_sample']['sentences'] == 48, d['parity_sample']
    assert d['batches']['one_enqueue_batches'] >= 2 and d['batches']['one_enqueue_overflows'] == 0, d['batches']

