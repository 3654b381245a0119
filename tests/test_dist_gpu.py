"""The result gather through torch.distributed's nccl (= RCCL) backend on the MI355X -- single rank: no multi-GPU node is
available to the builder (the exchange between ranks: tests/test_dist_cpu.py, two ranks over gloo)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


RCCL_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
import jumanpp_amd as J
from jumanpp_amd.dist import gather_packed, gather_packed_fixed
torch.cuda.set_device(0)
dist.init_process_group('nccl')     # = RCCL on ROCm
dev = torch.device('cuda', 0)
lines = [l.rstrip('\n') for l in open(%(txt)r, encoding='utf-8')]
ctx = J.Context(%(img)r, device=0)
enc = [s.encode('utf-8') for s in lines]
text = torch.frombuffer(bytearray(b''.join(enc)), dtype=torch.uint8).to(dev)
offs_in = torch.tensor([0] + [len(e) for e in enc], dtype=torch.int64).cumsum(0).to(torch.int32).to(dev)
res = ctx.analyze_device(text.data_ptr(), offs_in.data_ptr(), len(enc), int(text.numel()), torch.cuda.current_stream().cuda_stream)
offs = torch.zeros(len(enc) + 1, dtype=torch.int32, device=dev)
items = torch.zeros((4096, 2), dtype=torch.int32, device=dev)
res.pack(offs.data_ptr(), items.data_ptr(), items.shape[0])
torch.cuda.synchronize()
m = int(offs[-1].item())
assert m > 100
# the bench's gather (fixed shapes, no host sync before it) and the sized one, through the RCCL backend on device tensors
got = gather_packed_fixed(offs, items, dst=0)
torch.cuda.synchronize()
assert len(got) == 1 and torch.equal(got[0][0], offs) and torch.equal(got[0][1][:m], items[:m])
got = gather_packed(offs, items, dst=0)
torch.cuda.synchronize()
assert len(got) == 1 and torch.equal(got[0][0], offs) and torch.equal(got[0][1], items[:m])
t = torch.ones(4, device=dev)
dist.all_reduce(t)
dist.barrier()
torch.cuda.synchronize()
print('RCCL_OK', m, torch.cuda.nccl.version() if hasattr(torch.cuda, 'nccl') else '')
dist.destroy_process_group()
'''


@pytest.mark.gpu
def test_gpu_gather_through_the_rccl_backend_single_rank(gpu_lib, golden_dir, tmp_path):
    """No multi-GPU node is available to the builder: what CAN run here is the gather of the packed results through
    torch.distributed's nccl (= RCCL) backend with ONE rank on the MI355X -- the communicator forms, the collective
    calls jumanpp_amd/dist.py makes are accepted for device tensors, and the result comes back intact.  (The exchange
    between ranks is covered by the two-rank gloo tests above.)"""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / 'rccl_worker.py'
    script.write_text(RCCL_WORKER % dict(root=ROOT, txt=os.path.join(golden_dir, 'mini.txt'), img=os.path.join(golden_dir, 'mini.img')))
    env = dict(os.environ, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0 and 'RCCL_OK' in out, out[-2000:]


@pytest.mark.gpu
def test_gpu_bench_distributed_control_flow_through_rccl_single_rank(gpu_lib, ref_tools, tmp_path):
    """bench.py's N > 1 control flow -- the barriers around the model build, the RCCL gather of the packed results inside
    the timed loop, the max over ranks, the final barrier -- on the MI355X with ONE rank (JPPGPU_BENCH_DIST1=1): the line
    says which backend gathered, and the certified sample still has no mismatch.  Small sizes: this is a test of the
    path, not a measurement."""
    import json
    if ref_tools is None:
        pytest.skip('oracle/_ref not built')
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY='0', JPPGPU_BENCH_DIST1='1')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '2', '--batch', '4096', '--dict-entries', '20000',
                        '--weights-exp', '18', '--cache', str(tmp_path / 'cache'), '--no-cli', '--no-config5', '--no-realism', '--no-trainer',
                        '--no-overlap'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = p.stdout.decode().strip().splitlines()
    assert lines and lines[-1].startswith('{'), ('stdout must end with the JSON line', lines[-5:], p.stderr.decode()[-1500:])
    d = json.loads(lines[-1])
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['value'] > 0
    assert 'nccl' in d['config']['result_gather'], d['config']
    assert d['parity_sample']['mismatches'] == 0 and d['parity_sample']['sentences'] > 0, d['parity_sample']
    assert d['cpu_baseline']['value'] > 0
