"""Result gather for sentence-sharded multi-GPU runs.

Sentences are independent, so the data path has no collective.  The only
exchange is the gather of packed top-1 results to rank 0: an all-gather of the
per-rank sizes, then one gather of the padded payloads (RCCL over xGMI when the
tensors live on the GPUs and the backend is "nccl"; gloo on CPU in tests).
Sentence order is preserved because shards are contiguous blocks.
"""
import torch
import torch.distributed as dist

_bufs = {}  # (device, world, length) -> receive buffers of rank `dst`, reused across calls


def gather_packed(offsets, items, dst=0):
    """offsets: int32 [n+1] exclusive scan, items: int32 [m, 2] (8-byte morpheme records).
    Returns on `dst` a list of (offsets, items) per rank in rank order, None elsewhere.  The returned
    tensors are views of receive buffers that the next call reuses."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    dev = offsets.device
    n = offsets.numel() - 1
    m = int(offsets[-1].item())
    sizes = torch.tensor([n, m], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    max_n = max(int(x[0]) for x in all_sizes)
    max_m = max(int(x[1]) for x in all_sizes)
    length = max_n + 1 + 2 * max_m
    payload = torch.empty(length, dtype=torch.int32, device=dev)
    payload[:n + 1] = offsets
    payload[max_n + 1:max_n + 1 + 2 * m] = items[:m].reshape(-1)
    bufs = None
    if rank == dst:
        key = (str(dev), world)
        bufs = _bufs.get(key)
        if bufs is None or bufs[0].numel() < length:
            cap = length + length // 8  # slack: the per-batch sizes vary a little
            bufs = _bufs[key] = [torch.empty(cap, dtype=torch.int32, device=dev) for _ in range(world)]
        bufs = [b[:length] for b in bufs]
    dist.gather(payload, bufs, dst=dst)
    if rank != dst:
        return None
    out = []
    for r in range(world):
        rn, rm = int(all_sizes[r][0]), int(all_sizes[r][1])
        out.append((bufs[r][:rn + 1], bufs[r][max_n + 1:max_n + 1 + 2 * rm].reshape(-1, 2)))
    return out


def gather_packed_fixed(offsets, items, dst=0):
    """The same gather without a host sync: every rank contributes its WHOLE packed buffer -- offsets int32 [n+1] and
    the items array at its fixed capacity int32 [cap, 2], both the same shape on every rank (the bench's case: equal
    batch sizes) -- so no size has to cross to the host first.  What is sent beyond the used prefix is padding
    (the caller chooses the prefix of the items array that every rank sends: bench.py takes the largest morpheme count
    any of its batches packs, found before the timed region -- 13 MB per rank and step for 65 536 sentences).
    Returns on `dst` a list of (offsets, items) per rank; items[:offsets[-1]] is the valid part."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    dev = offsets.device
    n1 = offsets.numel()
    flat = items.reshape(-1)
    length = n1 + flat.numel()
    key = ('fixed', str(dev), world, length)
    state = _bufs.get(key)
    if state is None:
        state = _bufs[key] = {'send': torch.empty(length, dtype=torch.int32, device=dev),
                              'recv': [torch.empty(length, dtype=torch.int32, device=dev) for _ in range(world)]
                              if rank == dst else None}
    send = state['send']
    send[:n1].copy_(offsets)
    send[n1:].copy_(flat)
    dist.gather(send, state['recv'], dst=dst)
    if rank != dst:
        return None
    return [(b[:n1], b[n1:].reshape(-1, 2)) for b in state['recv']]


def shard_range(n, rank, world):
    """contiguous block partition by sentence index"""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)
