"""Multi-GPU plumbing of the path: one process per GPU, torch.distributed (backend "nccl" is RCCL
on ROCm; "gloo" on CPU for tests).  The path shards by sequence - inference is pure replication
with no data-path collective; training adds ONE sum all-reduce of the flat gradient buffer per
step.  Nothing here computes on the data path."""
import torch
import torch.distributed as dist


import os


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def dp_active():
    """True when the data-parallel branch (all-reduce of the flat gradient buffer, broadcast of the shuffle index) is to
    run: more than one rank, or FOV_FORCE_DIST=1 with an initialised process group of ANY size.  The second form lets a
    box with one GPU drive the real RCCL path (communicator, stream ordering, async work handles) at world size 1
    (tests/test_gpu_dist_nccl.py, bench.py --force-dist)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("FOV_FORCE_DIST", "") == "1"


def shard_range(n, rank=None, world_size=None):
    """Rank-contiguous [lo, hi) slice of n sequences: sizes differ by at most one, concatenation over
    ranks is range(n), so 1-GPU and G-GPU runs see the same global batch."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    base, extra = divmod(int(n), world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def allreduce_mean_(flat):
    """In-place mean over ranks of one contiguous buffer (the whole model's gradients)."""
    _, w = world()
    if w > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(w)
    return flat


def max_over_ranks(value, device=None):
    """Max of a python float over ranks (timed-region reduction of bench.py)."""
    _, w = world()
    if w == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_rows(local, n_total, device=None):
    """All-gather rank-contiguous row shards (tensor (n_local, ...)) back into (n_total, ...)."""
    _, w = world()
    if w == 1:
        return local
    sizes = [shard_range(n_total, r, w) for r in range(w)]
    maxn = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((maxn,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(w)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=0)


def broadcast_index(idx):
    """Rank 0's index permutation on every rank (fit(shuffle=True) under data parallelism)."""
    import numpy as np
    if not dp_active():
        return idx
    t = torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int64))
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.broadcast(t, src=0)
    return t.cpu().numpy()
