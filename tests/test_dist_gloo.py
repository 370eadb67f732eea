"""world_size-2 gloo (CPU) coverage of the N>1 plumbing: sharding, the flat gradient all-reduce,
the max-over-ranks timing reduction and the shard gather."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from longterm360fov_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world_size, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        n = 1025
        lo, hi = parallel.shard_range(n)
        flat = torch.full((1000,), float(rank + 1))
        parallel.allreduce_mean_(flat)
        mx = parallel.max_over_ranks(0.5 + rank)
        rows = torch.arange(lo, hi, dtype=torch.float32)[:, None].repeat(1, 3)
        full = parallel.gather_rows(rows, n)
        q.put((rank, lo, hi, float(flat[0]), float(flat.std()), mx, bool((full[:, 0] == torch.arange(n)).all())))
    finally:
        dist.destroy_process_group()


def test_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, m0, s0, mx0, ok0), (r1, lo1, hi1, m1, s1, mx1, ok1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 513, 513, 1025)          # contiguous cover, sizes differ by <= 1
    assert m0 == m1 == 1.5 and s0 == s1 == 0.0                  # mean of the flat buffer
    assert mx0 == mx1 == 1.5                                    # max over ranks
    assert ok0 and ok1                                          # gather restores the global order


def test_single_process_defaults():
    assert parallel.world() == (0, 1)
    assert parallel.shard_range(10) == (0, 10)
    assert parallel.shard_range(10, 2, 3) == (7, 10)
    assert [parallel.shard_range(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    t = torch.ones(4)
    assert parallel.allreduce_mean_(t) is t
    assert parallel.max_over_ranks(2.5) == 2.5
