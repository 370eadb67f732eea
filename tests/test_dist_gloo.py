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
        import numpy as np
        np.random.seed(100 + rank)                       # the ranks' own random states differ ...
        idx = np.arange(57)
        np.random.shuffle(idx)
        idx = parallel.broadcast_index(idx)              # ... rank 0's permutation wins (fit(shuffle=True) under DP)
        q.put((rank, lo, hi, float(flat[0]), float(flat.std()), mx, bool((full[:, 0] == torch.arange(n)).all()), idx.tolist()))
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
    (r0, lo0, hi0, m0, s0, mx0, ok0, idx0), (r1, lo1, hi1, m1, s1, mx1, ok1, idx1) = res
    import numpy as np
    np.random.seed(100)
    want = np.arange(57)
    np.random.shuffle(want)
    assert idx0 == idx1 == want.tolist()                        # both ranks slice rank 0's permutation
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


def test_bench_gpus_flag_starts_the_ranks():
    """`python bench.py --gpus 2` on its own launches two ranks (torch.distributed.run, 127.0.0.1) and rank 0 reports
    n_gpus = 2; under a launcher a mismatching --gpus is refused.  --dry-run: launch plumbing only, no GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    assert json.loads(line)["n_gpus"] == 2
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env2,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 2
