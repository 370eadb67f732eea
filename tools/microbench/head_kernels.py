"""Device time of the fused head kernels of mlp_head.hip alone (mixture head, batch 32, H = 400 padded to 512), warm (weights resident
in the caches from the previous call) and cold (every weight rewritten by an elementwise kernel on all CUs in between, as the optimizer
does in a training step).  HIP events around N back-to-back calls; the host side is pre-marshalled so it stays ahead of the device.
usage: python3 tools/microbench/head_kernels.py [N]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from longterm360fov_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B, H, n = 32, 512, 20
rng = np.random.default_rng(0)
dims = [H, 64, 128, 256, 10 * n]
flat = torch.empty(sum(a * b + b for a, b in zip(dims[:-1], dims[1:])), device="cuda")
flat.copy_(torch.from_numpy((0.05 * rng.standard_normal(flat.numel())).astype(np.float32)))
layers, off = [], 0
for l in range(4):
    W = flat[off:off + dims[l] * dims[l + 1]].view(dims[l], dims[l + 1]); off += W.numel()
    b = flat[off:off + dims[l + 1]]; off += b.numel()
    layers.append((W, b, "relu" if l < 3 else None))
gW = [torch.zeros_like(W) for W, _, _ in layers]
gb = [torch.zeros_like(b) for _, b, _ in layers]
h = torch.from_numpy(rng.standard_normal((B, H)).astype(np.float32)).cuda()
y = torch.from_numpy(rng.uniform(-1, 1, (B, 10, 90)).astype(np.float32)).cuda()
acts = ops.mlp_head_fwd(h, layers, n_mix=n)
loss, dpre = ops.gmm3d_loss_grad(acts[-1], y, 30, 1.0 / (B * 300))
sc = ops.Scratch()


def timed(fn, cold):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    if cold:      # time the rewrite alone, then subtract
        e0.record()
        for _ in range(N):
            flat.mul_(1.0)
        e1.record(); torch.cuda.synchronize()
        base = e0.elapsed_time(e1) / N
        e0.record()
        for _ in range(N):
            flat.mul_(1.0)
            fn()
        e1.record(); torch.cuda.synchronize()
        return (e0.elapsed_time(e1) / N - base) * 1e3
    e0.record()
    for _ in range(N):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N * 1e3


cases = {"head forward (1 launch)": lambda: ops.mlp_head_fwd(h, layers, n_mix=n),
         "mixture loss + gradient (1 launch)": lambda: ops.gmm3d_loss_grad(acts[-1], y, 30, 1.0 / (B * 300)),
         "head backward (chain + weight gradients, 2 launches)": lambda: ops.mlp_head_bwd(h, layers, acts, dpre, gW, gb, scratch=sc),
         "head backward without dx... (weight gradients + 3-layer chain)": lambda: ops.mlp_head_bwd(h, layers, acts, dpre, gW, gb, need_dx=False, scratch=sc)}
for name, fn in cases.items():
    print("%-62s warm %6.1f us   cold %6.1f us" % (name, timed(fn, False), timed(fn, True)))
