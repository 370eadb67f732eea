#!/usr/bin/env python3
"""fov_conv2d_wgrad at the ConvLSTM model's layer shapes (configs[3]: 256 sequences x 10 steps of 36 x 18 maps, k = 5): time per
call and fraction of the fp32 matrix peak.  FOV_NO_WGRAD_LINES=1 selects the tap-wise kernel for every layer (read once per process)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longterm360fov_amd import ops

maps = int(os.environ.get("PROBE_MAPS", "2560"))
H, W, k = 36, 18, 5
shapes = [(30, 128, 32), (32, 128, 32), (32, 64, 32), (16, 64, 16), (16, 32, 16), (8, 32, 8), (56, 512, 56), (1024, 30, 1024)]
if os.environ.get("PROBE_BIG"):
    shapes = [(512, 1024, 512)]
sc = ops.Scratch()
tot = 0.0
for C, N, ldx in shapes:
    g = torch.Generator(device="cuda"); g.manual_seed(C * 1000 + N)
    x = torch.rand((maps, H, W, ldx), device="cuda", generator=g) - 0.5
    dy = torch.rand((maps, H, W, N), device="cuda", generator=g) - 0.5
    xs = x[..., :C]
    dw = torch.empty((k, k, C, N), device="cuda")
    for _ in range(2):
        ops.conv2d_wgrad(xs, dy, k, k, dw=dw, scratch=sc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        ops.conv2d_wgrad(xs, dy, k, k, dw=dw, scratch=sc)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * maps * H * W * k * k * C * N
    # spot check against a float64 product over a few maps
    nb = 3
    ref = torch.zeros((k, k, C, N), dtype=torch.float64)
    xp = torch.nn.functional.pad(xs[:nb].double().cpu(), (0, 0, 2, 2, 2, 2))
    dd = dy[:nb].double().cpu()
    for i in range(k):
        for j in range(k):
            ref[i, j] = torch.einsum("bhwc,bhwn->cn", xp[:, i:i + H, j:j + W], dd)
    got = ops.conv2d_wgrad(xs[:nb], dy[:nb], k, k, scratch=ops.Scratch()).double().cpu()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    tot += ms
    print("C %4d -> N %4d: %8.3f ms  %6.1f TFLOP/s (%.2f of 157.3)   rel err (3 maps) %.1e" % (C, N, ms, fl / ms * 1e-9, fl / ms * 1e-9 / 157.3, err), flush=True)
print("sum %.2f ms" % tot)
