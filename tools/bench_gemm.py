"""Time the fp32 MFMA GEMM on the weight-gradient shapes of the training step (diagnostic).
usage: python tools/bench_gemm.py   (FOV_GEMM_SPLIT=n overrides the split-K factor)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longterm360fov_amd import ops  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    B, T, H = 1024, 30, 256
    hs = torch.randn(B, T, H, device=dev)
    dz = torch.randn(B, T, 4 * H, device=dev)
    R = torch.randn(H, 4 * H, device=dev) * 0.05
    res = torch.randn(B, T, 5 * H, device=dev).sigmoid()
    x = torch.randn(B, T, 90, device=dev)
    K = torch.randn(90, 4 * H, device=dev) * 0.05
    for split in os.environ.get("SPLITS", "0").split(","):
        if split != "0":
            os.environ["FOV_GEMM_SPLIT"] = split
        # NN product of the same size as dR: (256 x 30720) . (30720 x 1024)
        a = hs.reshape(B * T, H).t().contiguous()
        us = timeit(lambda: ops.matmul(a, dz.reshape(B * T, 4 * H)))
        fl = 2.0 * H * 4 * H * B * T
        print("split %s  NN 256x30720x1024: %.1f us  %.1f TFLOP/s" % (split, us, fl / us / 1e6), flush=True)
    os.environ.pop("FOV_GEMM_SPLIT", None)
    # per-step projections of the step-wise decoders: (B x H) . (H x 4H), and a large NN product
    for (M, Kd, N) in ((512, 256, 1024), (4096, 256, 1024), (30720, 256, 1024), (8192, 1024, 1024)):
        a = torch.randn(M, Kd, device=dev)
        b = torch.randn(Kd, N, device=dev)
        us = timeit(lambda: ops.matmul(a, b))
        print("NN %dx%dx%d: %.1f us  %.1f TFLOP/s" % (M, Kd, N, us, 2.0 * M * Kd * N / us / 1e6), flush=True)


if __name__ == "__main__":
    main()
