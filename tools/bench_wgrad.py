"""Time the weight-gradient products of the config-3 / config-5 training step (diagnostic): dW = X^T dZ over all
(step, sequence) rows, fp32 and bf16 operands.  FOV_GEMM_SPLIT / FOV_GEMM_BF16_SPLIT override the row split."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longterm360fov_amd import ops  # noqa: E402


def timeit(fn, n=30):
    for _ in range(5):
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
    sc = ops.Scratch()
    for rows, In, Out in ((5120, 256, 1024), (5120, 512, 1024), (30720, 256, 1024), (5120, 90, 1024)):
        x = torch.randn(rows, In, device=dev)
        dz = torch.randn(rows, Out, device=dev)
        W = torch.randn(In, Out, device=dev)
        dW = torch.empty(In, Out, device=dev)
        for dt in ("f32", "bf16"):
            for env, vals in (("FOV_GEMM_SPLIT" if dt == "f32" else "FOV_GEMM_BF16_SPLIT", os.environ.get("SPLITS", "0,8,16,32").split(",")),):
                for v in vals:
                    if v != "0":
                        os.environ[env] = v
                    else:
                        os.environ.pop(env, None)
                    us = timeit(lambda: ops.dense_bwd(x, W, dz, dW=dW, need_dx=False, need_db=False, scratch=sc, dtype=dt))
                    print("%-4s rows %5d  %4d x %4d  split %-2s : %6.1f us  %6.1f TFLOP/s" %
                          (dt, rows, In, Out, v, us, 2.0 * rows * In * Out / us / 1e6), flush=True)
                os.environ.pop(env, None)


if __name__ == "__main__":
    main()
