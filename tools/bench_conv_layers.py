"""Per-layer timing of the ConvLSTM cell convolutions at the config 4 shapes (B x 36 x 18 maps, 5x5, one convolution over
[x | h] per cell step) and of the three head convolutions; TFLOP/s per launch from HIP events."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from longterm360fov_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    B, H, W = a.batch, 36, 18
    shapes = [("enc0 [32|32]->128", 32, 32, 128), ("enc1 [32|16]->64", 32, 16, 64), ("enc2 [16|8]->32", 16, 8, 32),
              ("head0 56->512", 56, 0, 512), ("head1 512->1024", 512, 0, 1024), ("head2 1024->30", 1024, 0, 30)]
    out = []
    for name, c1, c2, n in shapes:
        x1 = torch.rand((B, H, W, c1), device="cuda")
        x2 = torch.rand((B, H, W, c2), device="cuda") if c2 else None
        w = torch.rand((5, 5, c1 + c2, n), device="cuda") * 0.01
        b = torch.zeros(n, device="cuda")
        y = torch.empty((B, H, W, n), device="cuda")
        f = (lambda: ops.conv2d_cat(x1, x2, w, b, out=y)) if c2 else (lambda: ops.conv2d(x1, w, b, activation="relu", out=y))
        f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        flop = 2.0 * 25 * (c1 + c2) * n * B * H * W
        out.append({"layer": name, "ms": round(ms, 4), "tflops": round(flop / ms / 1e9, 1)})
    # the ConvLSTM2D step itself (convolution + gates + state update in one launch): LDS-patch form and implicit-GEMM form
    for name, c1, F in (("cell0 [32|32] F=32", 32, 32), ("cell1 [32|16] F=16", 32, 16), ("cell2 [16|8] F=8", 16, 8)):
        x = torch.rand((B, H, W, c1), device="cuda")
        hp = torch.rand((B, H, W, F), device="cuda")
        w = torch.rand((5, 5, c1 + F, 4 * F), device="cuda") * 0.01
        b = torch.zeros(4 * F, device="cuda")
        c = torch.zeros((B, H, W, F), device="cuda")
        h = torch.empty((B, H, W, F), device="cuda")
        for tag, env in (("patch", None), ("igemm", "1")):
            if env:
                os.environ["FOV_NO_CELL_PATCH"] = env
            else:
                os.environ.pop("FOV_NO_CELL_PATCH", None)
            f = lambda: ops.convlstm_cell(x, hp, w, b, c, h, "hard_sigmoid")
            f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                f()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.reps
            flop = 2.0 * 25 * (c1 + F) * 4 * F * B * H * W
            out.append({"layer": name + " " + tag, "ms": round(ms, 4), "tflops": round(flop / ms / 1e9, 1)})
        os.environ.pop("FOV_NO_CELL_PATCH", None)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
