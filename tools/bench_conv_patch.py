"""The three head convolutions of configs[3] (B x 36 x 18 maps, 5 x 5) and two of their data gradients: map-resident form
(conv_patch.hip) and the tap-gathering implicit GEMM (FOV_NO_CONV_PATCH=1);
TFLOP/s per launch from HIP events.   usage: python3 tools/bench_conv_patch.py [--batch 256] [--reps 10]"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def run():
    import torch
    sys.path.insert(0, ROOT)
    from longterm360fov_amd import ops
    B, reps = int(os.environ["BCP_BATCH"]), int(os.environ["BCP_REPS"])
    H, W = 36, 18
    out = {}
    for name, c, n in (("head0 56->512", 56, 512), ("head1 512->1024", 512, 1024), ("head2 1024->30", 1024, 30),
                       ("dgrad 1024->512", 1024, 512), ("dgrad 512->56", 512, 56)):
        x = torch.rand((B, H, W, c), device="cuda")
        w = torch.rand((5, 5, c, n), device="cuda") * 0.01
        b = torch.zeros(n, device="cuda")
        y = torch.empty((B, H, W, n), device="cuda")
        f = lambda: ops.conv2d(x, w, b, activation="relu", out=y)
        f(); f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out[name] = {"ms": round(ms, 3), "tflops": round(2.0 * 25 * c * n * B * H * W / ms / 1e9, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    if os.environ.get("BCP_CHILD"):
        run()
        sys.exit(0)
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    for tag, env in (("map-resident (conv_patch.hip)", {}), ("tap-gathering implicit GEMM", {"FOV_NO_CONV_PATCH": "1"})):
        e = dict(os.environ, BCP_CHILD="1", BCP_BATCH=str(a.batch), BCP_REPS=str(a.reps), **env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=e, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print("%-42s %s" % (tag, line[-1] if line else "FAILED: " + r.stderr[-400:]))
