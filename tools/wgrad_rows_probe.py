"""Times fov_lstm_seq_wgrad_pair at lstm.py's shape (two stacked layers, 32 x 10 rows, H = 512) and at config 1's widths
(HIP events around back-to-back calls; run under `rocprofv3 --kernel-trace --stats` for the kernels' own durations).
FOV_NO_WGRAD_GROUP=1: the split products + reduces the one-launch kernels replaced."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longterm360fov_amd import ops

def run(B, T, F1, F2, H, group_off=False):
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    r = lambda *s: torch.rand(s, device="cuda", generator=g) - 0.5
    x1, hs1, dz1, x2, hs2, dz2, h0 = r(B, T, F1), r(B, T, H), r(B, T, 4 * H), r(B, T, F2), r(B, T, H), r(B, T, 4 * H), r(B, H)
    out = [torch.zeros(F1, 4 * H, device="cuda"), torch.zeros(H, 4 * H, device="cuda"), torch.zeros(4 * H, device="cuda"),
           torch.zeros(F2, 4 * H, device="cuda"), torch.zeros(H, 4 * H, device="cuda"), torch.zeros(4 * H, device="cuda")]
    sc = ops.Scratch()
    call = lambda: ops.lstm_seq_wgrad_pair((x1, hs1, h0, dz1) + tuple(out[:3]), (x2, hs2, h0, dz2) + tuple(out[3:]), scratch=sc)
    for _ in range(20): call()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): call()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
    fl = 2.0 * B * T * 4 * H * (F1 + F2 + 2 * H)
    print("%s  B %d T %d F %d/%d H %d: %.1f us per call (%.1f TFLOP/s)" % ("split products" if os.environ.get("FOV_NO_WGRAD_GROUP") else "one launch", B, T, F1, F2, H, best, fl / best * 1e-6), flush=True)

run(32, 10, 90, 512, 512)
run(32, 10, 90, 6, 128)
run(32, 10, 90, 6, 256)
