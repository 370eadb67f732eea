#!/usr/bin/env python3
"""Tile-pair kernel (lstm_pair.hip) (FOV_PAIR=1) against the four-workgroup kernel: event-timed fused seq2seq calls at several
phase lengths and batches -> where the time of each goes (encoder steps / decoder steps / fixed part)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from longterm360fov_amd import ops  # noqa: E402
from oracle import fov_oracle as O  # noqa: E402


def time_call(fn, iters=60):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    H = 256
    w = O.init_seq2seq(1234, 90, 6, H, bias_noise=0.05)
    dw = {k: torch.from_numpy(v).cuda() for k, v in w.items()}
    shapes = [(1024, 30, 30), (1024, 30, 1), (1024, 60, 1), (1024, 1, 30), (1024, 1, 60), (1024, 1, 1), (528, 30, 30)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
    for B, T_in, T_out in shapes:
        enc, dec0, _ = O.synthetic_batch(7, B, T_in, T_out)
        d_enc, d_dec0 = torch.from_numpy(enc).cuda(), torch.from_numpy(dec0).cuda()
        res = {}
        for name, env in (("pair", "1"), ("four", None)):
            if env:
                os.environ["FOV_PAIR"] = env
            else:
                os.environ.pop("FOV_PAIR", None)
            ws = ops.Workspace()
            out = torch.empty((B, T_out, 6), device="cuda")
            res[name] = time_call(lambda: ops.seq2seq_decode(d_enc, d_dec0, dw, T_out, workspace=ws, out=out))
            ws.check()
        os.environ.pop("FOV_PAIR", None)
        print("B=%4d T %2d->%2d: pair %7.1f us, four-workgroup %7.1f us  (%.3f)" % (B, T_in, T_out, res["pair"], res["four"], res["pair"] / res["four"]))


if __name__ == "__main__":
    main()
