#!/usr/bin/env python3
"""Secondary measurement for BASELINE.json configs[3]: ConvLSTM seq2seq on 36x18x30 heat maps, filters
32/16/8, k = 5, T 10 -> 10.  Reports the ConvLSTM CELL throughput (3-layer encoder over T_in steps: exactly
half of the cell FLOPs of the model, no head involved) and, with --whole, the full model including the
Conv2D 56->512->1024->30 head (the head is ~50x the cell's FLOPs)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from longterm360fov_amd import ops  # noqa: E402
from longterm360fov_amd.models import ConvLSTMSeq2Seq  # noqa: E402
from oracle import fov_oracle as O  # noqa: E402  (Keras initialisers only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--whole", action="store_true")
    ap.add_argument("--no-pad-input", action="store_true", help="cell bench: keep 30 input channels (scalar gather) "
                    "instead of the zero-padded 32 the model object uses")
    ap.add_argument("--two-launch", action="store_true", help="cell bench: convolution and gates as two launches (round-1 form)")
    ap.add_argument("--train", action="store_true", help="also time one training step (fwd + backward + RMSprop)")
    ap.add_argument("--train-batch", type=int, default=64)
    a = ap.parse_args()
    H, W, C, T = 36, 18, 30, 10
    w = O.init_convlstm_seq2seq(1, C=C, latent_dim=16, head="conv2d")
    dw = {k: torch.from_numpy(v).cuda() for k, v in w.items()}
    B = a.batch
    x = torch.rand((B, T, H, W, C), device="cuda")
    filters = (32, 16, 8)
    cin = (C,) + filters[:2]
    cell_flop_step = sum(2 * 25 * (ci + f) * 4 * f for ci, f in zip(cin, filters)) * H * W   # per sequence-step

    kr = [torch.cat([dw["enc%d_K" % l], dw["enc%d_R" % l]], 2).contiguous() for l in range(3)]
    x0 = x
    if not a.no_pad_input:   # as ConvLSTMSeq2Seq.predict does: zero channels 30, 31 in x and zero rows in K: same result, vector gather path
        x = torch.cat([x, torch.zeros((B, T, H, W, 2), device="cuda")], -1).contiguous()
        K0 = torch.cat([dw["enc0_K"], torch.zeros((5, 5, 2, 4 * filters[0]), device="cuda")], 2)
        kr[0] = torch.cat([K0, dw["enc0_R"]], 2).contiguous()

    def encoder():
        seq = [x[:, t] for t in range(T)]
        for l, F in enumerate(filters):
            h = torch.zeros((B, H, W, F), device="cuda")
            c = torch.zeros((B, H, W, F), device="cuda")
            KR, b = kr[l], dw["enc%d_b" % l]
            nxt = []
            for t in range(T):
                hn = torch.empty((B, H, W, F), device="cuda")
                if a.two_launch:
                    z = ops.conv2d_cat(seq[t], h, KR, b)     # conv(x_t, K) + conv(h, R) + b in one launch, gates in a second
                    ops.convlstm_gates(z, c, hn, "hard_sigmoid")
                else:
                    ops.convlstm_cell(seq[t], h, KR, b, c, hn, "hard_sigmoid")   # the whole step in one launch
                h = hn
                nxt.append(h)
            seq = nxt

    encoder()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        encoder()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    res = {"workload": "configs[3] ConvLSTM cell only: 3-layer encoder, T=10, 36x18x30, B=%d, fp32" % B,
           "ms": dt * 1e3, "sequences_per_s": B / dt, "cell_tflops": cell_flop_step * T * B / dt / 1e12,
           "frac_of_fp32_mfma_peak": cell_flop_step * T * B / dt / 1e12 / 157.3}
    if a.whole:
        m = ConvLSTMSeq2Seq(w, head="conv2d")
        xe = x0.cpu().numpy()
        m.predict([xe[:8], xe[:8, -1:]], predict_step=1)
        t0 = time.perf_counter()
        m.predict([xe, xe[:, -1:]], predict_step=T)
        dt = time.perf_counter() - t0
        head_flop = 2 * 25 * (56 * 512 + 512 * 1024 + 1024 * 30) * H * W
        tot = (2 * cell_flop_step + head_flop) * T * B
        res.update({"whole_model_ms": dt * 1e3, "whole_model_sequences_per_s": B / dt, "whole_model_tflops": tot / dt / 1e12})
    if a.train:
        from longterm360fov_amd.training import ConvLSTMTrainer
        Bt = a.train_batch
        tr = ConvLSTMTrainer(w, head="conv2d")
        xe = x0[:Bt].contiguous()
        tgt = torch.softmax(torch.rand((Bt, T, H, W, C), device="cuda"), -1)
        tr.train_step(xe, xe[:, -1:], tgt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            loss = tr.train_step(xe, xe[:, -1:], tgt)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        head_flop = 2 * 25 * (56 * 512 + 512 * 1024 + 1024 * 30) * H * W
        tot = 3 * (2 * cell_flop_step + head_flop) * T * Bt
        res.update({"train_batch": Bt, "train_step_ms": dt * 1e3, "train_sequences_per_s": Bt / dt,
                    "train_tflops_3x_forward": tot / dt / 1e12, "train_loss": float(loss.item())})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
