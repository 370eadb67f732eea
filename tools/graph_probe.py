"""Probe: does capturing one others-mixing training step in a HIP graph (torch.cuda.CUDAGraph around the
ctypes launches) pay?  Diagnostic only."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longterm360fov_amd.training import OthersMixingTrainer  # noqa: E402
from oracle import fov_oracle as O  # noqa: E402


def main():
    H, T_in, T_out, U, B = 256, 10, 10, 34, 512
    w = O.init_others_mixing(1234, H=H, num_user=U, bias_noise=0.05)
    enc, dec0, tgt, oth = O.synthetic_batch(1234, B, T_in, T_out, num_others=U - 1)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    a = [d(enc), d(oth), d(dec0), d(tgt)]
    tr = OthersMixingTrainer(w, optimizer="rmsprop")

    def timeit(fn, n=20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    for _ in range(3):
        tr.train_step(*a)
    print("eager  %.3f ms/step" % timeit(lambda: tr.train_step(*a)), flush=True)
    # host side alone: enqueue one step on an idle GPU and stop the clock before waiting for it
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_step(*a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("one step: host enqueue %.3f ms, then %.3f ms until the GPU is idle" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            tr.train_step(*a)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss = tr.train_step(*a)
    l0 = None
    for i in range(5):
        g.replay()
        torch.cuda.synchronize()
        if i == 0:
            l0 = float(loss.item())
    print("graph  %.3f ms/step   loss %.6f -> %.6f" % (timeit(g.replay), l0, float(loss.item())), flush=True)
    tr.ws.check(); tr.bwd_scratch.check()


if __name__ == "__main__":
    main()
