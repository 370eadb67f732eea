#!/usr/bin/env python3
"""Where does a step of the cluster kernel spend its cycles?  Loads the DIAGNOSTIC build
(make -C longterm360fov_amd/csrc stamps -> libfov360_hip_stamps.so), runs the bench workload
once and prints per-segment s_memtime deltas (shader cycles; clock from s_memrealtime) over the steps of
one wave (block 5, wave 0).  Shares, not absolute run time, are what to read (the stamps add
fences the shipped kernel does not have)."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from longterm360fov_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "longterm360fov_amd", "lib", "libfov360_hip_stamps.so")
from longterm360fov_amd import ops  # noqa: E402
from oracle import fov_oracle as O  # noqa: E402

SEG = ["remote-slice h.R MFMAs (192) [decoder: + sum of the Dense partials, tanh, y.K]", "cell update + publish", "(nothing)",
       "own h -> LDS [decoder: + own Dense partial, 4 MFMAs, publish] + barrier 1b",
       "x(t+1).K MFMAs (96, encoder)", "gather issue + own-slice MFMAs (64) + wait", "barrier 2"]


def main():
    B, T_in, T_out, H = 1024, 30, 30, 256
    w = O.init_seq2seq(1234, H=H, bias_noise=0.05)
    enc, dec0, _ = O.synthetic_batch(1234, B, T_in, T_out)
    dw = {k: torch.from_numpy(v).cuda() for k, v in w.items()}
    d_enc, d_dec0 = torch.from_numpy(enc).cuda(), torch.from_numpy(dec0).cuda()
    ws = ops.Workspace()
    for _ in range(3):
        ops.seq2seq_decode(d_enc, d_dec0, dw, T_out, impl="cluster", workspace=ws)
    ws.check()
    L = _lib.lib()
    buf = np.zeros((2, 64, 12), dtype=np.uint64)
    L.fov_debug_read_stamps.argtypes = [ctypes.c_void_p]
    assert L.fov_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    for mode, name, steps, nslot in ((0, "encoder (MODE_LAYER)", T_in, 8), (1, "decoder (MODE_DECODE)", T_out, 8)):
        st = buf[mode, :steps, :nslot].astype(np.int64)
        seg = np.diff(st, axis=1)                      # per step, per segment (shader cycles)
        step_total = np.diff(st[:, 0])                 # step start to next step start
        real = buf[mode, :steps, 9].astype(np.int64)   # s_memrealtime, 100 MHz
        ghz = (st[-1, 0] - st[0, 0]) / ((real[-1] - real[0]) * 10.0)
        print("== %s: %.0f cycles = %.2f us per step (median over %d steps), in-kernel clock %.2f GHz"
              % (name, np.median(step_total), np.median(step_total) / ghz * 1e-3, steps - 1, ghz))
        spins = buf[mode, :steps, 10].astype(np.int64)
        own = (buf[mode, 1:steps, 11].astype(np.int64) - st[1:, 5])
        print("   gather: extra sweeps per step: mean %.2f max %d; issue->own-MFMAs-done %.0f cyc, then wait+LDS %.0f cyc"
              % (spins.mean(), spins.max(), np.median(own), np.median(st[1:, 6] - buf[mode, 1:steps, 11].astype(np.int64))))
        entry, left = int(buf[mode, 63, 0]), int(buf[mode, 63, 1])
        print("   phase entry -> first step %.2f us, last stamped step top -> loop left %.2f us, whole phase %.1f us"
              % ((int(st[0, 0]) - entry) / ghz * 1e-3, (left - int(st[-1, 0])) / ghz * 1e-3, (left - entry) / ghz * 1e-3))
        pro = buf[mode, 63, :7].astype(np.int64)
        names = ["entry", "loop left", "weights issued (K in LDS)", "arrival ticket taken", "hello handshake done", "first barrier passed", "tile state + x tiles in LDS"]
        print("   prologue (us after phase entry): " + ", ".join("%s %.2f" % (names[i], (pro[i] - pro[0]) / ghz * 1e-3) for i in (2, 3, 4, 5, 6))
              + ", first step %.2f" % ((int(st[0, 0]) - pro[0]) / ghz * 1e-3))
        med = np.median(seg[1:], axis=0)
        for i, v in enumerate(med):
            print("   %-34s %8.0f cyc %7.0f ns  %5.1f%%" % (SEG[i], v, v / ghz, 100.0 * v / med.sum()))


if __name__ == "__main__":
    main()
