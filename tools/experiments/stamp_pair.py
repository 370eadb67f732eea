#!/usr/bin/env python3
"""Where does a step of the two-tiles-per-workgroup kernel (lstm_pair.hip) go?  Diagnostic build
(make -C longterm360fov_amd/csrc stamps), per-segment s_memtime deltas of wave 0 of each set of one workgroup."""
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

SEG = [["cell + publish + own h + flag A", "x(t+1).K MFMAs (48)", "gather request + wait flag A", "own-slice MFMAs (16)",
        "gather finish", "x -> LDS, flag B, wait", "partner MFMAs (112)"],
       ["cell + publish + own h + flag A", "wait flag A", "own-slice MFMAs (16)", "gather request + finish", "flag B + wait",
        "Dense partial (16 MFMAs) + flag D", "partner MFMAs (112)", "wait flag D", "y sum, tanh, y.K (4 MFMAs)"]]


def main():
    B, T_in, T_out, H = int(os.environ.get("B", 1024)), 30, 30, 256
    w = O.init_seq2seq(1234, H=H, bias_noise=0.05)
    enc, dec0, _ = O.synthetic_batch(1234, B, T_in, T_out)
    dw = {k: torch.from_numpy(v).cuda() for k, v in w.items()}
    d_enc, d_dec0 = torch.from_numpy(enc).cuda(), torch.from_numpy(dec0).cuda()
    ws = ops.Workspace()
    for _ in range(3):
        ops.seq2seq_decode(d_enc, d_dec0, dw, T_out, impl="cluster", workspace=ws)
    ws.check()
    L = _lib.lib()
    buf = np.zeros((2, 2, 64, 12), dtype=np.uint64)
    L.fov_debug_read_pair_stamps.argtypes = [ctypes.c_void_p]
    assert L.fov_debug_read_pair_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    for st_ in range(2):
        for ph, name, steps, nslot in ((0, "encoder", T_in, 8), (1, "decoder", T_out, 10)):
            st = buf[st_, ph, :steps, :nslot].astype(np.int64)
            seg = np.diff(st, axis=1)
            step_total = np.diff(st[:, 0])
            real = buf[st_, ph, :steps, 11].astype(np.int64)
            ghz = (st[-1, 0] - st[0, 0]) / ((real[-1] - real[0]) * 10.0)
            print("== set %d %s: %.0f cycles = %.2f us per step (median of %d), clock %.2f GHz"
                  % (st_, name, np.median(step_total), np.median(step_total) / ghz * 1e-3, steps - 1, ghz))
            med = np.median(seg[1:-1], axis=0)
            for i, v in enumerate(med):
                print("   %-40s %8.0f cyc %7.0f ns  %5.1f%%" % (SEG[ph][i], v, v / ghz, 100.0 * v / med.sum()))


if __name__ == "__main__":
    main()
