#!/usr/bin/env python3
"""Per-segment cycle shares of one step of the fused others-mixing decoder (diagnostic build:
make -C longterm360fov_amd/csrc stamps).  One wave (block 5, wave 0) stamps s_memtime at the phase boundaries."""
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

SEG = ["x.K1 MFMAs + cell 1", "publish h1 + gather issue + h2.R2 MFMAs (128)", "gather-1 wait + LDS", "barrier D",
       "h1.K2 MFMAs (128)", "cell 2 + publish h2", "gather issue + h1.R1 MFMAs (128)", "gather-2 wait + LDS", "barrier G",
       "head (Dense + mixing)", "barrier H"]


def main():
    B, T_out, H, U, NO = 512, 10, 256, 34, 6
    w = O.init_others_mixing(1234, H=H, num_user=U, bias_noise=0.05)
    dw = {k: torch.from_numpy(v).cuda() for k, v in w.items()}
    rng = np.random.default_rng(0)
    st = [torch.from_numpy((0.3 * rng.standard_normal((B, H))).astype(np.float32)).cuda() for _ in range(4)]
    dec0 = torch.from_numpy(rng.uniform(-1, 1, (B, 1, NO)).astype(np.float32)).cuda()
    oth_proj = torch.from_numpy(rng.uniform(-1, 1, (B, T_out, NO)).astype(np.float32)).cuda()
    Wp = dw["mix_W"][-NO:].contiguous()
    ws = ops.Workspace()
    for _ in range(3):
        ops.mix_decoder(dec0, st[0], st[1], st[2], st[3], oth_proj, dw, Wp, T_out, workspace=ws)
    ws.check()
    L = _lib.lib()
    buf = np.zeros((32, 12), dtype=np.uint64)
    L.fov_debug_read_mix_stamps.argtypes = [ctypes.c_void_p]
    assert L.fov_debug_read_mix_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    s = buf[:T_out].astype(np.int64)
    seg = np.diff(s, axis=1)
    step = np.diff(s[:, 0])
    print("step: median %.0f cycles (%.2f us at 2.17 GHz)" % (np.median(step), np.median(step) / 2170.0))
    med = np.median(seg[1:], axis=0)
    for i, v in enumerate(med):
        print("   %-48s %8.0f cyc  %5.1f%%" % (SEG[i], v, 100.0 * v / med.sum()))


if __name__ == "__main__" and "--bwd" not in sys.argv:
    main()


def main_bwd():
    """Same for the backward kernel (python tools/stamp_mix_decoder.py --bwd)."""
    SEGB = ["head backward", "barrier", "layer-2 gates backward (reserve loads) + dz2 -> LDS", "barrier + MFMAs (256) + publish",
            "barrier + gather of dh2 / dh1 pieces", "layer-1 gates backward + dz1 -> LDS", "barrier + MFMAs (128+32) + publish",
            "gather of dh1 / dx pieces + barrier"]
    from longterm360fov_amd.training import OthersMixingTrainer
    B, T_in, T_out, H, U = 512, 10, 10, 256, 34
    w = O.init_others_mixing(1234, H=H, num_user=U, bias_noise=0.05)
    enc, dec0, tgt, oth = O.synthetic_batch(1234, B, T_in, T_out, num_others=U - 1)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    tr = OthersMixingTrainer(w)
    for _ in range(3):
        tr.forward_backward(d(enc), d(oth), d(dec0), d(tgt))
    tr.ws.check(); tr.ws_bwd.check()
    L = _lib.lib()
    buf = np.zeros((32, 12), dtype=np.uint64)
    L.fov_debug_read_mixb_stamps.argtypes = [ctypes.c_void_p]
    assert L.fov_debug_read_mixb_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    s = buf[:T_out, :9].astype(np.int64)
    seg = np.diff(s, axis=1)
    step = np.diff(s[:, 0])
    print("backward step: median %.0f cycles (%.2f us at 2.17 GHz)" % (np.median(step), np.median(step) / 2170.0))
    med = np.median(seg[1:], axis=0)
    for i, v in enumerate(med):
        print("   %-60s %8.0f cyc  %5.1f%%" % (SEGB[i], v, 100.0 * v / med.sum()))


if __name__ == "__main__" and "--bwd" in sys.argv:
    main_bwd()
