#!/usr/bin/env python3
"""Per-segment cycle shares of one step of the bf16 LSTM layer kernel (diagnostic build: make -C longterm360fov_amd/csrc
stamps).  One wave (block 5, wave 0) stamps s_memtime at the phase boundaries."""
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

SEG = ["x staging (LDS write of x_{t+1}, loads of x_{t+2})", "cell update", "publish", "barrier 1",
       "own h -> LDS + x.K MFMAs", "gather issue + tape stores (issue)", "gather wait + LDS", "barrier 2", "h.R MFMAs"]


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    B, T, H = 512, 10, 256
    rng = np.random.default_rng(0)
    K, R, b = O.init_lstm(rng, F, H, np.float32)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    x = d(rng.uniform(-1, 1, (B, T, F)))
    dK, dR, db = d(K), d(R), d(b)
    ws = ops.Workspace()
    for _ in range(3):
        ops.lstm_seq_bf16(x, dK, dR, db, workspace=ws)
    ws.check()
    L = _lib.lib()
    buf = np.zeros((32, 12), dtype=np.uint64)
    L.fov_debug_read_q_stamps.argtypes = [ctypes.c_void_p]
    assert L.fov_debug_read_q_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    s = buf[:T, :10].astype(np.int64)
    seg = np.diff(s, axis=1)
    step = np.diff(s[:, 0])
    e = buf[31].astype(np.int64)
    print("F=%d: entry -> weights resident %d cycles; -> first step %d; last step end -> tiles done %d; leave %d; whole %d cycles"
          % (F, e[1] - e[0], s[0, 0] - e[1], e[2] - s[T - 1, 9], e[3] - e[2], e[3] - e[0]))
    print("step: median %.0f cycles (%.2f us at 2.1 GHz)" % (np.median(step), np.median(step) / 2100.0))
    med = np.median(seg[1:T - 1], axis=0)
    for i, v in enumerate(med):
        print("   %-52s %8.0f cyc  %5.1f%%" % (SEG[i], v, 100.0 * v / med.sum()))
    print("per-step segment table (cycles):")
    for t in range(T):
        print("   t=%d " % t + " ".join("%6d" % v for v in seg[t]))


def main_bwd():
    """python tools/stamp_bf16_layer.py --bwd : the N-split bf16 BPTT kernel (lstm_bwd8n_bf16_kernel)."""
    SEGB = ["top of step", "gates backward (incl. waits for the tape and dh)", "publish dz + own LDS", "tape loads + dz stores (issue)",
            "gather (issue, wait, LDS)", "barrier A", "MFMAs + partials -> LDS", "barrier B", "sum of the four partials + rotate"]
    B, T, F, H = 512, 10, 256, 256
    rng = np.random.default_rng(0)
    K, R, b = O.init_lstm(rng, F, H, np.float32)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    x = d(rng.uniform(-1, 1, (B, T, F)))
    dK, dR, db = d(K), d(R), d(b)
    hs, hT, cT, res = ops.lstm_seq_bf16(x, dK, dR, db)
    dhs = d(0.1 * rng.standard_normal((B, T, H)))
    sc = ops.Scratch()
    for _ in range(3):
        ops.lstm_seq_bwd(x, dK, dR, hs, res, dhs=dhs, need_state_grads=True, scratch=sc, dtype="bf16")
    sc.check()
    L = _lib.lib()
    buf = np.zeros((32, 12), dtype=np.uint64)
    L.fov_debug_read_b8_stamps.argtypes = [ctypes.c_void_p]
    assert L.fov_debug_read_b8_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    s = buf[:T, :10].astype(np.int64)
    seg = np.diff(s, axis=1)
    step = np.diff(s[:, 0])
    e = buf[31].astype(np.int64)
    print("BPTT bf16: entry -> first step %d cycles; whole %d cycles; step: median %.0f cycles (%.2f us at 2.1 GHz)"
          % (s[0, 0] - e[0], e[1] - e[0], np.median(step), np.median(step) / 2100.0))
    med = np.median(seg[1:T - 1], axis=0)
    for i, v in enumerate(med):
        print("   %-52s %8.0f cyc  %5.1f%%" % (SEGB[i], v, 100.0 * v / med.sum()))
    for t in range(T):
        print("   step %d " % t + " ".join("%6d" % v for v in seg[t]))


def main_stack2():
    """python tools/stamp_bf16_layer.py --stack2 : the two-layer bf16 forward kernel (lstm_stack2_bf16_kernel), both roles."""
    SEGS = ["cell update", "publish", "barrier 1", "own h -> LDS, x staging, gather issue", "tape stores (issue)",
            "gather of h wait + LDS", "gather of the lower layer's h (upper role)", "barrier 2", "x.K + h.R MFMAs"]
    B, T, F, H = 512, 10, 3, 256
    rng = np.random.default_rng(0)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    K1, R1, b1 = O.init_lstm(rng, F, H, np.float32)
    K2, R2, b2 = O.init_lstm(rng, H, H, np.float32)
    x = d(rng.uniform(-1, 1, (B, T, F)))
    W = [d(a) for a in (K1, R1, b1, K2, R2, b2)]
    ws = ops.Workspace()
    for _ in range(3):
        ops.lstm_stack2_bf16(x, W[:3], W[3:], workspace=ws)
    ws.check()
    L = _lib.lib()
    buf = np.zeros((2, 32, 12), dtype=np.uint64)
    L.fov_debug_read_s2_stamps.argtypes = [ctypes.c_void_p]
    assert L.fov_debug_read_s2_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    for role, name in enumerate(("lower layer (producer)", "upper layer (consumer)")):
        s = buf[role, :T, :10].astype(np.int64)
        seg = np.diff(s, axis=1)
        step = np.diff(s[:, 0])
        e = buf[role, 31].astype(np.int64)
        print("%s: entry -> first step %d cycles; recurrence %d; leave %d; whole %d cycles; step: median %.0f cycles (%.2f us at 2.1 GHz)"
              % (name, e[1] - e[0], e[2] - e[1], e[3] - e[2], e[3] - e[0], np.median(step), np.median(step) / 2100.0))
        pr = buf[role, 30, :6].astype(np.int64) - e[0]
        print("   prologue (cycles after entry): arrival counted %d, weights requested and packed %d, LDS zeroed %d, hello words seen %d, "
              "first barrier %d, x_0 in LDS %d, x_0.K done %d" % (*pr, e[1] - e[0]))
        med = np.median(seg[1:T - 1], axis=0)
        for i, v in enumerate(med):
            print("   %-52s %8.0f cyc  %5.1f%%" % (SEGS[i], v, 100.0 * v / med.sum()))
        for t in range(T):
            print("   step %d " % t + " ".join("%6d" % v for v in seg[t]))
    e0, e1 = buf[0, 31].astype(np.int64), buf[1, 31].astype(np.int64)
    print("upper role entered %+d cycles after the lower one; left %+d cycles after it" % (e1[0] - e0[0], e1[3] - e0[3]))


if __name__ == "__main__":
    main_stack2() if "--stack2" in sys.argv else main_bwd() if "--bwd" in sys.argv else main()
