#!/usr/bin/env python3
"""Per-segment s_memtime deltas of one wave of the tile-pair kernel (lstm_pair.hip, diagnostic build: make stamps), encoder blocks."""
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

SEG = ["x staging", "gather finish (wait + LDS)", "barrier", "h.R 128 MFMAs (+ gather request)", "cell update", "publish + own columns", "bias + x.K 48 MFMAs"]
B, T_in, T_out, H = 1024, 30, 2, 256
w = O.init_seq2seq(1234, H=H, bias_noise=0.05)
enc, dec0, _ = O.synthetic_batch(1234, B, T_in, T_out)
dw = {k: torch.from_numpy(v).cuda() for k, v in w.items()}
ws = ops.Workspace()
for _ in range(3):
    ops.seq2seq_decode(torch.from_numpy(enc).cuda(), torch.from_numpy(dec0).cuda(), dw, T_out, workspace=ws)
ws.check()
L = _lib.lib()
buf = np.zeros((128, 2, 10), dtype=np.uint64)
L.fov_debug_read_pair_stamps.argtypes = [ctypes.c_void_p]
assert L.fov_debug_read_pair_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
st = buf[:T_in, :, :8].astype(np.int64)
for i in range(2):
    seg = np.diff(st[2:, i, :], axis=1)
    print("tile %d: block %.0f cycles (median), segments:" % (i, np.median(seg.sum(1))))
    for k, v in enumerate(np.median(seg, axis=0)):
        print("   %-40s %7.0f" % (SEG[k], v))
pairstep = np.diff(st[2:, 0, 0])
gap01 = st[2:, 1, 0] - st[2:, 0, 7]
gap10 = st[3:, 0, 0] - st[2:-1, 1, 7]
print("pair-step %.0f cycles; end of block A -> start of block B %.0f, end of B -> start of next A %.0f" % (np.median(pairstep), np.median(gap01), np.median(gap10)))
