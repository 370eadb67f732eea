#!/usr/bin/env python3
"""Phase stamps of the three-stages-in-flight TN product (diagnostic build: make -C longterm360fov_amd/csrc stamps):
blocks 0, 100, 200 and the last one stamp entry / prologue loads issued / first stage in LDS / loop end / partials stored."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from longterm360fov_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "longterm360fov_amd", "lib", "libfov360_hip_stamps.so")
from longterm360fov_amd import ops
dev = torch.device("cuda:0")
TB, H = 5120, 256
h1, h2 = torch.randn(TB, H, device=dev), torch.randn(TB, H, device=dev)
dz = torch.randn(TB, 4 * H, device=dev)
out = torch.zeros((2 * H + 1) * 4 * H, device=dev)
sc = ops.Scratch()
for two in (True, False):
    o = out if two else out[:(H + 1) * 4 * H]
    for _ in range(5):
        ops.wgrad_fused(h1, h2 if two else None, dz, o, scratch=sc, dtype="bf16")
    torch.cuda.synchronize()
    L = _lib.lib()
    buf = np.zeros((4, 8), dtype=np.uint64)
    L.fov_debug_read_tn3_stamps.argtypes = [ctypes.c_void_p]
    assert L.fov_debug_read_tn3_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    s = buf.astype(np.int64)
    t0 = s[:, 0].min()
    print("[%s]^T dz:" % ("h1|h2|1" if two else "h1|1"))
    for i, name in enumerate(("block 0", "block 100", "block 200", "last block")):
        e = s[i]
        print("  %-10s entry +%6d | loads issued %5d | first stage in LDS %5d | loop (%d stages) %6d = %4d/stage | epilogue %5d | total %6d cycles"
              % (name, e[0] - t0, e[1] - e[0], e[2] - e[1], e[5], e[3] - e[2], (e[3] - e[2]) // max(e[5], 1), e[4] - e[3], e[4] - e[0]))
