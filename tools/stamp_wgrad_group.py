#!/usr/bin/env python3
"""Phase stamps of wgrad_group_kernel (diagnostic build: make -C longterm360fov_amd/csrc stamps) at lstm.py's shape
(two layers, 32 x 10 rows, H = 512, F = 90): entry / loads issued / first stage in LDS / loop end / stored."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from longterm360fov_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "longterm360fov_amd", "lib", "libfov360_hip_stamps.so")
from longterm360fov_amd import ops
dev = torch.device("cuda:0")
B, T, F, H = 32, 10, 90, 512
x = torch.randn(B, T, F, device=dev); hs = torch.randn(B, T, H, device=dev); dz = torch.randn(B, T, 4 * H, device=dev)
h0 = torch.randn(B, H, device=dev)
dK, dR, db = torch.zeros(F, 4 * H, device=dev), torch.zeros(H, 4 * H, device=dev), torch.zeros(4 * H, device=dev)
sc = ops.Scratch()
for _ in range(5):
    ops.lstm_seq_wgrad(x, hs, dz, dK=dK, dR=dR, db=db, h0=h0, scratch=sc)
torch.cuda.synchronize()
ref = torch.einsum("bth,btn->hn", torch.cat([h0[:, None], hs[:, :-1]], 1).double(), dz.double())
print("dR max err %.3e of %.3e; dK err %.3e; db err %.3e" % ((dR.double() - ref).abs().max().item(), ref.abs().max().item(),
      (dK.double() - torch.einsum("btf,btn->fn", x.double(), dz.double())).abs().max().item(), (db.double() - dz.double().sum((0, 1))).abs().max().item()))
L = _lib.lib()
buf = np.zeros((4, 8), dtype=np.uint64)
L.fov_debug_read_wg_stamps.argtypes = [ctypes.c_void_p]
assert L.fov_debug_read_wg_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
s = buf.astype(np.int64)
for i, name in enumerate(("block 0", "block 70", "block 150", "last block")):
    e = s[i]
    print("  %-10s loads issued %5d | first stage in LDS %5d | loop (%d stages) %6d = %4d/stage | epilogue %5d | total %6d cycles"
          % (name, e[1] - e[0], e[2] - e[1], e[5], e[3] - e[2], (e[3] - e[2]) // max(e[5], 1), e[4] - e[3], e[4] - e[0]))
