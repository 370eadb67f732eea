"""Time the bf16 BPTT kernel (data path only) with the shipped library or a timing variant (tools/b8_variants.sh):
   python tools/b8_variant_time.py [build/dbg/libfov_B8_NOTAPE.so ...]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, '.')
from longterm360fov_amd import _lib
libs = sys.argv[1:] or [None]
if libs[0]:
    _lib.LIB_PATH = os.path.abspath(libs[0])
from longterm360fov_amd import ops
from oracle import fov_oracle as O
B, T, F, H = 512, 40, 256, 256
rng = np.random.default_rng(0)
K, R, b = O.init_lstm(rng, F, H, np.float32)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
x = d(rng.uniform(-1, 1, (B, T, F)))
dK, dR, db = d(K), d(R), d(b)
hs, hT, cT, res = ops.lstm_seq_bf16(x, dK, dR, db)
dhs = d(0.1 * rng.standard_normal((B, T, H)))
sc = ops.Scratch()
dz = torch.empty((B, T, 4 * H), device="cuda")
def run(dx):
    ops.lstm_seq_bwd(x, dK, dR, hs, res, dhs=dhs, need_dx=dx, need_state_grads=True, scratch=sc, dtype="bf16", dz=dz, need_weight_grads=False)
for dx in (False, True):
    for _ in range(5): run(dx)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(30): run(dx)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 30 * 1e3
    print("%-36s dx=%d  %.1f us per call, %.2f us per step (T = %d)" % (os.path.basename(libs[0]) if libs[0] else "shipped", dx, us, us / T, T), flush=True)
