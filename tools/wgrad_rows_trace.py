#!/usr/bin/env python3
"""Per-workgroup timeline of wgrad_rows_kernel (diagnostic build: make stamps) at lstm.py's shape: entry / MFMAs done / exit of
every workgroup with the CU it ran on -> how many workgroups a CU runs at once, how long a workgroup takes alone and shared."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from longterm360fov_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "longterm360fov_amd", "lib", "libfov360_hip_stamps.so")
from longterm360fov_amd import ops
dev = torch.device("cuda:0")
B, T, F, H = 32, 10, 90, int(os.environ.get("TRACE_H", "512"))
r = lambda *s: torch.randn(s, device=dev)
x1, hs1, dz1, hs2, dz2, h0 = r(B, T, F), r(B, T, H), r(B, T, 4 * H), r(B, T, H), r(B, T, 4 * H), r(B, H)
out = [torch.zeros(F, 4 * H, device=dev), torch.zeros(H, 4 * H, device=dev), torch.zeros(4 * H, device=dev),
       torch.zeros(H, 4 * H, device=dev), torch.zeros(H, 4 * H, device=dev), torch.zeros(4 * H, device=dev)]
sc = ops.Scratch()
for _ in range(5):
    ops.lstm_seq_wgrad_pair((x1, hs1, h0, dz1) + tuple(out[:3]), (hs1, hs2, h0, dz2) + tuple(out[3:]), scratch=sc)
torch.cuda.synchronize()
L = _lib.lib()
buf = np.zeros((4096, 6), dtype=np.uint64)
L.fov_debug_read_wr_trace.argtypes = [ctypes.c_void_p]
assert L.fov_debug_read_wr_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
used = buf[:, 2] > 0
t = buf[used].astype(np.int64)
n = len(t)
t0 = t[:, 0].min()
st, mid, en = t[:, 0] - t0, t[:, 1] - t0, t[:, 2] - t0
hw = t[:, 3] & 0xffffffff
xcc = t[:, 3] >> 32
cu = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)      # xcc | se | sh | cu
rt0, rt1 = t[:, 4], t[:, 5]
print("real time (100 MHz counter): kernel span %.1f us; median workgroup %.2f us = %d cycles -> %.2f GHz" % ((rt1.max() - rt0.min()) / 100.0,
      np.median(rt1 - rt0) / 100.0, np.median(t[:, 2] - t[:, 0]), np.median((t[:, 2] - t[:, 0]) / np.maximum(rt1 - rt0, 1)) / 10.0))
rs, re_ = (rt0 - rt0.min()) / 100.0, (rt1 - rt0.min()) / 100.0
print("workgroups running at t = 2, 5, 10, 15, 20, 25, 30 us:", [int(((rs <= x) & (re_ > x)).sum()) for x in (2, 5, 10, 15, 20, 25, 30)])
print("%d workgroups, span %d cycles; workgroup duration: min %d median %d max %d; to MFMAs done: median %d"
      % (n, en.max(), (en - st).min(), np.median(en - st), (en - st).max(), np.median(mid - st)))
cus = np.unique(cu)
conc = []
for c in cus:
    m = cu == c
    ev = sorted([(s, 1) for s in st[m]] + [(e, -1) for e in en[m]])
    cur, last, busy = 0, 0, {}
    for tt, d in ev:
        busy[cur] = busy.get(cur, 0) + tt - last
        cur += d; last = tt
    conc.append((m.sum(), busy))
tot = {}
for _, b in conc:
    for k, v in b.items(): tot[k] = tot.get(k, 0) + v
print("%d CUs used; workgroups per CU: min %d max %d" % (len(cus), min(c[0] for c in conc), max(c[0] for c in conc)))
print("CU time by number of resident workgroups (mean cycles per CU): " + ", ".join("%d: %d" % (k, v // len(cus)) for k, v in sorted(tot.items())))
order = np.argsort(st)
print("entry times: first 8", st[order[:8]].tolist(), " ... start of last workgroup", st.max())
for c in cus[:2]:
    m = np.where(cu == c)[0]
    print("CU %x:" % c, [(int(st[i]), int(mid[i]), int(en[i])) for i in m[np.argsort(st[m])]])
