"""Time the headline call with a diagnostic build of the cluster kernel (make -C longterm360fov_amd/csrc variants):
   python tools/variant_time.py [build/dbg/libfov_NOGATHER.so | ..._NODENSE.so | ..._NOCELL.so]   (no argument: the shipped library)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, '.')
from longterm360fov_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from longterm360fov_amd import ops
from oracle import fov_oracle as O
B, T_in, T_out, H = 1024, 30, 30, 256
w = O.init_seq2seq(1234, H=H, bias_noise=0.05)
enc, dec0, _ = O.synthetic_batch(1234, B, T_in, T_out)
dw = {k: torch.from_numpy(v).cuda() for k, v in w.items()}
d_enc, d_dec0 = torch.from_numpy(enc).cuda(), torch.from_numpy(dec0).cuda()
ws = ops.Workspace()
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
full = timeit(lambda: ops.seq2seq_decode(d_enc, d_dec0, dw, T_out, impl="cluster", workspace=ws))
encu = timeit(lambda: ops.lstm_seq(d_enc, dw["enc_K"], dw["enc_R"], dw["enc_b"], act="sigmoid", impl="cluster", return_sequences=False, workspace=ws))
print("%-40s full %.1f us  encoder %.1f us  decoder ~%.1f us" % (sys.argv[1] if len(sys.argv) > 1 else "shipped", full, encu, full - encu), flush=True)
