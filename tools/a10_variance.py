"""Repeat-timing of the two-layer stack at (32, 10, 90) for H = 256 (persistent kernels): is the time stable?"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longterm360fov_amd import ops
from oracle import fov_oracle as O

H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B, T, F = 32, 10, 90
rng = np.random.default_rng(7)
x = torch.from_numpy(rng.uniform(-1, 1, (B, T, F)).astype(np.float32)).cuda()
lrng = np.random.default_rng(H)
layers = [O.init_lstm(lrng, F, H), O.init_lstm(lrng, H, H)]
dl = [tuple(torch.from_numpy(a).cuda() for a in l) for l in layers]
ws = ops.Workspace()

def step():
    inp = x
    for K, R, b in dl:
        inp, hT, cT = ops.lstm_seq(inp, K, R, b, act="sigmoid", workspace=ws)
    return inp

for _ in range(5):
    step()
torch.cuda.synchronize()
for rep in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        step()
    e1.record(); torch.cuda.synchronize()
    print("H=%d rep %d: %.4f ms per call, exchange mode %d" % (H, rep, e0.elapsed_time(e1) / 100, ws.exchange_mode()))
ws.check()
