"""Fixed cost of a persistent layer launch: time the layer at T = 1, 2, 4, 10, 30 - the intercept is dispatch + prologue + epilogue."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import fov_oracle as O
from longterm360fov_amd import ops
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
rng = np.random.default_rng(0)
for (B, F, H, name) in ((1024, 90, 256, "cluster enc"), (512, 90, 256, "wide-narrow (<=32 tiles)"), (512, 256, 256, "wide")):
    K, R, b = O.init_lstm(rng, F, H)
    K, R, b = dev(K), dev(R), dev(b)
    ws = ops.Workspace()
    for T in (1, 2, 4, 10, 30):
        x = torch.randn(B, T, F, device="cuda")
        us = timeit(lambda: ops.lstm_seq(x, K, R, b, act="sigmoid", workspace=ws))
        print("%-26s B=%d F=%d T=%2d : %7.1f us" % (name, B, F, T, us), flush=True)
    for T in (1, 2, 10):
        x = torch.randn(B, T, F, device="cuda")
        us = timeit(lambda: ops.lstm_seq_bf16(x, K, R, b, act="sigmoid", workspace=ws, reserve=False))
        print("%-26s bf16 B=%d F=%d T=%2d : %7.1f us" % (name, B, F, T, us), flush=True)
