import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from longterm360fov_amd import ops
from oracle import fov_oracle as O
rng = np.random.default_rng(0)
B, T, F, H = 512, 10, 256, 256
K, R, b = O.init_lstm(rng, F, H, np.float32)
x = rng.uniform(-1, 1, (B, T, F)).astype(np.float32)
d = lambda a: torch.from_numpy(a).cuda()
dx, dK, dR, db = d(x), d(K), d(R), d(b)
for forced in ("0", "1", "0"):
    os.environ["FOV_FORCE_SAFE_EXCHANGE"] = forced
    ws = ops.Workspace()
    for _ in range(3):
        ops.lstm_seq(dx, dK, dR, db, workspace=ws)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        ops.lstm_seq(dx, dK, dR, db, workspace=ws)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 50 * 1e3
    ws.check()
    print("wide layer (B=512, T=10, F=256): forced_safe=%s exchange mode %d, %.4f ms per call" % (forced, ws.exchange_mode(), ms), flush=True)
