import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from longterm360fov_amd import ops
from oracle import fov_oracle as O
B, H = 32, 512
rng = np.random.default_rng(0)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
for F in (90, 512):
    K, R, b = (d(a) for a in O.init_lstm(rng, F, H))
    ws = ops.Workspace(); pts = []
    for T in (2, 6, 10, 20):
        x = d(rng.uniform(-1, 1, (B, T, F)))
        for _ in range(20): ops.lstm_seq(x, K, R, b, workspace=ws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300): ops.lstm_seq(x, K, R, b, workspace=ws)
        e1.record(); torch.cuda.synchronize()
        pts.append((T, e0.elapsed_time(e1) / 300 * 1e3))
    a, c = np.polyfit([p[0] for p in pts], [p[1] for p in pts], 1)
    print("single layer F=%d: %s -> %.2f us per step + %.1f us per launch" % (F, ["%.1f" % p[1] for p in pts], a, c))
