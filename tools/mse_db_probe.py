import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from longterm360fov_amd import ops
rng = np.random.default_rng(0)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
for rows, O in ((30720, 6), (5120, 6), (320, 3)):
    y = d(np.tanh(rng.standard_normal((rows, O)))); t = d(rng.uniform(-1, 1, (rows, O)))
    db = torch.zeros(O, device="cuda"); sc = ops.Scratch(); dpre = torch.empty_like(y); loss = torch.zeros(1, device="cuda")
    for name, fn in (("with db", lambda: ops.mse_dense_grad(y, t, "tanh", scratch=sc, dpre=dpre, loss=loss, db=db)),
                     ("without", lambda: ops.mse_dense_grad(y, t, "tanh", scratch=sc, dpre=dpre, loss=loss))):
        for _ in range(20): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(500): fn()
        e1.record(); torch.cuda.synchronize()
        print("rows=%d O=%d %s: %.2f us per call" % (rows, O, name, e0.elapsed_time(e1) / 500 * 1e3))
