"""Time the configs[2] training step (and the fused inference call) with a given build of the library:
   python tools/time_train_mixing.py [lib.so] [f32|bf16]     (A/B of kernel variants on ONE box; bench.py is the reported number)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longterm360fov_amd import _lib  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1].endswith(".so"):
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
from longterm360fov_amd import models, training  # noqa: E402
from oracle import fov_oracle as O  # noqa: E402

B, T_in, T_out, U = 512, 10, 10, 34
w = O.init_others_mixing(1234, H=256, num_user=U, bias_noise=0.05)
enc, dec0, tgt, oth = O.synthetic_batch(1234, B, T_in, T_out, num_others=U - 1)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
batch = [d(enc), d(oth), d(dec0), d(tgt)]
tr = training.OthersMixingTrainer(w, dtype=dtype)
for _ in range(5):
    tr.train_step(*batch)
tr.check()
res = []
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        loss = tr.train_step(*batch)
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / 40 * 1e3)
tr.check()
print("%-40s %s train step %.4f ms (runs %s) loss %.6f" % (sys.argv[1] if len(sys.argv) > 1 else "shipped", dtype, min(res),
                                                          " ".join("%.4f" % r for r in res), float(loss.item())), flush=True)
