"""Where is the one-off stall?  Per-call host time (synchronised) for the mixing inference and for a trivial launch loop."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from longterm360fov_amd import ops
from longterm360fov_amd.models import OthersMixingSeq2Seq, _MIX_ORDER
from oracle import fov_oracle as O

def top(ts, tag):
    ts = np.array(ts)
    idx = np.argsort(ts)[::-1][:4]
    print(tag, "median %.3f ms; slowest calls:" % (np.median(ts) * 1e3), [(int(i), round(float(ts[i]) * 1e3, 2)) for i in idx])

which = sys.argv[1]
if which == "dense":
    x = torch.rand(4096, 256, device="cuda"); W = torch.rand(256, 6, device="cuda"); b = torch.rand(6, device="cuda")
    ts = []
    for i in range(4000):
        t0 = time.perf_counter(); ops.dense(x, W, b); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    top(ts, "dense x4000")
elif which == "torch":
    x = torch.rand(4096, 256, device="cuda")
    ts = []
    for i in range(4000):
        t0 = time.perf_counter(); y = x * 2.0; torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    top(ts, "torch mul x4000")
else:
    H, T_in, T_out, U, B = 256, 10, 10, 34, 512
    w = O.init_others_mixing(1234, H=H, num_user=U, bias_noise=0.05)
    enc, dec0, tgt, oth = O.synthetic_batch(1234, B, T_in, T_out, num_others=U - 1)
    m = OthersMixingSeq2Seq(latent_dim=H, num_user=U, recurrent_activation="sigmoid")
    m.set_weights([w[k] for k in _MIX_ORDER])
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    a_enc, a_oth, a_dec = d(enc), d(oth), d(dec0)
    if len(sys.argv) > 2 and sys.argv[2] == "freeze":
        import gc
        gc.collect(); gc.freeze()
    ts = []
    for i in range(400):
        t0 = time.perf_counter(); m.predict_device(a_enc, a_oth, a_dec); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    top(ts, "mixing inference x400")
