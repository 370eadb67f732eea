"""Does the per-step time of the mixing-model inference drift over a long run?  GPU event + host clock every 50 steps."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from longterm360fov_amd.models import OthersMixingSeq2Seq, _MIX_ORDER
from oracle import fov_oracle as O

H, T_in, T_out, U, B = 256, 10, 10, 34, 512
w = O.init_others_mixing(1234, H=H, num_user=U, bias_noise=0.05)
enc, dec0, tgt, oth = O.synthetic_batch(1234, B, T_in, T_out, num_others=U - 1)
m = OthersMixingSeq2Seq(latent_dim=H, num_user=U, recurrent_activation="sigmoid")
m.set_weights([w[k] for k in _MIX_ORDER])
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
a_enc, a_oth, a_dec = d(enc), d(oth), d(dec0)
for _ in range(10):
    m.predict_device(a_enc, a_oth, a_dec)
torch.cuda.synchronize()
mode = sys.argv[1] if len(sys.argv) > 1 else "free"
N, CH = 600, 50
evs = [torch.cuda.Event(enable_timing=True) for _ in range(N // CH + 1)]
host = []
evs[0].record()
t0 = time.perf_counter()
for i in range(N):
    m.predict_device(a_enc, a_oth, a_dec)
    if (i + 1) % CH == 0:
        evs[(i + 1) // CH].record()
        host.append(time.perf_counter() - t0)
        if mode == "sync":
            torch.cuda.synchronize()
torch.cuda.synchronize()
total = time.perf_counter() - t0
prev = 0.0
for k in range(N // CH):
    print("steps %3d-%3d: gpu %.4f ms/step   host enqueue %.4f ms/step" % (k * CH, (k + 1) * CH, evs[k].elapsed_time(evs[k + 1]) / CH, (host[k] - prev) / CH * 1e3))
    prev = host[k]
print("total %.4f ms/step (%s)" % (total / N * 1e3, mode))
