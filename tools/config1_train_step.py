"""One width of bench.py --mode config1 --train on its own (for a rocprofv3 timeline): python tools/config1_train_step.py H [impl] [steps]"""
import sys
import numpy as np
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longterm360fov_amd.models import Seq2SeqLSTM
from oracle import fov_oracle as O

H = int(sys.argv[1]); impl = sys.argv[2] if len(sys.argv) > 2 else "auto"; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
B, T_in, T_out = 32, 10, 10
enc, dec0, tgt = O.synthetic_batch(1234, B, T_in, T_out)
dec_in = np.concatenate([dec0, tgt[:, :-1]], axis=1)
m = Seq2SeqLSTM(latent_dim=H, recurrent_activation="sigmoid", impl=impl, seed=1)
m.compile(optimizer="Adam", loss="mean_squared_error")
tr = m._get_trainer()
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
batch = [d(enc), d(dec_in), d(tgt)]
for _ in range(5):
    tr.train_step(*batch)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(steps):
    tr.train_step(*batch)
ev1.record(); torch.cuda.synchronize()
tr.check()
print("H=%d impl=%s: %.4f ms per step" % (H, impl, ev0.elapsed_time(ev1) / steps))
