"""Host-side cost of one training step at batch 32 (cProfile over 200 steps, GPU work asynchronous)."""
import cProfile, os, pstats, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longterm360fov_amd.models import Seq2SeqLSTM
from oracle import fov_oracle as O

H = int(sys.argv[1]) if len(sys.argv) > 1 else 64
enc, dec0, tgt = O.synthetic_batch(1234, 32, 10, 10)
dec_in = np.concatenate([dec0, tgt[:, :-1]], axis=1)
m = Seq2SeqLSTM(latent_dim=H, recurrent_activation="sigmoid", seed=1)
m.compile(optimizer="Adam", loss="mean_squared_error")
tr = m._get_trainer()
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
batch = [d(enc), d(dec_in), d(tgt)]
for _ in range(20):
    tr.train_step(*batch)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    tr.train_step(*batch)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
