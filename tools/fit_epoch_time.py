"""model.fit at the reference's batch size (FoV_seq2seq.py:112-117: batch 32): wall time per step inside fit() - host path
(batch gather, upload) included - beside the bare train_step on resident tensors."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longterm360fov_amd.models import Seq2SeqLSTM
from oracle import fov_oracle as O

H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N, B = 3200, 32
enc, dec0, tgt = O.synthetic_batch(5, N, 10, 10)
dec_in = np.concatenate([dec0, tgt[:, :-1]], axis=1)
m = Seq2SeqLSTM(latent_dim=H, recurrent_activation="sigmoid", seed=1)
m.compile(optimizer="Adam", loss="mean_squared_error")
m.fit([enc, dec_in], tgt, batch_size=B, epochs=1, shuffle=True)          # warm-up epoch
torch.cuda.synchronize()
t0 = time.perf_counter()
h = m.fit([enc, dec_in], tgt, batch_size=B, epochs=3, shuffle=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("H=%d: fit() %.1f ms per epoch of %d steps = %.4f ms per step (loss %.5f)" % (H, dt / 3 * 1e3, N // B, dt / 3 / (N // B) * 1e3, h.history["loss"][-1]))
