"""configs[3] training step (ConvLSTM seq2seq, 36 x 18 x 30 maps, B = 256, T 10 -> 10, RMSprop + MSE): N steps for a rocprofv3
--kernel-trace --stats run.   usage: python3 tools/convlstm_train_step.py [steps] [batch]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longterm360fov_amd.training import ConvLSTMTrainer  # noqa: E402
from oracle import fov_oracle as O  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
w = O.init_convlstm_seq2seq(1, C=30, latent_dim=16, head="conv2d")
tr = ConvLSTMTrainer(w, head="conv2d")
x = torch.rand((B, 10, 36, 18, 30), device="cuda")
tgt = torch.rand((B, 10, 36, 18, 30), device="cuda")
tgt = tgt / tgt.sum(-1, keepdim=True)
dec0 = x[:, -1:].contiguous()
tr.train_step(x, dec0, tgt)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps):
    tr.train_step(x, dec0, tgt)
e1.record()
torch.cuda.synchronize()
print("%.1f ms per training step at batch %d" % (e0.elapsed_time(e1) / steps, B))
