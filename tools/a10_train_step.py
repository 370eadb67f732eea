"""lstm.py's training step at its native shape (2 x LSTMCell(400), batch 32, 10 steps, 90 features) on TFLSTMTrainer, padded to
width 512 (default) or unpadded (--nopad): a few steps for a rocprofv3 --kernel-trace timeline (tools/step_timeline.py <csv> rmsprop)
and the HIP-event time of a step.   usage: python3 tools/a10_train_step.py [--nopad] [--steps N] [--head meanvar|gmm|raw]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longterm360fov_amd.training import TFLSTMTrainer  # noqa: E402

pad = "--nopad" not in sys.argv
steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 100
B, T, F, H = 32, 10, 90, 400
rng = np.random.default_rng(11)
cells = [((rng.standard_normal((Fin + H, 4 * H)) / np.sqrt(Fin + H)).astype(np.float32), np.zeros(4 * H, np.float32)) for Fin in (F, H)]
head = {}
for br in ("mu", "var"):
    head[br + "_W1"] = (rng.standard_normal((H, 32)) / 20).astype(np.float32)
    head[br + "_b1"] = np.zeros(32, np.float32)
    head[br + "_W2"] = (rng.standard_normal((32, 3)) / np.sqrt(32)).astype(np.float32)
    head[br + "_b2"] = np.zeros(3, np.float32)
x = torch.from_numpy(rng.uniform(-1, 1, (B, T, F)).astype(np.float32)).cuda()
y = torch.from_numpy(rng.uniform(-1, 1, (B, 1, 90)).astype(np.float32)).cuda()
init = torch.zeros((2, 2, B, H), device="cuda")
kind = sys.argv[sys.argv.index("--head") + 1] if "--head" in sys.argv else "meanvar"      # meanvar | gmm | raw (lstm.py:424-509)
if kind == "gmm":
    dims = [H, 64, 128, 256, 200]
    head = {}
    for l in range(4):
        head["fc%d_W" % (l + 1)] = (rng.uniform(-1, 1, (dims[l], dims[l + 1])) * np.sqrt(6.0 / (dims[l] + dims[l + 1]))).astype(np.float32)
        head["fc%d_b" % (l + 1)] = np.zeros(dims[l + 1], np.float32)
    y = torch.from_numpy(rng.uniform(-1, 1, (B, 10, 90)).astype(np.float32)).cuda()
elif kind == "raw":
    dims = [H, 128, 256, 90]
    head = {}
    for l in range(3):
        head["conv%d_W" % (l + 1)] = (rng.uniform(-1, 1, (5, dims[l], dims[l + 1])) * np.sqrt(6.0 / (5 * (dims[l] + dims[l + 1])))).astype(np.float32)
        head["conv%d_b" % (l + 1)] = np.zeros(dims[l + 1], np.float32)
tr = TFLSTMTrainer(cells, head, lr=1e-5, fps=30, running_length=10, pad=pad, head_kind=kind)
state = init      # lstm.py:612-620: the state a step returns is the next step's fed state (state_view: no copies around it)
for _ in range(5):
    _, state = tr.train_step(x, y, state, state_view=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps):
    _, state = tr.train_step(x, y, state, state_view=True)
e1.record()
torch.cuda.synchronize()
tr.ws.check()
print("pad=%s head=%s  %.4f ms per training step" % (pad, kind, e0.elapsed_time(e1) / steps))
