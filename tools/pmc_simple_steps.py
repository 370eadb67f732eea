"""Small fixed workloads for HBM-counter passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per pass), each printing
how many identical steps it ran so that tools/pmc_run_total.py can divide the run's total by it:
    python3 tools/pmc_simple_steps.py config1   - fused encoder + decoder at configs[0] (H = 128, batch 32, T 10 -> 10)
    python3 tools/pmc_simple_steps.py a10       - lstm.py's 2 x LSTMCell(400) forward, padded to 512 (batch 32, 10 steps)
    python3 tools/pmc_simple_steps.py convlstm  - ConvLSTM seq2seq whole-model predict at configs[3] (B = 256)
Every step of a run is the same call on the same inputs (no separate warm-up shape), so total / steps is the per-step figure."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longterm360fov_amd import ops  # noqa: E402
from oracle import fov_oracle as O  # noqa: E402

what = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else {"config1": 20, "a10": 20, "convlstm": 2}[what]
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
if what == "config1":
    w = O.init_seq2seq(1234, H=128, bias_noise=0.05)
    enc, dec0, _ = O.synthetic_batch(1234, 32, 10, 10)
    dw = {k: d(v) for k, v in w.items()}
    e, d0, ws = d(enc), d(dec0), ops.Workspace()
    for _ in range(steps):
        ops.seq2seq_decode(e, d0, dw, 10, impl="auto", workspace=ws)
    torch.cuda.synchronize(); ws.check()
elif what == "a10":
    from longterm360fov_amd.models import pad_lstm
    rng = np.random.default_rng(400)
    layers = [O.init_lstm(rng, 90, 400), O.init_lstm(rng, 400, 400)]
    dl = [tuple(d(a) for a in pad_lstm(K, R, b, 512, pad_input=(l > 0))) for l, (K, R, b) in enumerate(layers)]
    x, ws = d(np.random.default_rng(7).uniform(-1, 1, (32, 10, 90)).astype(np.float32)), ops.Workspace()
    for _ in range(steps):
        if ops.lstm_stack2_supported(32, 10, 90, 512):      # both layers as one launch (fov_lstm_stack2_fwd)
            ops.lstm_stack2(x, dl[0], dl[1], workspace=ws)
            continue
        inp = x
        for K, R, b in dl:
            inp, hT, cT = ops.lstm_seq(inp, K, R, b, act="sigmoid", workspace=ws)
    torch.cuda.synchronize(); ws.check()
else:
    from longterm360fov_amd.models import ConvLSTMSeq2Seq
    w = O.init_convlstm_seq2seq(1, C=30, latent_dim=16, head="conv2d")
    m = ConvLSTMSeq2Seq(w, head="conv2d")
    xe = np.random.default_rng(1).random((256, 10, 36, 18, 30), dtype=np.float32)
    for _ in range(steps):
        m.predict([xe, xe[:, -1:]], predict_step=10)
    torch.cuda.synchronize()
print("steps %d" % steps)
