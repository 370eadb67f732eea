"""CPU baseline legs on torch's own CPU kernels (oneDNN / MKL-backed torch.nn.LSTM, LSTMCell, Linear).

TEST INFRASTRUCTURE ONLY (like everything under oracle/): imported by bench.py's `cpu_baseline` leg and by tests,
never by the product.  SURVEY.md 8(d) / BASELINE.md section 4 ask for the reference's CPU path to be timed as the
restated graph on torch CPU ops next to the C port (oracle/lstm_ref.c): Keras/TensorFlow themselves cannot be
installed on either box.  Weight mapping: W_ih = K^T, W_hh = R^T, b_ih = b, b_hh = 0;
torch's gate order i,f,g,o is Keras's i,f,c,o; recurrent activation = sigmoid only (torch has no hard_sigmoid LSTM).
"""
import os
import time

import numpy as np
import torch


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


class Seq2SeqCPU:
    """FoV_seq2seq.py:82-97,137-178 (1-layer encoder, 1-layer decoder fed its own Dense(tanh) output) on torch CPU."""

    def __init__(self, w, threads=None):
        self.threads = int(threads or usable_cores())
        torch.set_num_threads(self.threads)
        F_enc, H4 = w["enc_K"].shape
        H = H4 // 4
        F_dec = w["dec_K"].shape[0]
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        self.enc = torch.nn.LSTM(F_enc, H, batch_first=True)
        self.cell = torch.nn.LSTMCell(F_dec, H)
        self.lin = torch.nn.Linear(H, F_dec)
        with torch.no_grad():
            self.enc.weight_ih_l0.copy_(t(w["enc_K"].T))
            self.enc.weight_hh_l0.copy_(t(w["enc_R"].T))
            self.enc.bias_ih_l0.copy_(t(w["enc_b"]))
            self.enc.bias_hh_l0.zero_()
            self.cell.weight_ih.copy_(t(w["dec_K"].T))
            self.cell.weight_hh.copy_(t(w["dec_R"].T))
            self.cell.bias_ih.copy_(t(w["dec_b"]))
            self.cell.bias_hh.zero_()
            self.lin.weight.copy_(t(w["dense_W"].T))
            self.lin.bias.copy_(t(w["dense_b"]))

    @torch.no_grad()
    def decode(self, enc_in, dec_in0, T_out):
        x = torch.from_numpy(np.ascontiguousarray(enc_in, dtype=np.float32))
        y = torch.from_numpy(np.ascontiguousarray(dec_in0[:, 0], dtype=np.float32))
        if x.shape[1] > 0:
            _, (h, c) = self.enc(x)
            h, c = h[0], c[0]
        else:
            h = torch.zeros((x.shape[0], self.cell.hidden_size))
            c = torch.zeros_like(h)
        outs = []
        for _ in range(T_out):
            h, c = self.cell(y, (h, c))
            y = torch.tanh(self.lin(h))
            outs.append(y)
        return torch.stack(outs, 1).numpy()


def timed_median(fn, budget_s=10.0, min_iters=10, max_iters=200, warmup=3):
    """Median seconds per call of fn(): `warmup` untimed calls, then at least `min_iters` timed ones, more while the
    time budget allows."""
    for _ in range(warmup):
        fn()
    times = []
    t_start = time.perf_counter()
    while len(times) < min_iters or (time.perf_counter() - t_start < budget_s and len(times) < max_iters):
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
    return float(np.median(times)), len(times)
