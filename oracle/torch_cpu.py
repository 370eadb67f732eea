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

    def train_step(self, enc_in, dec_in, target):
        """One teacher-forced training step (FoV_seq2seq.py:82-103,112-117: MSE, Adam) with torch autograd on the CPU."""
        torch.set_num_threads(self.threads)
        if not hasattr(self, "_dec"):
            H = self.cell.hidden_size
            self._dec = torch.nn.LSTM(self.cell.input_size, H, batch_first=True)
            with torch.no_grad():
                self._dec.weight_ih_l0.copy_(self.cell.weight_ih); self._dec.weight_hh_l0.copy_(self.cell.weight_hh)
                self._dec.bias_ih_l0.copy_(self.cell.bias_ih); self._dec.bias_hh_l0.zero_()
            params = list(self.enc.parameters()) + list(self._dec.parameters()) + list(self.lin.parameters())
            self._opt = torch.optim.Adam(params, lr=1e-3, eps=1e-7)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        self._opt.zero_grad(set_to_none=True)
        _, st = self.enc(t(enc_in))
        hs, _ = self._dec(t(dec_in), st)
        loss = torch.mean((torch.tanh(self.lin(hs)) - t(target)) ** 2)
        loss.backward()
        self._opt.step()
        return float(loss)

    @torch.no_grad()
    def decode(self, enc_in, dec_in0, T_out):
        torch.set_num_threads(self.threads)
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


class Seq2SeqSgemmCPU:
    """The same graph as a hand-arranged sgemm loop (what BASELINE.md section 4 asks for as leg 1): the encoder's input
    projection for ALL steps as one (B*T, F) x (F, 4H) sgemm, then per step one (B, H) x (H, 4H) sgemm (torch.addmm ->
    MKL / oneDNN sgemm) and fused elementwise gates; the decoder per step one (B, F_dec + H) x (F_dec + H, 4H) sgemm on
    [y | h].  Gate order i, f, c, o as in Keras; sigmoid recurrent activation."""

    def __init__(self, w, threads=None):
        self.threads = int(threads or usable_cores())
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        self.eK, self.eR, self.eb = t(w["enc_K"]), t(w["enc_R"]), t(w["enc_b"])
        self.dW = torch.cat([t(w["dec_K"]), t(w["dec_R"])], 0)      # [K ; R]: (F_dec + H, 4H)
        self.db = t(w["dec_b"])
        self.W, self.b = t(w["dense_W"]), t(w["dense_b"])
        self.H = self.eR.shape[0]

    @staticmethod
    def _cell(z, c, H):
        i, f, g, o = torch.sigmoid(z[:, :H]), torch.sigmoid(z[:, H:2 * H]), torch.tanh(z[:, 2 * H:3 * H]), torch.sigmoid(z[:, 3 * H:])
        c = f * c + i * g
        return o * torch.tanh(c), c

    @torch.no_grad()
    def decode(self, enc_in, dec_in0, T_out):
        torch.set_num_threads(self.threads)
        x = torch.from_numpy(np.ascontiguousarray(enc_in, dtype=np.float32))
        y = torch.from_numpy(np.ascontiguousarray(dec_in0[:, 0], dtype=np.float32))
        B, T, F = x.shape
        H = self.H
        h = torch.zeros((B, H))
        c = torch.zeros((B, H))
        if T > 0:
            zx = torch.addmm(self.eb, x.reshape(B * T, F), self.eK).reshape(B, T, 4 * H)
            for t in range(T):
                h, c = self._cell(torch.addmm(zx[:, t], h, self.eR), c, H)
        outs = []
        for _ in range(T_out):
            h, c = self._cell(torch.addmm(self.db, torch.cat([y, h], 1), self.dW), c, H)
            y = torch.tanh(torch.addmm(self.b, h, self.W))
            outs.append(y)
        return torch.stack(outs, 1).numpy()


class OthersMixingCPU:
    """given_others_gt_mean_var_seq2seq.py:98-130,203-299 (2-layer encoder, unrolled 2-layer decoder without teacher forcing,
    Dense(6, tanh) + others mixing Dense(204 -> 6, tanh), user-major flatten with the prediction last) on torch CPU ops:
    nn.LSTM for the encoder stack, two LSTMCell + two Linear per decoder step.  Forward only (inference baseline)."""

    def __init__(self, w, threads=None):
        self.threads = int(threads or usable_cores())
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        F, H4 = w["enc1_K"].shape
        H, O = H4 // 4, w["dense_W"].shape[1]
        self.enc = torch.nn.LSTM(F, H, num_layers=2, batch_first=True)
        self.c1, self.c2 = torch.nn.LSTMCell(O, H), torch.nn.LSTMCell(H, H)
        self.lin, self.mix = torch.nn.Linear(H, O), torch.nn.Linear(w["mix_W"].shape[0], O)
        with torch.no_grad():
            for l, name in ((0, "enc1"), (1, "enc2")):
                getattr(self.enc, "weight_ih_l%d" % l).copy_(t(w[name + "_K"].T))
                getattr(self.enc, "weight_hh_l%d" % l).copy_(t(w[name + "_R"].T))
                getattr(self.enc, "bias_ih_l%d" % l).copy_(t(w[name + "_b"]))
                getattr(self.enc, "bias_hh_l%d" % l).zero_()
            for cell, name in ((self.c1, "dec1"), (self.c2, "dec2")):
                cell.weight_ih.copy_(t(w[name + "_K"].T))
                cell.weight_hh.copy_(t(w[name + "_R"].T))
                cell.bias_ih.copy_(t(w[name + "_b"]))
                cell.bias_hh.zero_()
            self.lin.weight.copy_(t(w["dense_W"].T)); self.lin.bias.copy_(t(w["dense_b"]))
            self.mix.weight.copy_(t(w["mix_W"].T)); self.mix.bias.copy_(t(w["mix_b"]))

    def _forward(self, enc_in, others, dec_in0):
        x = torch.from_numpy(np.ascontiguousarray(enc_in, dtype=np.float32))
        oth = torch.from_numpy(np.ascontiguousarray(others, dtype=np.float32))
        y = torch.from_numpy(np.ascontiguousarray(dec_in0[:, 0], dtype=np.float32))
        _, (h, c) = self.enc(x)
        h1, c1, h2, c2 = h[0], c[0], h[1], c[1]
        B, T_out = oth.shape[0], oth.shape[1]
        outs = []
        for t in range(T_out):
            h1, c1 = self.c1(y, (h1, c1))
            h2, c2 = self.c2(h1, (h2, c2))
            p = torch.tanh(self.lin(h2))
            y = torch.tanh(self.mix(torch.cat([oth[:, t].reshape(B, -1), p], 1)))
            outs.append(y)
        return torch.stack(outs, 1)

    def train_step(self, enc_in, others, dec_in0, target):
        """One training step of the unrolled graph (given_others_gt_mean_var_seq2seq.py:308,494-506: MSE, Adam), torch autograd."""
        torch.set_num_threads(self.threads)
        if not hasattr(self, "_opt"):
            params = [q for m in (self.enc, self.c1, self.c2, self.lin, self.mix) for q in m.parameters()]
            self._opt = torch.optim.Adam(params, lr=1e-3, eps=1e-7)
        self._opt.zero_grad(set_to_none=True)
        loss = torch.mean((self._forward(enc_in, others, dec_in0) - torch.from_numpy(np.ascontiguousarray(target, dtype=np.float32))) ** 2)
        loss.backward()
        self._opt.step()
        return float(loss)

    @torch.no_grad()
    def predict(self, enc_in, others, dec_in0):
        torch.set_num_threads(self.threads)
        return self._forward(enc_in, others, dec_in0).numpy()


def convlstm_encoder_cpu(x, layers, threads, act="hard_sigmoid"):
    """convlstm_seq2seq.py:100-126 (3 stacked ConvLSTM2D, return_sequences) on torch CPU ops: per layer and step ONE
    conv2d over [x_t | h_{t-1}] (NCHW, 'same' padding) + the gates.  x (B,T,H,W,C) float32; layers = [(K (kh,kw,C,4F),
    R (kh,kw,F,4F), b (4F,)), ...] in Keras layout, gate order i,f,c,o.  Returns the last layer's sequence (B,T,H,W,F)."""
    import torch.nn.functional as Fn
    torch.set_num_threads(int(threads))
    rec = (lambda z: torch.clamp(0.2 * z + 0.5, 0.0, 1.0)) if act == "hard_sigmoid" else torch.sigmoid
    with torch.no_grad():
        seq = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).permute(1, 0, 4, 2, 3).contiguous()   # (T,B,C,H,W)
        for K, R, b in layers:
            W = torch.from_numpy(np.concatenate([K, R], axis=2).astype(np.float32)).permute(3, 2, 0, 1).contiguous()   # (4F, C+F, kh, kw)
            bt = torch.from_numpy(np.ascontiguousarray(b, dtype=np.float32))
            Fh = R.shape[2]
            pad = (K.shape[0] // 2, K.shape[1] // 2)
            h = torch.zeros((seq.shape[1], Fh) + tuple(seq.shape[3:]))
            c = torch.zeros_like(h)
            out = []
            for t in range(seq.shape[0]):
                z = Fn.conv2d(torch.cat([seq[t], h], 1), W, bt, padding=pad)
                i, f, g, o = rec(z[:, :Fh]), rec(z[:, Fh:2 * Fh]), torch.tanh(z[:, 2 * Fh:3 * Fh]), rec(z[:, 3 * Fh:])
                c = f * c + i * g
                h = o * torch.tanh(c)
                out.append(h)
            seq = torch.stack(out, 0)
        return seq.permute(1, 0, 3, 4, 2).contiguous().numpy()


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


def sgemm_leg_in_child(enc, dec0, w, T_out, threads, budget_s, wall_limit_s):
    """Time Seq2SeqSgemmCPU.decode with `threads` threads in a CHILD process that the caller can give up on: on a shared
    8-GPU host a pool over every visible hardware thread (256) can take minutes per pass (oversubscription), and a torch op
    cannot be interrupted from inside its own process.  Returns (median_s, passes), or None when the child did not finish
    within `wall_limit_s` (it is killed by its PID)."""
    import json
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "leg.npz")
        np.savez(path, enc=enc, dec0=dec0, **{"w_" + k: v for k, v in w.items()})
        env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
        try:
            r = subprocess.run([sys.executable, "-m", "oracle.torch_cpu", path, str(int(T_out)), str(int(threads)), str(float(budget_s))],
                               cwd=root, env=env, capture_output=True, text=True, timeout=wall_limit_s)
        except subprocess.TimeoutExpired:
            return None
    if r.returncode != 0:
        raise RuntimeError("sgemm leg child failed: " + r.stderr[-400:])
    out = json.loads(r.stdout.strip().splitlines()[-1])
    return out["median_s"], out["passes"]


if __name__ == "__main__":   # the child of sgemm_leg_in_child
    import json
    import sys
    _path, _T_out, _thr, _budget = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
    _z = np.load(_path)
    _w = {k[2:]: _z[k] for k in _z.files if k.startswith("w_")}
    _sg = Seq2SeqSgemmCPU(_w, threads=_thr)
    _med, _n = timed_median(lambda: _sg.decode(_z["enc"], _z["dec0"], _T_out), budget_s=_budget, min_iters=3, warmup=1)
    print(json.dumps({"median_s": _med, "passes": _n}))
