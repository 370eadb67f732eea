"""Is a training step bound by the host (Python issuing ~20 launches) or by the GPU?  For each trainer: HIP-event time per step
of a long back-to-back region, and the host's own time to ENQUEUE a step when the GPU is kept out of the way (the queue is
drained first, then N steps are issued and the clock stops before any synchronisation).
usage: python3 tools/host_vs_gpu_step.py [mixing_bf16|mixing_f32|a10_gmm|a10_meanvar|config1_train ...]"""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fov_oracle as O  # noqa: E402

d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()


def make(name):
    if name.startswith("mixing"):
        from longterm360fov_amd.training import OthersMixingTrainer
        w = O.init_others_mixing(1234, H=256, num_user=34, bias_noise=0.05)
        enc, dec0, tgt, oth = O.synthetic_batch(1234, 512, 10, 10, num_others=33)
        tr = OthersMixingTrainer(w, dtype="bf16" if name.endswith("bf16") else "f32")
        a = (d(enc), d(oth), d(dec0), d(tgt))
        return lambda: tr.train_step(*a)
    if name.startswith("a10"):
        from longterm360fov_amd.training import TFLSTMTrainer
        rng = np.random.default_rng(11)
        H, B, T, F = 400, 32, 10, 90
        cells = [((rng.standard_normal((Fin + H, 4 * H)) / np.sqrt(Fin + H)).astype(np.float32), np.zeros(4 * H, np.float32)) for Fin in (F, H)]
        if name.endswith("gmm"):
            dims = [H, 64, 128, 256, 200]
            head = {}
            for l in range(4):
                head["fc%d_W" % (l + 1)] = (rng.uniform(-1, 1, (dims[l], dims[l + 1])) * np.sqrt(6.0 / (dims[l] + dims[l + 1]))).astype(np.float32)
                head["fc%d_b" % (l + 1)] = np.zeros(dims[l + 1], np.float32)
            y = d(rng.uniform(-1, 1, (B, 10, 90)).astype(np.float32))
            tr = TFLSTMTrainer(cells, head, head_kind="gmm")
        else:
            head = {}
            for br in ("mu", "var"):
                head[br + "_W1"] = (rng.standard_normal((H, 32)) / 20).astype(np.float32)
                head[br + "_b1"] = np.zeros(32, np.float32)
                head[br + "_W2"] = (rng.standard_normal((32, 3)) / np.sqrt(32)).astype(np.float32)
                head[br + "_b2"] = np.zeros(3, np.float32)
            y = d(rng.uniform(-1, 1, (B, 1, 90)).astype(np.float32))
            tr = TFLSTMTrainer(cells, head)
        x = d(rng.uniform(-1, 1, (B, T, F)).astype(np.float32))
        init = torch.zeros((2, 2, B, H), device="cuda")
        return lambda: tr.train_step(x, y, init)
    from longterm360fov_amd.training import Seq2SeqTrainer
    w = O.init_seq2seq(1234, H=128, bias_noise=0.05)
    enc, dec0, tgt = O.synthetic_batch(1234, 32, 10, 10)
    dec_in = np.concatenate([dec0, tgt[:, :-1]], axis=1)
    tr = Seq2SeqTrainer(w)
    a = (d(enc), d(dec_in), d(tgt))
    return lambda: tr.train_step(*a)


for name in (sys.argv[1:] or ["mixing_bf16", "mixing_f32", "a10_gmm", "a10_meanvar", "config1_train"]):
    step = make(name)
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300):
        step()
    e1.record()
    torch.cuda.synchronize()
    gpu_ms = e0.elapsed_time(e1) / 300
    # host alone: a long kernel first, so that nothing the host enqueues can start - and block the host - while it issues
    blocker = torch.empty(1 << 28, device="cuda")
    host = []
    for _ in range(5):
        torch.cuda.synchronize()
        for _ in range(8):
            blocker.mul_(1.0)          # ~1 ms each: the GPU stays busy while the steps are issued
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        host.append((time.perf_counter() - t0) / 10)
        torch.cuda.synchronize()
    print("%-14s  GPU-side %.4f ms per step (events, 300 back-to-back steps)   host enqueue %.4f ms per step (min of 5 x 10)" %
          (name, gpu_ms, min(host) * 1e3))
