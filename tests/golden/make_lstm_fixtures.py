#!/usr/bin/env python3
"""Golden input/output vectors for the LSTM arithmetic, produced by the fp64 oracle and
cross-checked against torch.nn.LSTM before being written (the reference's Keras/TF cannot be
run anywhere in this pipeline: "parity unpinned", see oracle/fov_oracle.py)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import fov_oracle as O  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lstm_small.npz")


def main():
    H, B, T_in, T_out, U = 32, 5, 6, 4, 4
    w = O.init_seq2seq(21, H=H, bias_noise=0.1)
    wm = O.init_others_mixing(22, H=H, num_user=U, bias_noise=0.1)
    enc, dec0, tgt, oth = O.synthetic_batch(23, B, T_in, T_out, num_others=U - 1)
    d = lambda a: a.astype(np.float64)
    w64 = {k: d(v) for k, v in w.items()}
    wm64 = {k: d(v) for k, v in wm.items()}
    out = {"enc": enc, "dec0": dec0, "tgt": tgt, "oth": oth, "T_out": np.int64(T_out)}
    out.update({"w_" + k: v for k, v in w.items()})
    out.update({"wm_" + k: v for k, v in wm.items()})
    dec_in = np.concatenate([dec0, tgt[:, :-1]], axis=1)
    for act in (0, 1):
        out["decode_act%d" % act] = O.seq2seq_decode(d(enc), d(dec0), w64, T_out, act).astype(np.float32)
        out["tf_act%d" % act] = O.seq2seq_teacher_forced(d(enc), d(dec_in), w64, act).astype(np.float32)
    out["mix_act0"] = O.others_mixing_forward(d(enc), d(oth), d(dec0), wm64, 0).astype(np.float32)
    # independent check of the teacher-forced graph with torch before committing the vectors
    with torch.no_grad():
        e = torch.nn.LSTM(90, H, batch_first=True).double(); dd = torch.nn.LSTM(6, H, batch_first=True).double()
        for m, p in ((e, "enc"), (dd, "dec")):
            m.weight_ih_l0.copy_(torch.from_numpy(w64[p + "_K"].T)); m.weight_hh_l0.copy_(torch.from_numpy(w64[p + "_R"].T))
            m.bias_ih_l0.copy_(torch.from_numpy(w64[p + "_b"])); m.bias_hh_l0.zero_()
        _, st = e(torch.from_numpy(d(enc)))
        hs, _ = dd(torch.from_numpy(d(dec_in)), st)
        y = torch.tanh(hs @ torch.from_numpy(w64["dense_W"]) + torch.from_numpy(w64["dense_b"])).numpy()
    assert np.abs(y - out["tf_act0"]).max() < 1e-6
    np.savez_compressed(OUT, **out)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
