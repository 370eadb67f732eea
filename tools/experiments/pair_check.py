"""Cross-check of the two-tiles-per-workgroup fused kernel (lstm_pair.hip) against the one-tile kernels and the fp64 oracle.
usage: python tools/pair_check.py [B ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longterm360fov_amd import ops  # noqa: E402
from oracle import fov_oracle as O  # noqa: E402


def run(B, T_in, T_out, H=256, act="sigmoid"):
    w = O.init_seq2seq(1, H=H, bias_noise=0.05)
    enc, dec0, _ = O.synthetic_batch(2, B, T_in, T_out)
    dw = {k: torch.from_numpy(v).cuda() for k, v in w.items()}
    e, d = torch.from_numpy(enc).cuda(), torch.from_numpy(dec0).cuda()
    outs = {}
    for name, env in (("pair", None), ("single", "1")):
        if env:
            os.environ.pop("FOV_PAIR", None)
        else:
            os.environ["FOV_PAIR"] = "1"
        ws = ops.Workspace()
        out = ops.seq2seq_decode(e, d, dw, T_out, impl="cluster", workspace=ws, act=act)
        torch.cuda.synchronize()
        ws.check()
        outs[name] = out.cpu().numpy()
        t0 = time.perf_counter()
        for _ in range(20):
            ops.seq2seq_decode(e, d, dw, T_out, impl="cluster", workspace=ws, act=act)
        torch.cuda.synchronize()
        ws.check()
        outs[name + "_ms"] = (time.perf_counter() - t0) / 20 * 1e3
    nchk = min(B, 48)
    ref = O.seq2seq_decode(enc[:nchk].astype(np.float64), dec0[:nchk].astype(np.float64),
                           {k: v.astype(np.float64) for k, v in w.items()}, T_out, act=act)
    err = float(np.abs(outs["pair"][:nchk] - ref).max())
    diff = float(np.abs(outs["pair"] - outs["single"]).max())
    print("B=%5d T=%d->%d act=%s: pair vs oracle %.3e, pair vs single %.3e, pair %.4f ms, single %.4f ms" %
          (B, T_in, T_out, act, err, diff, outs["pair_ms"], outs["single_ms"]), flush=True)
    assert err < 2e-5 and diff < 2e-5, (err, diff)


if __name__ == "__main__":
    Bs = [int(a) for a in sys.argv[1:]] or [48, 16, 17, 100, 512, 1024]
    for B in Bs:
        run(B, 6, 5)
    run(1024, 30, 30)
    run(1000, 30, 30, act="hard_sigmoid")
    print("pair_check ok")
