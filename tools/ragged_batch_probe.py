"""The mixing model's prediction (bf16 and fp32) at batches whose tile count is no multiple of eight: time per call.
A/B against a library without the padded grids through FOV_LIB_PATH.  usage: python tools/ragged_batch_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longterm360fov_amd.models import OthersMixingSeq2Seq  # noqa: E402
from oracle import fov_oracle as O  # noqa: E402

H, T_in, T_out, U = 256, 10, 10, 34
d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
F_enc = O.synthetic_batch(1234, 2, T_in, T_out, num_others=U - 1)[0].shape[2]
for dtype in ("bf16", "f32"):
    m = OthersMixingSeq2Seq(num_encoder_tokens=F_enc, latent_dim=H, num_user=U, recurrent_activation="sigmoid", seed=1, dtype=dtype)
    for B in (100, 200, 330, 512):
        enc, dec0, tgt, oth = O.synthetic_batch(1234, B, T_in, T_out, num_others=U - 1)
        a = [d(enc), d(oth), d(dec0)]
        for _ in range(10):
            m.predict_device(*a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            m.predict_device(*a)
        e1.record()
        torch.cuda.synchronize()
        print("%s B=%3d (%2d tiles): %.1f us per prediction" % (dtype, B, (B + 15) // 16, e0.elapsed_time(e1) / 200 * 1e3))
