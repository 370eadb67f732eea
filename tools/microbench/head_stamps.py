"""Where do the fused head kernels (mlp_head.hip) spend their time?  Loads the DIAGNOSTIC build (make -C longterm360fov_amd/csrc stamps),
runs each kernel at the mixture head's shape and prints s_memtime deltas of wave 0 of workgroup 0 (100 MHz ticks -> us)."""
import ctypes
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from longterm360fov_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(ROOT, "longterm360fov_amd", "lib", "libfov360_hip_stamps.so")
from longterm360fov_amd import ops  # noqa: E402

B, H, n = 32, 512, 20
rng = np.random.default_rng(0)
dims = [H, 64, 128, 256, 10 * n]
layers = []
for l in range(4):
    layers.append((torch.from_numpy((0.05 * rng.standard_normal((dims[l], dims[l + 1]))).astype(np.float32)).cuda(),
                   torch.zeros(dims[l + 1], device="cuda"), "relu" if l < 3 else None))
gW = [torch.zeros_like(W) for W, _, _ in layers]
gb = [torch.zeros_like(b) for _, b, _ in layers]
h = torch.from_numpy(rng.standard_normal((B, H)).astype(np.float32)).cuda()
y = torch.from_numpy(rng.uniform(-1, 1, (B, 10, 90)).astype(np.float32)).cuda()
for _ in range(3):
    acts = ops.mlp_head_fwd(h, layers, n_mix=n)
    loss, dpre = ops.gmm3d_loss_grad(acts[-1], y, 30, 1.0 / (B * 300))
    ops.mlp_head_bwd(h, layers, acts, dpre, gW, gb)
torch.cuda.synchronize()
L = _lib.lib()
buf = np.zeros((3, 48), dtype=np.uint64)
L.fov_debug_read_mh_stamps.argtypes = [ctypes.c_void_p]
assert L.fov_debug_read_mh_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
for k, name in enumerate(("forward", "backward chain", "mixture loss")):
    st = buf[k].astype(np.int64)
    nz = [i for i in range(48) if st[i]]
    print(name, "total %.2f us" % ((st[nz].max() - st[nz].min()) / 100.0))
    print("   " + "  ".join("%d:%.2f" % (i, (st[i] - st[nz[0]]) / 100.0) for i in nz))
