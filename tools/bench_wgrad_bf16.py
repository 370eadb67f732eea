"""Weight-gradient products of the bf16 training step at the config-5 shapes (512 sequences, T 10, H 256), each called N times:
run under `rocprofv3 --kernel-trace --stats` for the kernels' own durations; prints the results' agreement with torch fp64.
FOV_GEMM_BF16_SHALLOW=1 selects the one-stage-in-flight kernel."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longterm360fov_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
TB, H = 5120, 256
h1 = torch.randn(TB, H, device=dev, generator=g)
h2 = torch.randn(TB, H, device=dev, generator=g)
dz = torch.randn(TB, 4 * H, device=dev, generator=g) * 0.1
out2 = torch.zeros((2 * H + 1) * 4 * H, device=dev)
out1 = torch.zeros((H + 1) * 4 * H, device=dev)
sc = ops.Scratch()
n = int(os.environ.get("N", "20"))
for _ in range(n):
    ops.wgrad_fused(h1, h2, dz, out2, scratch=sc, dtype="bf16")
    ops.wgrad_fused(h1, None, dz, out1, scratch=sc, dtype="bf16")
torch.cuda.synchronize()
bf = lambda t: t.to(torch.bfloat16).to(torch.float64)
ref2 = torch.cat([bf(h1), bf(h2)], 1).t() @ bf(dz)
got2 = out2[:2 * H * 4 * H].view(2 * H, 4 * H).double()
print("fused [h1|h2|1]^T dz: max err %.3e of %.3e; bias err %.3e" % ((got2 - ref2).abs().max().item(), ref2.abs().max().item(),
      (out2[2 * H * 4 * H:].double() - dz.double().sum(0)).abs().max().item()))
ref1 = bf(h1).t() @ bf(dz)
print("fused [h1|1]^T dz:    max err %.3e" % ((out1[:H * 4 * H].view(H, 4 * H).double() - ref1).abs().max().item()))
# encoder layer form: (B,T) rows with the time shift
B, T = 512, 10
x = h1.view(B, T, H); hs = h2.view(B, T, H); dzb = dz.view(B, T, 4 * H)
flat = torch.zeros((2 * H + 1) * 4 * H, device=dev)
dK, dR, db = flat[:H * 4 * H].view(H, 4 * H), flat[H * 4 * H:2 * H * 4 * H].view(H, 4 * H), flat[2 * H * 4 * H:]
for _ in range(n):
    ops.lstm_seq_wgrad(x, hs, dzb, dK=dK, dR=dR, db=db, scratch=sc, dtype="bf16")
torch.cuda.synchronize()
refR = torch.einsum("bth,btn->hn", bf(hs[:, :-1]), bf(dzb[:, 1:]))
print("encoder dR (shifted): max err %.3e of %.3e" % ((dR.double() - refR).abs().max().item(), refR.abs().max().item()))
# encoder layer 1: 6-wide input (dK through the skinny kernel), dR | db fused
x6 = torch.randn(B, T, 6, device=dev, generator=g)
flat6 = torch.zeros((6 + H + 1) * 4 * H, device=dev)
dK6, dR6, db6 = flat6[:6 * 4 * H].view(6, 4 * H), flat6[6 * 4 * H:(6 + H) * 4 * H].view(H, 4 * H), flat6[(6 + H) * 4 * H:]
for _ in range(n):
    ops.lstm_seq_wgrad(x6, hs, dzb, dK=dK6, dR=dR6, db=db6, scratch=sc, dtype="bf16")
torch.cuda.synchronize()
ref6 = torch.einsum("btf,btn->fn", x6.double(), dzb.double())
print("encoder dK (6-wide):  max err %.3e of %.3e" % ((dK6.double() - ref6).abs().max().item(), ref6.abs().max().item()))
