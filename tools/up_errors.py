"""Accuracy of the composed Up-block kernels (forward / backward) against float64 torch on random operands: max error relative to the
largest reference value, where it sits, and the same for the two-launch reference path (transposed conv + two-source conv)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch, torch.nn.functional as F
from popcorn_amd import ops, _lib as L

torch.manual_seed(0)
g = torch.Generator().manual_seed(5)
B, Cs, H, W = 2, 8, 128, 128
Cz = Cs
skip = torch.randn(B, Cs, H, W, generator=g); z = torch.relu(torch.randn(B, Cz, H // 2, W // 2, generator=g))
w = torch.randn(8, Cs + Cz, 3, 3, generator=g) * 0.1; wt = torch.randn(Cz, Cz, 2, 2, generator=g) * 0.2; bt = torch.randn(Cz, generator=g)
bias = torch.zeros(8)
u = F.conv_transpose2d(z.double(), wt.double(), bt.double(), stride=2)
ref = torch.relu(F.conv2d(torch.cat([skip.double(), u], 1), w.double(), None, padding=1))
d = lambda t: t.cuda()
out = torch.empty(B, 8, H, W, device="cuda")
bz = torch.zeros(8, device="cuda")
ops.conv3x3_up_fwd_group([{"skip": d(skip), "z": d(z), "w": d(w), "wt": d(wt), "bt": d(bt), "bn": L.bn(bz), "out": out, "_k": bz}])
e = (out.cpu() - ref.float()).abs()
idx = torch.nonzero(e == e.max())[0].tolist()
print("composed fwd: max rel err", (e.max() / ref.abs().max()).item(), "at", idx, "interior-only", (e[:, :, 4:-4, 4:-4].max() / ref.abs().max()).item())
# two-launch path
u32 = torch.empty(B, Cz, H, W, device="cuda")
ops.convt2x2_group([{"x": d(z), "w": d(wt), "bias": d(bt), "out": u32}])
out2 = torch.empty(B, 8, H, W, device="cuda")
ops.conv3x3_fwd_group([{"a": d(skip), "b": u32, "w": d(w), "bn": L.bn(bz), "out": out2}])
e2 = (out2.cpu() - ref.float()).abs()
print("two-launch fwd: max rel err", (e2.max() / ref.abs().max()).item())
# ---- backward
zz = z.double().requires_grad_(True); ww = w.double().requires_grad_(True); wtt = wt.double().requires_grad_(True); btt = bt.double().requires_grad_(True)
G = torch.randn(B, 8, H, W, generator=g)
uu = F.conv_transpose2d(zz, wtt, btt, stride=2)
yy = F.conv2d(uu, ww[:, Cs:], None, padding=1)
(yy * G.double()).sum().backward()
one = torch.ones(Cz, device="cuda")
pr = {"g": d(G), "z": d(z), "z_bn": L.bn(None, one, one * 0, one * 0, one - 1e-5), "gz": torch.empty(B, Cz, H // 2, W // 2, device="cuda"),
      "w": d(w), "wt": d(wt), "bt": d(bt), "dw": torch.zeros(8, Cs + Cz, 3, 3, device="cuda"), "dwt": torch.zeros(Cz, Cz, 2, 2, device="cuda"),
      "dbt": torch.zeros(Cz, device="cuda"), "_k": one}
slots = ops.conv3x3_up_fwd_group([{"skip": d(skip), "z": d(z), "w": d(w), "wt": d(wt), "bt": d(bt), "bn": L.bn(bz), "out": out}])
pr["fwd_ws"] = slots[0]
ops.conv3x3_up_bwd_group([pr])
torch.cuda.synchronize()
rel = lambda a, r: ((a.cpu().double() - r).abs().max() / r.abs().max()).item()
print("gz ", rel(pr["gz"], zz.grad * (z > 0)), " dw", rel(pr["dw"][:, Cs:], ww.grad[:, Cs:]), " dwt", rel(pr["dwt"], wtt.grad), " dbt", rel(pr["dbt"], btt.grad))
# reference two-launch backward: dgrad + convT bwd ... (weight gradient of the up block through autograd in fp32 on the GPU for scale)
z32 = d(z).requires_grad_(True); w32 = d(w).requires_grad_(True); wt32 = d(wt).requires_grad_(True); bt32 = d(bt).requires_grad_(True)
y32 = F.conv2d(F.conv_transpose2d(z32, wt32, bt32, stride=2), w32[:, Cs:], None, padding=1)
(y32 * d(G)).sum().backward()
print("torch fp32 GPU autograd: gz", rel(z32.grad * (d(z) > 0), zz.grad * (z > 0)), " dw", rel(w32.grad[:, Cs:], ww.grad[:, Cs:]), " dwt", rel(wt32.grad, wtt.grad), " dbt", rel(bt32.grad, btt.grad))
