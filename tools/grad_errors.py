"""Per-tensor gradient error of the HIP path against the fp64 oracle on the golden train sample (fixture g5), next to the fp32 oracle's
own distance from fp64:   python3 tools/grad_errors.py     (POPCORN_COMPOSED_UP=0/1 etc. select the path)"""
import os
import sys
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from oracle import popcorn_oracle as O
from popcorn_amd.model import POPCORN
from popcorn_amd.utils.losses import get_loss

g = np.load("tests/golden/g5_train.npz")
torch.manual_seed(1600)
m = POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
cpu = {k: torch.from_numpy(g[k]) for k in ("input", "admin_mask", "census_idx", "y")}
torch.manual_seed(1700)
s = {k: v.cuda() for k, v in cpu.items()}
m.train()
o = m(s, train=True, padding=False, sparse=True)
loss, _ = get_loss(o, s, scale=o["scale"], loss=["log_l1_loss"], lam=[1.0], scale_regularization=0.01, tag="weak")
(loss * 100.0).backward()
hip = {n: p.grad.cpu() for n, p in m.named_parameters() if p.grad is not None}
torch.manual_seed(1700)
_, _, g32, _ = O.train_step_grads(sd, dict(cpu))
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
cpu64 = {k: (v.double() if v.is_floating_point() else v) for k, v in cpu.items()}
torch.manual_seed(1700)
_, _, g64, _ = O.train_step_grads(sd64, cpu64)
rel = lambda a, r: ((a.double() - r.double()).abs().max() / max(r.abs().max().item(), 1e-3)).item()
rows = sorted(((rel(hip[n], g64[n]), rel(g32[n], g64[n]), g64[n].abs().max().item(), n) for n in g64), reverse=True)
for eh, e32, mag, n in rows[:14]:
    print(f"{n:60s} |g|max {mag:9.3e}  HIP-fp64 {eh:9.2e}  fp32oracle-fp64 {e32:9.2e}")
