"""How close is the HIP path to the bounds of the golden-fixture gradient test (tests/test_gpu_model.py::test_train_step_vs_reference_golden)?
Prints e / (2e-4 * max|ref|) per parameter, largest first."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from popcorn_amd.model import POPCORN
from popcorn_amd.utils.losses import get_loss
G = "tests/golden"
g = np.load(os.path.join(G, "g5_train.npz"))
torch.manual_seed(1600)
m = POPCORN(input_channels=6, feature_extractor="DDA", occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
m.train()
sample = {k: torch.from_numpy(g[k]).cuda() for k in ("input", "admin_mask", "census_idx", "y")}
torch.manual_seed(1700)
o = m(sample, train=True, padding=False, sparse=True)
loss, ld = get_loss(o, sample, scale=o["scale"], loss=["log_l1_loss"], lam=[1.0], scale_regularization=0.01, tag="weak")
(loss * 100.0).backward()
rows = []
for n, p in m.named_parameters():
    if p.grad is not None:
        ref = g["step0/grad/" + n]
        e = np.abs(p.grad.cpu().numpy() - ref).max()
        rows.append((e / (2e-4 * max(np.abs(ref).max(), 1e-3)), n))
for r, n in sorted(rows, reverse=True)[:8]:
    print(f"{r:.3f}  {n}")
