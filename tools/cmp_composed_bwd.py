import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from popcorn_amd import engine as E
from popcorn_amd.model import POPCORN
from popcorn_amd.train import FusedTrainStep
g = np.load("tests/golden/g5_train.npz")
s = {k: torch.from_numpy(g[k]).cuda() for k in ("input", "admin_mask", "census_idx", "y")}
grads = {}
for flag in (True, False):
    E.COMPOSED_UP = flag
    torch.manual_seed(1600)
    m = POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
    tr = FusedTrainStep(m, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01)
    torch.manual_seed(1700)
    tr.step(dict(s))
    torch.cuda.synchronize()
    grads[flag] = {k: v.clone() for k, v in tr.grads.items()}
rows = []
for k in grads[True]:
    a, b = grads[True][k], grads[False][k]
    rows.append((((a - b).abs().max() / max(b.abs().max().item(), 1e-3)).item(), k))
for e, k in sorted(rows, reverse=True)[:60]:
    print(f"{e:9.2e} {k}")
