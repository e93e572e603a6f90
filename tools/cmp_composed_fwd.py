import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from popcorn_amd import engine as E
from popcorn_amd.model import POPCORN
g = np.load("tests/golden/g5_train.npz")
torch.manual_seed(1600)
m = POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
eng_u, eng_b = m.engines()
X = torch.from_numpy(g["input"]).cuda()
H, W = X.shape[2:]
print(X.shape)
outs = {}
for flag in (True, False):
    E.COMPOSED_UP = flag
    (f_b, f_u), (_, saved) = E.forward_multi([eng_b, eng_u], X, 14, 14, H + 28, W + 28, [False, True], logit_only=[True, False])
    torch.cuda.synchronize()
    outs[flag] = (f_b.clone(), f_u.clone(), {s: {k: saved[s][k].clone() for k in ("c1", "c2", "e1", "e2", "f1")} for s in ("sar_stream", "optical_stream")})
for s in ("sar_stream", "optical_stream"):
    for k in ("c1", "c2", "e1", "e2", "f1"):
        a, b = outs[True][2][s][k], outs[False][2][s][k]
        d = (a - b).abs()
        i = torch.nonzero(d == d.max())[0].tolist()
        print(s, k, "max abs diff", d.max().item(), "rel", (d.max() / b.abs().max()).item(), "at", i, "shape", tuple(a.shape))
print("feat", ((outs[True][1] - outs[False][1]).abs().max() / outs[False][1].abs().max()).item(), "logit", ((outs[True][0] - outs[False][0]).abs().max() / outs[False][0].abs().max()).item())
