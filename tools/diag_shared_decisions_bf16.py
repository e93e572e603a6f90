"""Diagnostic (bf16 mode): the golden train step -- HIP-bf16 gradients against the bf16 restatement of the oracle, as it is and with the
oracle forced to the HIP forward's ReLU / arg-max decisions (O.ForceDecisions): how much of the bf16 tolerance band is ties?
    python tools/diag_shared_decisions_bf16.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from oracle import popcorn_oracle as O
from popcorn_amd import _lib as L
from popcorn_amd.model import POPCORN
from popcorn_amd.model.popcorn import pad_geometry
from popcorn_amd.train import FusedTrainStep

G = os.path.join("tests", "golden")
g = np.load(os.path.join(G, "g5_train.npz"))
s = {k: torch.from_numpy(g[k]) for k in ("input", "admin_mask", "census_idx", "y")}
rel = lambda a, b: ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()  # noqa: E731
torch.manual_seed(1600)
m = POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
with O.bf16_mode():
    torch.manual_seed(3)
    _, _, g16, _ = O.train_step_grads(sd, dict(s))
torch.manual_seed(3)
_, _, g32, _ = O.train_step_grads(sd, dict(s))
m.set_precision("bf16")
tr = FusedTrainStep(m, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01)
torch.manual_seed(3)
tr.step({k: v.cuda() for k, v in s.items()})
torch.cuda.synchronize()
hip = {n: tr.grads[n].cpu() for n in tr.grads}
eh = {n: rel(hip[n], g16[n]) for n in g16}
print("unforced: HIP-bf16 vs bf16 oracle worst %.2e median %.2e; bf16 oracle vs fp32 oracle worst %.2e" % (
    max(eh.values()), float(np.median(list(eh.values()))), max(rel(g16[n], g32[n]) for n in g16)))
# the HIP forward's decisions (bf16 mode): saved activations of a fresh bf16 model
m2 = POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
m2.load_state_dict(sd)
m2.set_precision("bf16")
x = s["input"].cuda()
H, W = x.shape[2:]
pt, pb, pl, pr = pad_geometry(H, W, False)
with L.precision("bf16"):
    _, saved = m2.engines()[0].forward(x, pt, pl, H + pt + pb, W + pl + pr, save=True)
acts, pools = [], []
for st in ("sar_stream", "optical_stream"):
    sv = saved[st]
    acts += [sv[k].float().cpu().contiguous() for k in ("a1", "a2", "b1", "b2", "c1", "c2", "e1", "e2", "f1")]
    f0 = 0 if st == "sar_stream" else 8
    acts.append(saved["feats"][:, f0:f0 + 8].float().cpu().contiguous())
    pools += [sv["a2"].float().cpu().contiguous(), sv["b2"].float().cpu().contiguous()]
with O.bf16_mode():
    torch.manual_seed(3)
    with O.ForceDecisions(acts, pools) as f:
        _, _, gf, _ = O.train_step_grads(sd, dict(s))
ef = {n: rel(hip[n], gf[n]) for n in gf}
print("forced:   HIP-bf16 vs bf16 oracle under the HIP forward's decisions worst %.2e median %.2e; flips %s" % (
    max(ef.values()), float(np.median(list(ef.values()))), f.flips))
print("worst tensors forced:", sorted(((round(v, 5), n.replace("unetmodel.", "")) for n, v in ef.items()), reverse=True)[:5])
