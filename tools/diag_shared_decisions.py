"""Diagnostic: one train step on a census-region batch -- HIP gradients against the fp32 / fp64 oracle, unforced and under the HIP forward's
decisions (O.ForceDecisions), and the head's near-tie units.   python tools/diag_shared_decisions.py B H W seed [disc|full]"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from oracle import popcorn_oracle as O
from popcorn_amd import ops
from popcorn_amd.data import stats
from popcorn_amd.data.synthetic import make_raw_batch
from popcorn_amd.model import POPCORN
from popcorn_amd.train import FusedTrainStep
from tests.tie_adjudication import forced_decision_distance, hip_decision_sites, rel

B, H, W, seed = (int(v) for v in sys.argv[1:5])
region = sys.argv[5] if len(sys.argv) > 5 else "disc"
torch.manual_seed(1600)
model = POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
batch = make_raw_batch(B, H, W, seed=seed, region=region)
x_ref = O.select_normalize(batch["raw"])
x = ops.select_normalize(batch["raw"].cuda(), stats.BAND6, stats.MEAN6, stats.STD6)
tr = FusedTrainStep(model, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01)
torch.manual_seed(3)
tr.step({"input": x, "admin_mask": batch["admin_mask"].cuda(), "census_idx": batch["census_idx"].cuda(), "y": batch["y"].cuda()})
torch.cuda.synchronize()
hip = {n: tr.grads[n].cpu() for n in tr.grads}
cpu = {"input": x_ref, "admin_mask": batch["admin_mask"], "census_idx": batch["census_idx"], "y": batch["y"]}
torch.manual_seed(3)
_, _, g32, _ = O.train_step_grads(sd, dict(cpu))
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
cpu64 = {k: (v.double() if v.is_floating_point() else v) for k, v in cpu.items()}
torch.manual_seed(3)
_, o64, g64, _ = O.train_step_grads(sd64, dict(cpu64))
print("unforced: HIP vs fp32 oracle %.2e, HIP vs fp64 %.2e, fp32 oracle vs fp64 %.2e" % (
    max(rel(hip[n], g32[n]) for n in g32), max(rel(hip[n], g64[n]) for n in g64), max(rel(g32[n], g64[n]) for n in g64)))
for fp64 in (False, True):
    wf, name, flips, _ = forced_decision_distance(sd, cpu, x, hip, 3, fp64=fp64)
    print("forced (%s oracle): %.2e (%s) flips %s" % ("fp64" if fp64 else "fp32", wf, name, flips))
errs = sorted(((rel(hip[n], g64[n]), n) for n in g64), reverse=True)[:6]
print("worst tensors vs fp64 (unforced):", [(f"{e:.1e}", n.replace("unetmodel.", "")) for e, n in errs])
# head near-ties: hidden pre-activations of the fp64 head on the HIP features at the selected pixels
import torch.nn.functional as F
_, _, feats = hip_decision_sites(sd, x)
torch.manual_seed(3)
with torch.no_grad():
    fo = O.popcorn_forward(sd64, dict(cpu64), padding=False, sparse=True, return_features=True)
mask = fo["mask"]
xx = feats.double().permute(1, 0, 2, 3).reshape(feats.shape[1], -1, 1)[:, mask.reshape(-1)]
for i in (0, 2, 4):
    pre = F.conv2d(xx, sd64[f"head.{i}.weight"], sd64[f"head.{i}.bias"])
    a = pre.abs()
    print(f"head layer {i}: {pre.numel()} units, |pre| mean {a.mean():.3e}, smallest {a.min():.3e}, units with |pre| < 1e-6 x mean: {int((a < 1e-6 * a.mean()).sum())}, < 1e-5: {int((a < 1e-5 * a.mean()).sum())}")
    xx = F.relu(pre)
out = F.conv2d(xx, sd64["head.6.weight"], sd64["head.6.bias"])
print("final: |out0| smallest %.3e (mean %.3e)" % (out[0].abs().min(), out[0].abs().mean()))

from tests.tie_adjudication import head_near_ties
cands = head_near_ties(sd, feats, mask, 1e-5)[:8]
print("head candidates (layer, unit, column):", cands)
base = forced_decision_distance(sd, cpu, x, hip, 3)[0]
for c in cands:
    wf = forced_decision_distance(sd, cpu, x, hip, 3, head_flips=[c])[0]
    print("  flip", c, "-> %.2e" % wf, "<-- explains it" if wf < 0.3 * base else "")
