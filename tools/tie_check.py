"""One train step at a given geometry / data seed, HIP vs the fp32 oracle vs the fp64 oracle, with the tie proof of
tests/test_gpu_fuzz.py: does a discrete decision (ReLU mask bit / pooling arg-max of the trainable U-Net) differ between the HIP
forward and the fp32 oracle?   usage: tools/tie_check.py B H W data_seed [disc|full]"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from oracle import popcorn_oracle as O
from popcorn_amd import ops
from popcorn_amd.data import stats
from popcorn_amd.data.synthetic import make_raw_batch
from popcorn_amd.model import POPCORN
from popcorn_amd.model.popcorn import pad_geometry
from popcorn_amd.train import FusedTrainStep

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
loss = tr.step({"input": x, "admin_mask": batch["admin_mask"].cuda(), "census_idx": batch["census_idx"].cuda(), "y": batch["y"].cuda()})
cpu = {"input": x_ref, "admin_mask": batch["admin_mask"], "census_idx": batch["census_idx"], "y": batch["y"]}
torch.manual_seed(3)
with O.TieProbe() as probe32:
    ref_loss, _, ref_grads, _ = O.train_step_grads(sd, cpu)
rel = lambda a, r: ((a.double() - r.double()).abs().max() / max(r.abs().max().item(), 1e-3)).item()  # noqa: E731
errs = {n: rel(tr.grads[n].cpu(), r) for n, r in ref_grads.items()}
worst = max(errs, key=errs.get)
print(f"HIP vs fp32 oracle: worst {errs[worst]:.2e} at {worst}; loss {loss[0].item():.7f} vs {ref_loss.item():.7f}")
model2 = POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
model2.load_state_dict(sd)
pt, pb, pl, pr = pad_geometry(H, W, False)
_, saved = model2.engines()[0].forward(x, pt, pl, H + pt + pb, W + pl + pr, save=True)
hip_acts, hip_pools = [], []
for s in ("sar_stream", "optical_stream"):
    sv = saved[s]
    hip_acts += [sv[k].cpu() for k in ("a1", "a2", "b1", "b2", "c1", "c2", "e1", "e2", "f1")]
    f0 = 0 if s == "sar_stream" else 8
    hip_acts.append(saved["feats"][:, f0:f0 + 8].cpu())
    hip_pools += [sv["a2"].cpu(), sv["b2"].cpu()]
print("decisions that differ between the HIP forward and the fp32 oracle:", probe32.decisions_differ(hip_acts, hip_pools))
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
cpu64 = {k: (v.double() if v.is_floating_point() else v) for k, v in cpu.items()}
torch.manual_seed(3)
l64, _, g64, _ = O.train_step_grads(sd64, cpu64)
print(f"vs fp64 oracle: HIP {max(rel(tr.grads[n].cpu(), g64[n]) for n in g64):.2e}, fp32 oracle {max(rel(ref_grads[n], g64[n]) for n in g64):.2e}; "
      f"loss HIP - fp64 {loss[0].item() - l64.item():.2e}")
