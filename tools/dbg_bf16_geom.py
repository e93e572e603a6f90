"""bf16 mode on other geometries than the bench tile: per-tensor gradient distance HIP-bf16 vs oracle-bf16 and the rounding band."""
import sys
import torch
sys.path.insert(0, ".")
from oracle import popcorn_oracle as O
from popcorn_amd.data.synthetic import make_raw_batch
from popcorn_amd.data import stats
from popcorn_amd import ops
from popcorn_amd.model import POPCORN
from popcorn_amd.train import FusedTrainStep

B, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
region = sys.argv[4] if len(sys.argv) > 4 else "disc"
dev = torch.device("cuda:0")
torch.manual_seed(1600)
model = POPCORN(input_channels=6, feature_extractor="DDA", occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).to(dev)
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
batch = make_raw_batch(B, H, W, seed=7, region=region)
x_ref = O.select_normalize(batch["raw"])
sample = {"input": x_ref.to(dev), "admin_mask": batch["admin_mask"].to(dev), "census_idx": batch["census_idx"].to(dev), "y": batch["y"].to(dev)}
cpu_sample = {"input": x_ref, "admin_mask": batch["admin_mask"], "census_idx": batch["census_idx"], "y": batch["y"]}
model.set_precision("bf16")
tr = FusedTrainStep(model, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01, use_graph=False)
torch.manual_seed(3)
loss = tr.step(sample)
torch.cuda.synchronize()
torch.manual_seed(3)
l32, o32, g32, _ = O.train_step_grads(sd, cpu_sample)
with O.bf16_mode():
    torch.manual_seed(3)
    l16, o16, g16, _ = O.train_step_grads(sd, cpu_sample)
print("loss hip16", loss[0].item(), "o16", l16.item(), "o32", l32.item())
rel = lambda a, b: ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()
rows = sorted(((rel(tr.grads[n].cpu(), g16[n]), rel(g16[n], g32[n]), n) for n in g16), reverse=True)
for e, band, n in rows[:10]:
    print(f"{e:.2e}  band {band:.2e}  {n}  |g|max {g16[n].abs().max().item():.2e}")
print("popcount", rel(tr.last["popcount"].cpu(), o16["popcount"]))
