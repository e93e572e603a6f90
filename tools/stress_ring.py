"""Stress of the head backward's producer / consumer ring (and of the whole replayed step) on SPARSE selections: disc-shaped census
regions leave many 16-pixel groups without a selected pixel, so the producers take their skip paths in irregular patterns.  Thousands of
replayed steps on several batches; the loss must stay finite and the run must finish (a lost wake-up in the ring protocol would hang)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from popcorn_amd.model import POPCORN
from popcorn_amd.train import FusedTrainStep
from popcorn_amd.data.synthetic import make_raw_batch
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
torch.manual_seed(1600)
m = POPCORN(6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
m.set_precision(prec)
tr = FusedTrainStep(m, lr=1e-5, weight_decay=1e-5, gradient_clip=0.01, use_graph=True)
batches = [make_raw_batch(64, 100, 100, seed=200 + i, device="cuda", region="disc" if i % 2 == 0 else "full") for i in range(6)]
st = tr.static_buffers(64, 100, 100, raw_channels=15)
t0 = time.time()
for it in range(nsteps):
    b = batches[it % len(batches)]
    if it % 50 == 0:
        st["raw"].copy_(b["raw"]); st["admin_mask"].copy_(b["admin_mask"]); st["census_idx"].copy_(b["census_idx"]); st["y"].copy_(b["y"])
    loss = tr.step(st)
    if it % 500 == 499:
        v = loss.tolist()
        assert all(x == x and abs(x) < 1e6 for x in v), (it, v)
        print(it + 1, "steps", "%.1f s" % (time.time() - t0), "loss", ["%.4f" % x for x in v], flush=True)
torch.cuda.synchronize()
print("ok", prec, nsteps, "steps in %.1f s" % (time.time() - t0))
