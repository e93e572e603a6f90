"""Host-side cost of one FusedTrainStep.step call (graph replay): wall time to ENQUEUE n steps vs the time the device needs for them."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from popcorn_amd.model import POPCORN
from popcorn_amd.train import FusedTrainStep
from popcorn_amd.data.synthetic import make_raw_batch
torch.manual_seed(1600)
m = POPCORN(6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
tr = FusedTrainStep(m, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01, use_graph=True)
b = make_raw_batch(64, 100, 100, seed=1, device="cuda")
st = tr.static_buffers(64, 100, 100, raw_channels=15)
st["raw"].copy_(b["raw"]); st["admin_mask"].copy_(b["admin_mask"]); st["census_idx"].copy_(b["census_idx"]); st["y"].copy_(b["y"])
for _ in range(20): tr.step(st)
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n): tr.step(st)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.3f ms/step, complete %.3f ms/step" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
t0 = time.perf_counter()
for _ in range(n): tr._draw_selection(100, 100)
print("selection draw alone %.3f ms" % ((time.perf_counter() - t0) / n * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(n): tr.step(st)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
