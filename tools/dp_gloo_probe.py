"""Two gloo ranks on ONE GPU (functional data-parallel run): per-step wall times of the fused step, to find host-side stalls of the
multi-rank code path that a single process never shows.   python -m torch.distributed.run --nproc-per-node 2 tools/dp_gloo_probe.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("POPCORN_DIST_BACKEND", "gloo")
import torch
from popcorn_amd.distributed import FlatReducer, init_from_env
from popcorn_amd.model import POPCORN
from popcorn_amd.train import FusedTrainStep
from popcorn_amd.data.synthetic import make_raw_batch
rank, local_rank, world = init_from_env()
torch.cuda.set_device(0)
torch.manual_seed(1600)
m = POPCORN(6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
tr = FusedTrainStep(m, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01, reducer=FlatReducer(), use_graph=True)
b = make_raw_batch(int(os.environ.get("PB", "64")), 100, 100, seed=1 + rank, device="cuda")
st = tr.static_buffers(b["raw"].shape[0], 100, 100, raw_channels=15)
st["raw"].copy_(b["raw"]); st["admin_mask"].copy_(b["admin_mask"]); st["census_idx"].copy_(b["census_idx"]); st["y"].copy_(b["y"])
acc = {"stats": [], "grads": []}
def wrap(name, fn):
    def f(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); acc[name].append((time.perf_counter() - t) * 1e3)
        return r
    return f
if os.environ.get("PWRAP"):
    tr.reducer.reduce_stats = wrap("stats", tr.reducer.reduce_stats)
    tr.reducer.reduce_grads = wrap("grads", tr.reducer.reduce_grads)
ts = []
T0 = time.perf_counter()
for i in range(int(os.environ.get("PN", "40"))):
    t0 = time.perf_counter()
    tr.step(st)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
if rank == 0:
    print("per-step ms:", " ".join("%.1f" % t for t in ts[:40]))
    slow = [(i, round(t)) for i, t in enumerate(ts) if t > 20]
    for k, v in acc.items():
        if v: print(k, "all-reduce: median %.2f ms, max %.1f ms, >20 ms: %d of %d" % (sorted(v)[len(v) // 2], max(v), sum(x > 20 for x in v), len(v)))
    print("slow steps (index, ms):", slow[-30:], "of", len(ts), "total wall %.1f s" % (time.perf_counter() - T0))
t0 = time.perf_counter()
for i in range(40):
    tr.step(st)
torch.cuda.synchronize()
if rank == 0:
    print("40 steps without per-step sync: %.2f ms/step" % ((time.perf_counter() - t0) * 25))
