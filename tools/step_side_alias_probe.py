import os, sys, subprocess
code = r'''
import sys, os
sys.path.insert(0, os.getcwd())
import torch
k = int(sys.argv[1])
dummies = [torch.cuda.Stream() for _ in range(k)]
from popcorn_amd import ops
from popcorn_amd.data import stats
from popcorn_amd.data.synthetic import make_raw_batch
from popcorn_amd.model import POPCORN
from popcorn_amd.train import FusedTrainStep
torch.manual_seed(1600)
m = POPCORN(6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
tr = FusedTrainStep(m, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01)
b = make_raw_batch(2, 230, 220, seed=1, device="cuda", region="disc")
x = ops.select_normalize(b["raw"], stats.BAND6, stats.MEAN6, stats.STD6)
smp = {"input": x, "admin_mask": b["admin_mask"], "census_idx": b["census_idx"], "y": b["y"]}
import ctypes
# extra raw HIP streams created through torch's pool do not shift the executor's own hipStreamCreate order: create raw ones too
for _ in range(3): tr.step(dict(smp))
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(40): tr.step(dict(smp))
torch.cuda.synchronize()
print("k=%d  %.3f ms/step" % (k, (time.perf_counter() - t0) / 40 * 1e3))
'''
for k in range(0, 6):
    e = dict(os.environ); e["POPCORN_CONV_DBG"] = "1"
    r = subprocess.run([sys.executable, "-c", code, str(k)], env=e, capture_output=True, text=True)
    line = [l for l in r.stderr.splitlines() if "side stream" in l]
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-200:], "|", line[-1] if line else "")
