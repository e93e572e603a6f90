"""Ablation of the producer/consumer head-backward kernel (B=64, 100x100, all pixels selected).  GPU only.
POPCORN_HEAD_DBG bits: 1 consumer idle, 2 no hand-off (producer never touches the ring)."""
import os, subprocess, sys
code = r'''
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from popcorn_amd import ops, _lib as L
from popcorn_amd.model import POPCORN
torch.manual_seed(0)
m = POPCORN(6, occupancymodel=True, pretrained=True, biasinit=0.9, sentinelbuildings=True).cuda()
B, H, W = 64, 100, 100
feats = torch.randn(B, 16, 128, 128, device="cuda"); building = torch.rand(B, 1, H, W, device="cuda")
admin = torch.ones(B, H, W, device="cuda"); census = torch.ones(B, dtype=torch.int64, device="cuda")
gpc = torch.ones(B, device="cuda"); gsc = torch.full((1,), 1e-3, device="cuda")
grads = [torch.empty_like(t) for t in m.head_tensors()]; gf = torch.empty(B, 16, 128, 128, device="cuda")
eng = m.engines()[0]
def run():
    ops.head_bwd(feats, 14, 14, H, W, m.head_tensors(), building, admin_mask=admin, census_idx=census, g_popcount=gpc,
                 g_scale_const=gsc, grads=grads, g_feat=gf, feat_bn=eng.feat_bn())
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print("%.1f us" % (e0.elapsed_time(e1) * 100))
'''
for tag, env in [("pc full", {}), ("single-role", {"POPCORN_HEAD_BWD_SINGLE_ROLE": "1"}), ("consumer idle", {"POPCORN_HEAD_DBG": "1"}),
                 ("no hand-off", {"POPCORN_HEAD_DBG": "2"})]:
    e = dict(os.environ); e.update(env)
    out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
    if "--phases" in sys.argv:       # profiling build (tools/head_phases.sh): the last launch's producer phase cycles
        ph = [l for l in out.stderr.splitlines() if "producer phases" in l]
        print(f"{tag:24s}", ph[-1].split("):")[-1] if ph else out.stderr[-300:])
        continue
    print(f"{tag:24s}", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
