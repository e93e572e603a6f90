"""Device time of the fused building-score + sparsity-mask launch (pc_building_score_mask) at the bench geometry (B = 64, 100 x 100)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from popcorn_amd import ops
B, H, W = 64, 100, 100
f = torch.randn(B, 2, 128, 128, device="cuda")
w = torch.ones(2, device="cuda"); bias = torch.zeros(1, device="cuda")
admin = torch.arange(1, B + 1, device="cuda").view(B, 1, 1).float().expand(B, H, W).contiguous()
cid = torch.arange(1, B + 1, device="cuda", dtype=torch.int64)
sel = (torch.rand(H + W, device="cuda") < 0.6).to(torch.uint8)
g = torch.cuda.CUDAGraph()
for _ in range(3): ops.building_score_mask(f, w, bias, H, W, 14, 14, admin, cid, sel[:H], sel[H:])
torch.cuda.synchronize()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        for _ in range(50): out = ops.building_score_mask(f, w, bias, H, W, 14, 14, admin, cid, sel[:H], sel[H:])
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print("score_mask: %.2f us per launch (50 launches back to back in a graph), counts %s" % (e0.elapsed_time(e1) * 1e3 / 50, out[2].tolist()))
