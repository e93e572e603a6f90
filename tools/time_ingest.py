import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from popcorn_amd import ops
from popcorn_amd.data import stats
from popcorn_amd.data.synthetic import make_raw_batch
b = make_raw_batch(64, 100, 100, seed=1, device="cuda")
b6 = list(stats.BAND6)
s2 = b["raw"][:, b6[:4]].to(torch.int32).cpu().to(torch.uint16).cuda().contiguous()
s1 = b["raw"][:, b6[4:]].contiguous()
raw6 = b["raw"][:, b6].contiguous()
order = [4, 5, 2, 1, 0, 3]
mean = [stats.MEAN6[c] for c in order]; std = [stats.STD6[c] for c in order]
def t(fn, n=50):
    """device time per launch: n launches back to back inside one replayed graph (eager calls are host-bound at ~12 us)"""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("select_normalize_pad 15-band", t(lambda: ops.select_normalize_pad(b["raw"], [stats.BAND6[c] for c in order], mean, std, 14, 14, 14, 14)))
print("select_normalize_pad 6-band", t(lambda: ops.select_normalize_pad(raw6, order, mean, std, 14, 14, 14, 14)))
print("ingest_split fp32 out", t(lambda: ops.ingest_split(s2, s1, order, mean, std, 14, 14, 14, 14, cl8=False)))
print("ingest_split cl8 out", t(lambda: ops.ingest_split(s2, s1, order, mean, std, 14, 14, 14, 14, cl8=True)))
print("ingest_cl8 15-band", t(lambda: ops.ingest_cl8(b["raw"], [stats.BAND6[c] for c in order], mean, std, 14, 14, 14, 14)))
