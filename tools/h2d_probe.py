"""The double-buffered host feed of bench.py's h2d leg alone (6-band tiles), for `rocprofv3 --kernel-trace --stats`: do the H2D copies run
as shader blit kernels (they then show up as __amd_rocclr_copyBuffer with long durations and compete with the step's kernels for CUs)?"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from popcorn_amd.model import POPCORN
from popcorn_amd.train import FusedTrainStep
from popcorn_amd.data import stats
from popcorn_amd.data.synthetic import make_raw_batch
nband = int(sys.argv[1]) if len(sys.argv) > 1 else 6
torch.manual_seed(1600)
m = POPCORN(6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
tr = FusedTrainStep(m, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01, use_graph=True)
b = make_raw_batch(64, 100, 100, seed=1, device="cuda")
raw = b["raw"][:, list(stats.BAND6)].contiguous() if nband == 6 else b["raw"]
if nband == 6:
    tr.raw_norm = (tuple(range(6)), stats.MEAN6, stats.STD6)
host = {"raw": raw.cpu().pin_memory(), "_packed": tr.pack_small(b["admin_mask"].cpu(), b["y"].cpu(), b["census_idx"].cpu()).pin_memory()}
sets = [tr.static_buffers(64, 100, 100, raw_channels=raw.shape[1], slot=s) for s in (0, 1)]
copied = [torch.cuda.Event() for _ in range(2)]; consumed = [torch.cuda.Event() for _ in range(2)]
cs = torch.cuda.Stream(); cur = torch.cuda.current_stream()
for e in consumed: e.record(cur)
COPY = True
NSPLIT = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # raw tile split over this many copy streams (one SDMA engine each)
cs2 = [torch.cuda.Stream() for _ in range(NSPLIT - 1)]
ev2 = [[torch.cuda.Event() for _ in range(NSPLIT - 1)] for _ in range(2)]
def feed(i):
    s = i & 1
    B = host["raw"].shape[0]
    with torch.cuda.stream(cs):
        cs.wait_event(consumed[s])
    if COPY:
        for j, st_ in enumerate(cs2):
            with torch.cuda.stream(st_):
                st_.wait_event(consumed[s])
                lo, hi = B * (j + 1) // NSPLIT, B * (j + 2) // NSPLIT
                sets[s]["raw"][lo:hi].copy_(host["raw"][lo:hi], non_blocking=True)
                ev2[s][j].record(st_)
    with torch.cuda.stream(cs):
        if COPY:
            sets[s]["raw"][:B // NSPLIT].copy_(host["raw"][:B // NSPLIT], non_blocking=True)
            sets[s]["_packed"].copy_(host["_packed"], non_blocking=True)
            for e in ev2[s]: cs.wait_event(e)
        copied[s].record(cs)
def run(n, do_feed=True):
    if do_feed: feed(0)
    for i in range(n):
        if do_feed and i + 1 < n: feed(i + 1)
        s = i & 1
        if do_feed: cur.wait_event(copied[s])
        tr.step(sets[s])
        if do_feed: consumed[s].record(cur)
for mode, cp in ((True, True), (False, True), (True, False), (True, True), (False, True), (True, False)):
    COPY = cp
    run(10, mode); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(200, mode); torch.cuda.synchronize()
    print(("feed" if cp else "events only, no copies") if mode else "resident (same two graphs, no events)", "%.4f ms/step" % ((time.perf_counter() - t0) * 5))
