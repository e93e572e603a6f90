"""Host-side cost of an EAGER FusedTrainStep.step on a small census-region batch (the reference's variable-size regions cannot replay a
captured graph): wall time per step vs device time, and the cProfile top of the enqueue path.

    python tools/host_time_eager.py [B H W]
"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.getcwd())
import torch                                                         # noqa: E402
from popcorn_amd import ops                                          # noqa: E402
from popcorn_amd.data import stats                                   # noqa: E402
from popcorn_amd.data.synthetic import make_raw_batch                # noqa: E402
from popcorn_amd.model import POPCORN                                # noqa: E402
from popcorn_amd.train import FusedTrainStep                         # noqa: E402

B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2, 230, 220)
torch.manual_seed(1600)
m = POPCORN(6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
tr = FusedTrainStep(m, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01)
b = make_raw_batch(B, H, W, seed=1, device="cuda", region="disc")
x = ops.select_normalize(b["raw"], stats.BAND6, stats.MEAN6, stats.STD6)
smp = {"input": x, "admin_mask": b["admin_mask"], "census_idx": b["census_idx"], "y": b["y"]}
for _ in range(5):
    tr.step(dict(smp))
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n):
    tr.step(dict(smp))
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
print("eager %dx%dx%d: enqueue %.3f ms/step, complete %.3f ms/step" % (B, H, W, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
# the trainer bounds the host's run-ahead to 8 steps (FusedTrainStep._throttle): over 50 steps the figure above is mostly the DEVICE's
# pace.  The host's own cost per step = 7 steps enqueued into an idle queue (inside the run-ahead window), best of 5
best = None
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(7):
        tr.step(dict(smp))
    dt = (time.perf_counter() - t0) / 7
    best = dt if best is None or dt < best else best
torch.cuda.synchronize()
print("eager %dx%dx%d: host enqueue alone (inside the run-ahead window) %.3f ms/step" % (B, H, W, best * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    tr.step(dict(smp))
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
# ---- where the host's time goes: the pieces of one native step, each alone (7 calls into an idle queue, best of 5)
import ctypes as C                                                   # noqa: E402
from popcorn_amd import _lib as L                                    # noqa: E402


def best_of(fn, reps=5, n=7):
    b_ = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        d = (time.perf_counter() - t0) / n
        b_ = d if b_ is None or d < b_ else b_
    torch.cuda.synchronize()
    return b_ * 1e3


if tr.native_steps:
    s = {k: v.contiguous() for k, v in smp.items()}
    s["admin_mask"] = s["admin_mask"].float()
    sel = tr._draw_selection(H, W)
    with L.precision(m.precision), L.stream_scope():
        h = tr._native_handle()
        io = tr._native_io(s, sel, False, False)
        stream = L.stream_ptr()
        io.arena, io.arena_bytes = tr._arena.data_ptr(), tr._arena.numel()
        lib = L.lib()
        full = L.PC_STEP_FWD | L.PC_STEP_BWD | L.PC_STEP_UPD
        print("pieces (ms/step): draw_selection %.3f  sample dict %.3f  _native_io %.3f  pc_train_step %.3f (fwd %.3f, bwd %.3f, upd %.3f)  phases=0 %.3f" % (
            best_of(lambda: tr._draw_selection(H, W)),
            best_of(lambda: {k: (v.contiguous() if torch.is_tensor(v) else v) for k, v in smp.items()}["admin_mask"].float()),
            best_of(lambda: tr._native_io(s, sel, False, False)),
            best_of(lambda: lib.pc_train_step(h, C.byref(io), full, stream)),
            best_of(lambda: lib.pc_train_step(h, C.byref(io), L.PC_STEP_FWD, stream)),
            best_of(lambda: lib.pc_train_step(h, C.byref(io), L.PC_STEP_BWD, stream)),
            best_of(lambda: lib.pc_train_step(h, C.byref(io), L.PC_STEP_UPD, stream)),
            best_of(lambda: lib.pc_train_step(h, C.byref(io), 0, stream))))
        ns = (C.c_double * 2)()
        lib.pc_debug_step_host_ns.argtypes = [C.POINTER(C.c_double)]
        acc = [0.0, 0.0]
        torch.cuda.synchronize()
        for _ in range(7):
            lib.pc_train_step(h, C.byref(io), full, stream)
            lib.pc_debug_step_host_ns(ns)
            acc[0] += ns[0]
            acc[1] += ns[1]
        torch.cuda.synchronize()
        print("inside pc_train_step: sizing pass %.1f us, launching pass %.1f us, %d launches + event calls" % (acc[0] / 7e3, acc[1] / 7e3, io.launches))
