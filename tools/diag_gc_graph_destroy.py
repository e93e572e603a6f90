"""Does destroying an older captured step DURING a graph capture abort the process?  (what `train.py: _capturing` guards against)
    python tools/diag_gc_graph_destroy.py guarded|raw
A trainer with a captured graph sits in a reference cycle; its last outside reference is dropped in the middle of the NEXT trainer's
capture, with the collector's thresholds at 1 so that an automatic collection follows immediately."""
import contextlib
import gc
import os
import sys

sys.path.insert(0, os.getcwd())
import torch                                              # noqa: E402
import popcorn_amd.train as T                             # noqa: E402
from popcorn_amd import ops                               # noqa: E402
from popcorn_amd.data import stats                        # noqa: E402
from popcorn_amd.data.synthetic import make_raw_batch     # noqa: E402
from popcorn_amd.model import POPCORN                     # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "guarded"
if mode == "raw":
    @contextlib.contextmanager
    def raw(g, **kw):
        with torch.cuda.graph(g, **kw):
            yield
    T._capturing = raw


def trainer():
    torch.manual_seed(1600)
    m = POPCORN(6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
    return T.FusedTrainStep(m, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01, use_graph=True)


def sample(B, H, W, seed):
    b = make_raw_batch(B, H, W, seed=seed, device="cuda", region="disc")
    return {"input": ops.select_normalize(b["raw"], stats.BAND6, stats.MEAN6, stats.STD6), "admin_mask": b["admin_mask"],
            "census_idx": b["census_idx"], "y": b["y"]}


old = trainer()
old.step(sample(2, 64, 64, 11))
torch.cuda.synchronize()
old._cycle = old                      # the captured graph + its pool now die only through the cyclic collector
holder = [old]
del old
new = trainer()
orig, calls = new._backward, [0]


def backward(*a, **k):
    calls[0] += 1
    if calls[0] == 3:                 # (two warm-up passes, then the captured one)
        holder.clear()
        gc.set_threshold(1, 1, 1)
    return orig(*a, **k)


new._backward = backward
loss = new.step(sample(3, 64, 64, 12))
gc.set_threshold(700, 10, 10)
torch.cuda.synchronize()
print(mode, "ok: loss", loss.tolist(), "collector enabled:", gc.isenabled())
