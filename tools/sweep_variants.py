import os, sys, itertools, traceback
sys.path.insert(0, os.getcwd())
import torch
from oracle import popcorn_oracle as O
from popcorn_amd.model import POPCORN
from popcorn_amd.train import FusedTrainStep
from popcorn_amd.data.synthetic import make_raw_batch
bad = 0
for ic, prec, (B, H, W), graph in itertools.product((6, 2, 4), ("fp32", "bf16"), ((2, 100, 100), (1, 64, 48), (2, 333, 201), (3, 128, 128)), (False, True)):
    try:
        torch.manual_seed(1600)
        m = POPCORN(input_channels=ic, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
        m.set_precision(prec)
        tr = FusedTrainStep(m, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01, use_graph=graph)
        b = make_raw_batch(B, H, W, seed=3, region="disc")
        x6 = O.select_normalize(b["raw"])
        x = {6: x6, 2: x6[:, 4:6], 4: x6[:, 0:4]}[ic].contiguous().cuda()
        smp = {"input": x, "admin_mask": b["admin_mask"].cuda(), "census_idx": b["census_idx"].cuda(), "y": b["y"].cuda()}
        ls = []
        for i in range(3):
            torch.manual_seed(5 + i)
            ls.append(tr.step(dict(smp))[0].item())
        torch.cuda.synchronize()
        ok = all(l == l and abs(l) < 1e6 for l in ls) and bool(torch.isfinite(tr.flat_p).all())
        # eager and graph must agree bit for bit: remember eager result
        key = (ic, prec, B, H, W)
        if not graph:
            ref = {key: (ls, tr.flat_p.clone())}
            globals().setdefault("REF", {}).update(ref)
        else:
            l0, p0 = REF[key]
            ok = ok and l0 == ls and torch.equal(p0, tr.flat_p)
        if not ok:
            bad += 1
        print(f"ic={ic} {prec} {B}x{H}x{W} graph={graph}: losses {['%.5f' % l for l in ls]} {'ok' if ok else 'BAD'}", flush=True)
    except Exception as e:
        bad += 1
        print(f"ic={ic} {prec} {B}x{H}x{W} graph={graph}: EXCEPTION {type(e).__name__}: {e}", flush=True)
        traceback.print_exc()
print("bad cases:", bad)
