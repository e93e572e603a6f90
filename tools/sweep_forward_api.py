"""Robustness sweep of the drop-in forward API (popcorn.py:100-193): modality x precision x occupancymodel x sentinelbuildings x padding x
sparse x admin_mask x geometry (incl. autograd through the module), every combination against the CPU oracle.  Exceptions are the main
target; fp32 results are held to 1e-4, bf16 results to the looser band of tests/test_gpu_bf16.py."""
import itertools
import os
import sys
import traceback

sys.path.insert(0, os.getcwd())
import torch                                              # noqa: E402
from oracle import popcorn_oracle as O                    # noqa: E402
from popcorn_amd.model import POPCORN                     # noqa: E402

bad = n = 0
shapes = ((2, 100, 100), (1, 37, 53), (2, 96, 160))
for ic, prec, occ, senb, padding, sparse, with_admin, (B, H, W) in itertools.product(
        (6, 2, 4), ("fp32", "bf16"), (True, False), (True, False), (True, False), (True, False), (True, False), shapes):
    if sparse and not with_admin:
        continue                                          # popcorn.py:363: sparse needs admin_mask + census_idx
    n += 1
    tag = f"ic={ic} {prec} occ={int(occ)} senb={int(senb)} pad={int(padding)} sparse={int(sparse)} admin={int(with_admin)} {B}x{H}x{W}"
    try:
        torch.manual_seed(1600)
        m = POPCORN(input_channels=ic, occupancymodel=occ, pretrained=True, biasinit=0.9407, sentinelbuildings=senb).cuda().eval()
        sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        m.set_precision(prec)
        g = torch.Generator().manual_seed(B * 1000 + H)
        x = torch.randn(B, ic, H, W, generator=g)
        inp = {"input": x}
        if with_admin:
            yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
            adm = torch.stack([torch.where((yy + xx) % 7 < 5, float(b + 1), float(b + 9)) for b in range(B)])
            inp["admin_mask"], inp["census_idx"] = adm, torch.arange(1, B + 1)
        if not senb:
            inp["building_counts"] = torch.rand(B, 1, H, W, generator=g)
        with torch.no_grad():
            torch.manual_seed(7)
            if prec == "bf16":
                with O.bf16_mode():
                    ref = O.popcorn_forward(sd, {k: v.clone() for k, v in inp.items()}, padding=padding, sparse=sparse, occupancymodel=occ, sentinelbuildings=senb)
            else:
                ref = O.popcorn_forward(sd, {k: v.clone() for k, v in inp.items()}, padding=padding, sparse=sparse, occupancymodel=occ, sentinelbuildings=senb)
            torch.manual_seed(7)
            out = m({k: v.cuda() for k, v in inp.items()}, padding=padding, sparse=sparse)
        tol = 1e-4 if prec == "fp32" else 3e-2
        worst = 0.0
        for key in ("popdensemap", "popcount"):
            a, r = out[key].float().cpu(), ref[key].float()
            e = ((a - r).abs().max() / r.abs().max().clamp_min(1e-30)).item()
            worst = max(worst, e)
        sc_ok = (out.get("scale") is None) == (ref.get("scale") is None)
        if sc_ok and out.get("scale") is not None:
            sc_ok = tuple(out["scale"].shape) == tuple(ref["scale"].shape)
        ok = worst < tol and sc_ok
        # one autograd pass through the module (train mode)
        m.train()
        torch.manual_seed(7)
        o2 = m({k: v.cuda() for k, v in inp.items()}, train=True, padding=padding, sparse=sparse)
        o2["popcount"].sum().backward()
        gn = sum(float(p.grad.abs().sum()) for p in m.parameters() if p.grad is not None)
        ok = ok and gn == gn and gn > 0
        if not ok:
            bad += 1
            print(tag, f"worst {worst:.2e} scale_ok {sc_ok} gradsum {gn:.3e} BAD", flush=True)
    except Exception as e:
        bad += 1
        print(tag, f"EXCEPTION {type(e).__name__}: {str(e)[:300]}", flush=True)
        traceback.print_exc(limit=4)
print(f"{n} combinations, bad: {bad}")
