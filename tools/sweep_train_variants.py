"""Robustness sweep of the fused train step: loss recipe x occupancymodel x sentinelbuildings (own building layer) x regime x precision x
graph, against the CPU oracle's loss and gradients (fp32: 2e-4 unless the mismatch is of the tie class ~1e-3, reported; bf16: loss only).
Exceptions and non-finite values are the main target."""
import itertools
import os
import sys
import traceback

sys.path.insert(0, os.getcwd())
import torch                                              # noqa: E402
from oracle import popcorn_oracle as O                    # noqa: E402
from popcorn_amd.data.synthetic import make_raw_batch     # noqa: E402
from popcorn_amd.model import POPCORN                     # noqa: E402
from popcorn_amd.train import FusedTrainStep              # noqa: E402

LOSSES = [(("log_l1_loss",), (1.0,)), (("l1_loss",), (1.0,)), (("mse_loss",), (1e-3,)), (("log_mse_loss", "l1_loss"), (1.0, 0.01))]
REG = {0: {}, 1: dict(encoder_no_grad=True), 2: dict(encoder_no_grad=True, unet_no_grad=True)}
bad = n = ties = 0
for (loss, lam), occ, senb, sreg, reg, prec, graph in itertools.product(LOSSES, (True, False), (True, False), (0.01, 0.0), (0, 1, 2),
                                                                         ("fp32", "bf16"), (False, True)):
    n += 1
    tag = f"{'+'.join(loss)} occ={int(occ)} senb={int(senb)} sreg={sreg} regime={reg} {prec} graph={int(graph)}"
    try:
        torch.manual_seed(1600)
        m = POPCORN(input_channels=6, occupancymodel=occ, pretrained=True, biasinit=0.9407, sentinelbuildings=senb).cuda()
        sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        m.set_precision(prec)
        tr = FusedTrainStep(m, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01, loss=loss, lam=lam, scale_regularization=sreg, use_graph=graph)
        b = make_raw_batch(3, 100, 100, seed=3, region="disc")
        cpu = {"input": O.select_normalize(b["raw"]), "admin_mask": b["admin_mask"], "census_idx": b["census_idx"], "y": b["y"]}
        if not senb:
            cpu["building_counts"] = torch.rand(3, 1, 100, 100, generator=torch.Generator().manual_seed(4))
        torch.manual_seed(5)
        l = tr.step({k: v.cuda() for k, v in cpu.items()}, **REG[reg])
        torch.cuda.synchronize()
        torch.manual_seed(5)
        kw = dict(loss=loss, lam=lam, scale_regularization=sreg, **REG[reg])
        if prec == "bf16":
            with O.bf16_mode():
                rl, _, rg, _ = _ = O.train_step_grads(sd, {k: v.clone() for k, v in cpu.items()}, **kw) if occ and senb else (None, None, None, None)
        else:
            work = {k: v.clone() for k, v in cpu.items()}
            names = O.trainable_names(sd)
            wsd = dict(sd)
            for nm in names:
                wsd[nm] = sd[nm].detach().clone().requires_grad_(True)
            out = O.popcorn_forward(wsd, work, padding=False, sparse=True, occupancymodel=occ, sentinelbuildings=senb, **REG[reg])
            rl, _ = O.get_loss(out, work, scale=out["scale"], loss=loss, lam=lam, scale_regularization=sreg, tag="weak")
            (rl * 100.0).backward()
            rg = {nm: wsd[nm].grad for nm in names if wsd[nm].grad is not None}
            rl = rl.detach()
        ok = bool(torch.isfinite(tr.flat_p).all()) and l[0].item() == l[0].item()
        note = ""
        if rl is not None:
            le = abs(l[0].item() - rl.item()) / max(1.0, abs(rl.item()))
            ok = ok and le < (1e-5 if prec == "fp32" else 2e-3)
            if prec == "fp32":
                worst = max(((tr.grads[nm].cpu() - g).abs().max() / max(g.abs().max().item(), 1e-3)).item() for nm, g in rg.items())
                if worst >= 2e-4:
                    if worst < 5e-3:
                        ties += 1
                        note = f" (tie-class gradient mismatch {worst:.1e})"
                    else:
                        ok = False
                        note = f" GRAD {worst:.1e}"
                for nm in tr.names:
                    if nm not in rg and float(tr.grads[nm].abs().max()) != 0.0:
                        ok, note = False, note + f" nonzero grad on frozen {nm}"
        if not ok or note:
            print(tag, "ok" if ok else "BAD", note, flush=True)
        bad += 0 if ok else 1
    except Exception as e:
        bad += 1
        print(tag, f"EXCEPTION {type(e).__name__}: {str(e)[:300]}", flush=True)
        traceback.print_exc(limit=5)
print(f"{n} combinations, bad: {bad}, tie-class mismatches: {ties}")
