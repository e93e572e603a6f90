"""Fused train step at the reference's REAL region geometry (run_train.py:186-202: weak_batch_size 2, variable-size census
regions up to limit1 = 9e6 px, arguments/train.py:16,34-36): per shape -- finite loss, determinism over two runs, ms / step
(eager), peak HBM, and optionally the 56 gradients against the CPU oracle.

    python tools/region_probe.py 2x517x389 2x1030x770 --oracle
    python tools/region_probe.py 2x2100x2150 --regimes 0,1,2
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from popcorn_amd import ops                                   # noqa: E402
from popcorn_amd.data import stats                            # noqa: E402
from popcorn_amd.data.synthetic import make_raw_batch         # noqa: E402
from popcorn_amd.model import POPCORN                         # noqa: E402
from popcorn_amd.train import FusedTrainStep                  # noqa: E402

REGIMES = {0: (False, False), 1: (True, False), 2: (True, True)}      # none | limit1 (encoder frozen) | limit2 (U-Net frozen)


def fresh(precision="fp32"):
    torch.manual_seed(1600)
    m = POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
    m.set_precision(precision)
    return FusedTrainStep(m, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shapes", nargs="+")
    ap.add_argument("--regimes", default="0")
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--fp64", action="store_true", help="with --oracle: also the fp64 oracle (which fp32 side is the neighbour of the exact gradients?)")
    ap.add_argument("--region", default="disc")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    res = []
    for shp in a.shapes:
        B, H, W = (int(v) for v in shp.split("x"))
        batch = make_raw_batch(B, H, W, seed=H * 1000 + W, region=a.region)
        x = ops.select_normalize(batch["raw"].cuda(), stats.BAND6, stats.MEAN6, stats.STD6)
        del batch["raw"]
        dev = {"input": x, "admin_mask": batch["admin_mask"].cuda(), "census_idx": batch["census_idx"].cuda(), "y": batch["y"].cuda()}
        for r in (int(v) for v in a.regimes.split(",")):
            enc_ng, unet_ng = REGIMES[r]
            rec = {"shape": shp, "regime": r, "px": B * H * W}
            runs = []
            for rep in range(2):
                tr = fresh(a.precision)
                torch.cuda.reset_peak_memory_stats()
                torch.manual_seed(3)
                loss = tr.step(dict(dev), encoder_no_grad=enc_ng, unet_no_grad=unet_ng)
                torch.cuda.synchronize()
                runs.append((loss.tolist(), tr.flat_g.clone(), tr.flat_p.clone()))
                rec["peak_gib"] = torch.cuda.max_memory_allocated() / 2 ** 30
            rec["loss"] = runs[0][0]
            rec["finite"] = bool(torch.isfinite(runs[0][1]).all() and torch.isfinite(runs[0][2]).all())
            rec["deterministic"] = bool(torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2]) and runs[0][0] == runs[1][0])
            t0 = time.time()
            for _ in range(a.steps):
                tr.step(dict(dev), encoder_no_grad=enc_ng, unet_no_grad=unet_ng)
            torch.cuda.synchronize()
            rec["ms_per_step"] = (time.time() - t0) / a.steps * 1e3
            rec["mpx_per_s"] = B * H * W / rec["ms_per_step"] / 1e3
            if a.oracle:
                from oracle import popcorn_oracle as O
                sd = {k: v.detach().cpu().clone() for k, v in fresh(a.precision).model.state_dict().items()}
                cpu = {k: v.cpu() for k, v in dev.items()}
                torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
                t0 = time.time()
                torch.manual_seed(3)
                ref_loss, _, ref_grads, _ = O.train_step_grads(sd, cpu, encoder_no_grad=enc_ng, unet_no_grad=unet_ng)
                rec["oracle_s"] = time.time() - t0
                tr = fresh(a.precision)
                torch.manual_seed(3)
                loss = tr.step(dict(dev), encoder_no_grad=enc_ng, unet_no_grad=unet_ng)
                rec["loss_rel"] = abs(loss[0].item() - ref_loss.item()) / max(1.0, abs(ref_loss.item()))
                errs = {n: ((tr.grads[n].cpu() - g).abs().max() / max(g.abs().max().item(), 1e-3)).item() for n, g in ref_grads.items()}
                rec["grad_worst"] = max(errs.values())
                rec["grad_worst_name"] = max(errs, key=errs.get)
                rec["n_grads"] = len(errs)
                if a.fp64:
                    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
                    cpu64 = {k: (v.double() if v.is_floating_point() else v) for k, v in cpu.items()}
                    torch.manual_seed(3)
                    _, _, g64, _ = O.train_step_grads(sd64, cpu64, encoder_no_grad=enc_ng, unet_no_grad=unet_ng)
                    rl = lambda x, r: ((x.double() - r).abs().max() / max(r.abs().max().item(), 1e-3)).item()  # noqa: E731
                    eh = {n: rl(tr.grads[n].cpu(), g64[n]) for n in g64}
                    er = {n: rl(ref_grads[n], g64[n]) for n in g64}
                    rec["hip_vs_fp64"] = max(eh.values())
                    rec["hip_vs_fp64_name"] = max(eh, key=eh.get)
                    rec["ref32_vs_fp64"] = max(er.values())
                    rec["ref32_vs_fp64_name"] = max(er, key=er.get)
            print(json.dumps(rec), flush=True)
            res.append(rec)
            del tr, runs
            torch.cuda.empty_cache()
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
