"""fp32 TRAINING in the two multiplication forms of the conv kernels (round 6): split-operand kernels on the bf16 matrix pipe (forward convs
csrc/conv3x3_fwd_s3.h, fused backward conv3x3_bwd_s3_kernel; pc_set_conv_split 1) against the v_mfma_f32_16x16x4_f32 kernels
(pc_set_conv_split 0), same data, same seeds, same selection grids: the setting of tests/bf16_quality.py (a teacher labels synthetic census
regions, 200 steps, B = 16).  Both are fp32 arithmetic with different rounding, so the trajectories separate only as fast as training
amplifies rounding (decision flips at ReLU / arg-max ties); reported: per-step loss distance, distance of the parameters after N steps
relative to how far training moved them, R^2 of both students.

    python3 tools/split_form_training.py [steps] > gpurun_out/r6_split_form_training.json
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(steps=200, B=16, nbatches=8, lr=5e-4, wd=5e-7):
    from popcorn_amd import _lib as L
    from popcorn_amd import ops
    from popcorn_amd.data import stats
    from popcorn_amd.data.synthetic import make_raw_batch
    from popcorn_amd.model import POPCORN
    from popcorn_amd.train import FusedTrainStep

    def model(seed):
        torch.manual_seed(seed)
        return POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.2267, sentinelbuildings=True).cuda()

    teacher = model(77)
    batches = []
    for i in range(nbatches):
        b = make_raw_batch(B, 100, 100, seed=4000 + i, device="cuda", region="disc")
        x = ops.select_normalize(b["raw"], stats.BAND6, stats.MEAN6, stats.STD6)
        s = {"input": x, "admin_mask": b["admin_mask"], "census_idx": b["census_idx"]}
        with torch.no_grad():
            torch.manual_seed(5)
            y = teacher(dict(s), train=False, padding=False, sparse=True)["popcount"].detach().clone()
        s["y"] = y * 1.5 + 3.0
        batches.append(s)
    out, p0 = {}, None
    # control: a THIRD fp32 evaluation that differs from "fp32mfma" by rounding elsewhere (the head's products in fp32-MFMA form instead of the
    # split form, pc_set_head_split 0): how far two runs drift apart through rounding alone, whatever its source
    for form, hform, name in ((1, 1, "split"), (0, 1, "fp32mfma"), (0, 0, "control_fp32mfma_head_too")):
        prev = L.lib().pc_set_conv_split(form)
        hprev = L.lib().pc_set_head_split(hform)
        try:
            m = model(1600)
            tr = FusedTrainStep(m, lr=lr, weight_decay=wd, gradient_clip=0.01, use_graph=True)
            if p0 is None:
                p0 = tr.flat_p.clone()
            losses, preds, ys = [], [], []
            for it in range(steps):
                s = batches[it % nbatches]
                torch.manual_seed(1000 + it)
                l = tr.step(dict(s))
                losses.append(l[0].item())
                preds.append(tr.last["popcount"].detach().float().clone())
                ys.append(s["y"])
            torch.cuda.synchronize()
        finally:
            L.lib().pc_set_conv_split(prev)
            L.lib().pc_set_head_split(hprev)
        r2 = []
        for it in range(nbatches - 1, steps, nbatches):
            p = torch.cat(preds[it - nbatches + 1:it + 1])
            y = torch.cat(ys[it - nbatches + 1:it + 1])
            r2.append(1.0 - float(((p - y) ** 2).sum() / ((y - y.mean()) ** 2).sum()))
        out[name] = {"loss": losses, "r2": r2, "params": tr.flat_p.clone()}
    a, b, c = out["split"], out["fp32mfma"], out["control_fp32mfma_head_too"]
    relc = [abs(x - y) / max(abs(y), 1e-12) for x, y in zip(c["loss"], b["loss"])]
    dc = (c["params"] - b["params"]).norm().item()
    rel = [abs(x - y) / max(abs(y), 1e-12) for x, y in zip(a["loss"], b["loss"])]
    moved = (b["params"] - p0).norm().item()
    d = (a["params"] - b["params"]).norm().item()
    k = nbatches
    return {"steps": steps, "batch": B, "lr": lr,
            "loss_rel_distance_step_1": rel[0], "loss_rel_distance_max_first_10_steps": max(rel[:10]),
            "loss_rel_distance_max_all_steps": max(rel), "loss_rel_distance_median": sorted(rel)[len(rel) // 2],
            "loss_first_epoch": {n: sum(out[n]["loss"][:k]) / k for n in out}, "loss_last_epoch": {n: sum(out[n]["loss"][-k:]) / k for n in out},
            "r2_last_epoch": {n: out[n]["r2"][-1] for n in out}, "r2_last_5_epochs": {n: sum(out[n]["r2"][-5:]) / 5 for n in out},
            "param_distance_split_vs_fp32mfma": d, "param_distance_moved_by_training": moved, "relative_param_distance": d / max(moved, 1e-12),
            "control": {"what": "fp32mfma convs with the head in fp32-MFMA form vs fp32mfma convs with the head in split form (rounding elsewhere)",
                        "loss_rel_distance_step_1": relc[0], "loss_rel_distance_max_first_10_steps": max(relc[:10]),
                        "loss_rel_distance_median": sorted(relc)[len(relc) // 2], "param_distance": dc,
                        "relative_param_distance": dc / max(moved, 1e-12)},
            "note": "two fp32 evaluations of the same training run; for scale: tests/bf16_quality.py reports the same quantities for bf16 vs fp32"}


if __name__ == "__main__":
    print(json.dumps(run(steps=int(sys.argv[1]) if len(sys.argv) > 1 else 200), indent=1))
