"""fp32 vs bf16 TRAINING on the same data from the same seed (the che recipe, reference README.md:182,
stands behind BASELINE config 4).  A teacher (the same architecture with another head initialisation) labels synthetic
census regions, so that there is something to learn; two students start from identical parameters and see identical batches
and selection grids, one in fp32 and one in PC_PREC_BF16.  Shared by tests/test_gpu_bf16.py (assertion) and
tools/bf16_training_quality.py (the table in DESIGN_HISTORY.md section 7)."""
import torch


def run(steps=200, B=16, nbatches=8, lr=5e-4, wd=5e-7, use_graph=True):
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
        s["y"] = y * 1.5 + 3.0                  # the students have to move: scaled + shifted teacher counts
        batches.append(s)
    out = {}
    p0 = None
    for prec in ("fp32", "bf16"):
        m = model(1600)
        m.set_precision(prec)
        tr = FusedTrainStep(m, lr=lr, weight_decay=wd, gradient_clip=0.01, use_graph=use_graph)
        if p0 is None:
            p0 = tr.flat_p.clone()
        losses, preds, ys = [], [], []
        for it in range(steps):
            s = batches[it % nbatches]
            torch.manual_seed(1000 + it)        # identical selection grids for both precisions
            l = tr.step(dict(s))
            losses.append(l[0].item())
            preds.append(tr.last["popcount"].detach().float().clone())
            ys.append(s["y"])
        torch.cuda.synchronize()
        r2 = []
        for it in range(nbatches - 1, steps, nbatches):        # R^2 over the last `nbatches` batches (one pass over the set)
            p = torch.cat(preds[it - nbatches + 1:it + 1])
            y = torch.cat(ys[it - nbatches + 1:it + 1])
            r2.append(1.0 - float(((p - y) ** 2).sum() / ((y - y.mean()) ** 2).sum()))
        out[prec] = {"loss": losses, "r2": r2, "params": tr.flat_p.clone()}
    d = (out["bf16"]["params"] - out["fp32"]["params"]).norm().item()
    moved = (out["fp32"]["params"] - p0).norm().item()
    k = nbatches
    res = {"steps": steps, "batch": B, "lr": lr,
           "loss_first_epoch": {p: sum(out[p]["loss"][:k]) / k for p in out},
           "loss_last_epoch": {p: sum(out[p]["loss"][-k:]) / k for p in out},
           "loss_last_5_epochs": {p: sum(out[p]["loss"][-5 * k:]) / (5 * k) for p in out},
           "r2_first_epoch": {p: out[p]["r2"][0] for p in out}, "r2_last_epoch": {p: out[p]["r2"][-1] for p in out},
           "r2_last_5_epochs": {p: sum(out[p]["r2"][-5:]) / 5 for p in out},
           "r2_median_last_10_epochs": {p: sorted(out[p]["r2"][-10:])[5] for p in out},
           "loss_median_last_10_epochs": {p: sorted(sum(out[p]["loss"][i:i + k]) / k for i in range(steps - 10 * k, steps, k))[5] for p in out},
           "r2_trajectory": {p: [round(v, 4) for v in out[p]["r2"]] for p in out},
           "loss_trajectory_epoch_means": {p: [round(sum(out[p]["loss"][i:i + k]) / k, 5) for i in range(0, steps - k + 1, k)] for p in out},
           "param_distance_bf16_vs_fp32": d, "param_distance_fp32_moved": moved, "relative_param_distance": d / max(moved, 1e-12)}
    return res
