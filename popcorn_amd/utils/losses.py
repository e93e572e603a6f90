"""Loss / metric glue with the reference's surface (utils/losses.py:12-127): ``get_loss``, ``mape_func``, ``r2``.

These are O(batch) scalar reductions on B-element vectors (plus one mean over the selected-pixel scale vector);
they are host-side glue in torch ops, device-agnostic.  The fused training path (popcorn_amd/train.py) computes the
same optimisation loss and its gradient in one HIP kernel instead (pc_loss_fwd_bwd) and never calls ``.item()``.
"""
import torch


def mape_func(pred, gt, eps=1e-8):
    """utils/losses.py:91-97."""
    pos_mask = gt > 0.1
    return ((pred[pos_mask] - gt[pos_mask]).abs() / (gt[pos_mask] + eps)).mean() * 100


def r2(pred, gt, eps=1e-8):
    """R2 = 1 - SS_res / SS_tot.  utils/losses.py:101-127."""
    gt_mean = torch.mean(gt)
    ss_tot = torch.sum((gt - gt_mean) ** 2)
    ss_res = torch.sum((gt - pred) ** 2)
    return 1 - ss_res / (ss_tot + eps)


def _population_terms(pred, y):
    """The per-batch population metrics of get_loss (utils/losses.py:49-60), keyed as the reference logs them."""
    lp, ly = torch.log(pred + 1), torch.log(y + 1)
    several = len(pred) > 1                      # R2 / Pearson need at least two samples (losses.py:57,59)
    zero = torch.tensor(0.0)
    return {
        "l1_loss": (pred - y).abs().mean(),
        "log_l1_loss": (lp - ly).abs().mean(),
        "mse_loss": (pred - y).square().mean(),
        "log_mse_loss": (lp - ly).square().mean(),
        "mr2": r2(pred, y) if several else zero,
        "mape": mape_func(pred, y),
        "mCorrelation": torch.corrcoef(torch.stack((pred, y)))[0, 1] if several else zero,
    }


def get_loss(output, gt, scale=None, loss=["l1_loss"], lam=[1.0], tag="", scale_regularization=0.0):
    """utils/losses.py:12-88: weighted sum of the selected population losses + ``scale_regularization * mean|scale|``;
    returns (optimisation loss tensor, {"Population[_tag]/<metric>": float, "optimization_loss": float}).  The float
    conversion synchronises the device, exactly like the reference (the fused step avoids it)."""
    for key in ("popcount", "popdensemap", "scale"):
        t = output.get(key)
        if t is not None and t.dtype != torch.float32:
            output[key] = t.float()
    pred = output["popcount"]
    terms = _population_terms(pred, gt["y"])
    total = torch.zeros((), device=pred.device, dtype=pred.dtype)
    for name, weight in zip(loss, lam):
        if name in terms:
            total = total + weight * terms[name]
    if scale is not None:
        for bad, what in ((torch.isnan, "nan"), (torch.isinf, "inf")):
            if bad(scale).any():
                raise ValueError(f"{what} values detected in scale.")
        terms["scale"] = scale.float().abs().mean()
        if scale_regularization > 0.0:
            total = total + scale_regularization * terms["scale"]
    prefix = "Population/" if tag == "" else f"Population_{tag}/"
    log = {prefix + name: value.detach().item() for name, value in terms.items()}
    log["optimization_loss"] = total.detach().item()
    return total, log
