"""Loss / metric glue with the reference's surface (utils/losses.py:12-127): ``get_loss``, ``mape_func``, ``r2``.

These are O(batch) scalar reductions on B-element vectors (plus one mean over the selected-pixel scale vector);
they are host-side glue in torch ops, device-agnostic.  The fused training path (popcorn_amd/train.py) computes the
same optimisation loss and its gradient in one HIP kernel instead (pc_loss_fwd_bwd) and never calls ``.item()``.
"""
from collections import defaultdict

import torch
import torch.nn.functional as F


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


def get_loss(output, gt, scale=None, loss=["l1_loss"], lam=[1.0], tag="", scale_regularization=0.0):
    """utils/losses.py:12-88: weighted sum of population losses + scale regularisation; auxdict of floats
    (``.item()`` -> device sync, exactly like the reference)."""
    auxdict = defaultdict(float)
    for k in ("popcount", "popdensemap", "scale"):
        if output.get(k) is not None and output[k].dtype != torch.float32:
            output[k] = output[k].float()
    y_pred, y_gt = output["popcount"], gt["y"]
    metricdict = {
        "l1_loss": F.l1_loss(y_pred, y_gt),
        "log_l1_loss": F.l1_loss(torch.log(y_pred + 1), torch.log(y_gt + 1)),
        "mse_loss": F.mse_loss(y_pred, y_gt),
        "log_mse_loss": F.mse_loss(torch.log(y_pred + 1), torch.log(y_gt + 1)),
        "mr2": r2(y_pred, y_gt) if len(y_pred) > 1 else torch.tensor(0.0),
        "mape": mape_func(y_pred, y_gt),
        "mCorrelation": torch.corrcoef(torch.stack([y_pred, y_gt]))[0, 1] if len(y_pred) > 1 else torch.tensor(0.0),
    }
    optimization_loss = torch.tensor(0, device=y_pred.device, dtype=y_pred.dtype)
    for lo, la in zip(loss, lam):
        if lo in metricdict:
            optimization_loss = optimization_loss + metricdict[lo] * la
    if scale is not None:
        if torch.isnan(scale).any():
            raise ValueError("nan values detected in scale.")
        if torch.isinf(scale).any():
            raise ValueError("inf values detected in scale.")
        metricdict["scale"] = scale.float().abs().mean()
        if scale_regularization > 0.0:
            optimization_loss = optimization_loss + scale_regularization * metricdict["scale"]
    pre = "Population/" if tag == "" else "Population_" + tag + "/"
    auxdict = {**auxdict, **{pre + key: value for key, value in metricdict.items()}}
    auxdict["optimization_loss"] = optimization_loss
    auxdict = {key: value.detach().item() for key, value in auxdict.items()}
    return optimization_loss, auxdict
