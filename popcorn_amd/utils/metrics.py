"""Census-level evaluation metrics with the key names the reference logs (utils/metrics.py:12-24):
``get_test_metrics(pred, y, tag)`` -> {"Population_<tag>/<metric>": tensor}."""
import torch

from .losses import mape_func, r2

# metric name -> function of (pred, y, log1p(pred), log1p(y)); insertion order = the reference's key order
_METRICS = (
    ("l1_loss", lambda p, y, lp, ly: (p - y).abs().mean()),
    ("r2", lambda p, y, lp, ly: r2(p, y)),
    ("mape", lambda p, y, lp, ly: mape_func(p, y)),
    ("log_l1_loss", lambda p, y, lp, ly: (lp - ly).abs().mean()),
    ("mse_loss", lambda p, y, lp, ly: (p - y).square().mean()),
    ("log_mse_loss", lambda p, y, lp, ly: (lp - ly).square().mean()),
    ("Correlation", lambda p, y, lp, ly: torch.corrcoef(torch.stack((p, y)))[0, 1]),
)


def get_test_metrics(pred, y, tag=""):
    lp, ly = torch.log(pred + 1), torch.log(y + 1)
    prefix = f"Population_{tag}/"
    return {prefix + name: fn(pred, y, lp, ly) for name, fn in _METRICS}
