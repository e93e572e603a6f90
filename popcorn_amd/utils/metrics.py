"""utils/metrics.py:12-24 surface: ``get_test_metrics(pred, y, tag)`` on census-level vectors."""
import torch
import torch.nn.functional as F

from .losses import mape_func, r2


def get_test_metrics(pred, y, tag=""):
    log_dict = {
        "l1_loss": F.l1_loss(pred, y),
        "r2": r2(pred, y),
        "mape": mape_func(pred, y),
        "log_l1_loss": F.l1_loss(torch.log(pred + 1), torch.log(y + 1)),
        "mse_loss": F.mse_loss(pred, y),
        "log_mse_loss": F.mse_loss(torch.log(pred + 1), torch.log(y + 1)),
        "Correlation": torch.corrcoef(torch.stack([pred, y]))[0, 1],
    }
    return {"Population_" + tag + "/" + key: value for key, value in log_dict.items()}
