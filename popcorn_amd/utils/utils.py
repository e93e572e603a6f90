"""Sample preparation of the train / eval loops (reference utils/utils.py:105-214), device-agnostic: the reference
hard-codes ``.cuda()``; here the constants follow the tensors."""
from __future__ import annotations

import torch

from ..data import stats


def default_dataset_stats():
    """The entries of data/config/dataset_stats.json the model path uses."""
    return {"sen2springNIR": {"mean": torch.tensor(stats.S2_MEAN), "std": torch.tensor(stats.S2_STD)},
            "sen2spring": {"mean": torch.tensor(stats.S2_MEAN[:3]), "std": torch.tensor(stats.S2_STD[:3])},
            "sen1": {"mean": torch.tensor(stats.S1_MEAN), "std": torch.tensor(stats.S1_STD)}}


def _norm(x, st):
    mean = st["mean"].to(x.device, x.dtype).view(1, -1, 1, 1)
    std = st["std"].to(x.device, x.dtype).view(1, -1, 1, 1)
    return (x - mean) / std


def apply_normalize(indata, dataset_stats):
    """utils/utils.py:105-127: per-channel (x - mean) / std for S2 (4-band 'sen2springNIR' or 3-band 'sen2spring') and S1."""
    if "S2" in indata:
        indata["S2"] = _norm(indata["S2"], dataset_stats["sen2springNIR" if indata["S2"].shape[1] == 4 else "sen2spring"])
    if "S1" in indata:
        indata["S1"] = _norm(indata["S1"], dataset_stats["sen1"])
    return indata


def apply_transformations_and_normalize(sample, transform, dataset_stats, buildinginput=False, segmentationinput=False):
    """utils/utils.py:130-214: modality-wise transforms -> normalise -> input = cat[S2, S1] -> the 'general' transform
    applied jointly to the input and the stacked {admin_mask, building_counts, ...} maps."""
    if transform is not None:
        for key in ("S2", "S1"):
            if key in transform and key in sample:
                sample[key] = transform[key](sample[key])
    sample = apply_normalize(sample, dataset_stats)
    parts = [sample[k] for k in ("S2", "S1") if k in sample]
    sample["input"] = torch.cat(parts, dim=1) if parts else None
    if transform is not None and "general" in transform and sample["input"] is not None:
        keys = [k for k in ("admin_mask", "positional_encoding", "building_counts", "building_segmentation") if k in sample]
        data = [sample[k].unsqueeze(1) if k == "admin_mask" else sample[k] for k in keys]
        if data:
            lens = [d.shape[1] for d in data]
            sample["input"], stacked = transform["general"]((sample["input"], torch.cat(data, dim=1)))
            start = 0
            for k, n in zip(keys, lens):
                sample[k] = stacked[:, start:start + n]
                if k == "admin_mask":
                    sample[k] = sample[k][:, 0]
                start += n
        else:
            sample["input"] = transform["general"](sample["input"])
    return sample
