"""Batch assembly with the reference's semantics (Population_Dataset_collate_fn, data/PopulationDataset.py:885-958):
variable-size census-region crops are zero-padded to the batch maximum (bottom/right), ``admin_mask`` is padded with
-1, ``census_idx`` concatenated.  Host-side glue (runs in DataLoader workers), plain torch."""
from __future__ import annotations

import torch


def Population_Dataset_collate_fn(batch):
    use_S2, use_S1 = "S2" in batch[0], "S1" in batch[0]
    max_x = max_y = 0
    for key in ("S2", "S1", "building_counts"):
        if key in batch[0]:
            max_x = max(item[key].shape[1] for item in batch)
            max_y = max(item[key].shape[2] for item in batch)
    n = len(batch)
    out = {}
    if use_S2:
        out["S2"] = torch.zeros(n, batch[0]["S2"].shape[0], max_x, max_y)
    if use_S1:
        out["S1"] = torch.zeros(n, batch[0]["S1"].shape[0], max_x, max_y)
    if "building_counts" in batch[0]:
        out["building_counts"] = torch.zeros(n, 1, max_x, max_y)
    admin = (-1) * torch.ones(n, max_x, max_y)
    y = torch.zeros(n)
    for i, item in enumerate(batch):
        for key in ("S2", "S1", "building_counts"):
            if key in out:
                xs, ys = item[key].shape[1], item[key].shape[2]
                out[key][i, :, :xs, :ys] = item[key]
        y[i] = item["y"]
        xs, ys = item["admin_mask"].shape[0], item["admin_mask"].shape[1]
        admin[i, :xs, :ys] = item["admin_mask"]
    out.update({
        "admin_mask": admin,
        "y": y,
        "img_coords": [item["img_coords"] for item in batch],
        "valid_coords": [item["valid_coords"] for item in batch],
        "season": torch.tensor([item["season"] for item in batch]),
        "census_idx": torch.cat([item["census_idx"] for item in batch]),
    })
    return out


def collate_into(batch, alloc):
    """``Population_Dataset_collate_fn`` with the output tensors taken from ``alloc(key, shape, dtype)`` (the feed's pinned staging ring:
    the batch is assembled where the H2D copy reads it, one pass over the items instead of collate + a second copy) and only the PADDING
    filled (zeros; -1 for ``admin_mask``) instead of whole tensors.  Same keys, shapes and values as the plain collate."""
    max_x = max_y = 0
    for key in ("S2", "S1", "building_counts"):
        if key in batch[0]:
            max_x = max(item[key].shape[1] for item in batch)
            max_y = max(item[key].shape[2] for item in batch)
    n = len(batch)
    out = {}
    for key in ("S2", "S1", "building_counts"):
        if key in batch[0]:
            t = alloc(key, (n, 1 if key == "building_counts" else batch[0][key].shape[0], max_x, max_y), torch.float32)
            for i, item in enumerate(batch):
                xs, ys = item[key].shape[1], item[key].shape[2]
                t[i, :, :xs, :ys] = item[key]
                if xs < max_x:
                    t[i, :, xs:, :] = 0
                if ys < max_y:
                    t[i, :, :xs, ys:] = 0
            out[key] = t
    admin = alloc("admin_mask", (n, max_x, max_y), torch.float32)
    y = alloc("y", (n,), torch.float32)
    for i, item in enumerate(batch):
        xs, ys = item["admin_mask"].shape[0], item["admin_mask"].shape[1]
        admin[i, :xs, :ys] = item["admin_mask"]
        if xs < max_x:
            admin[i, xs:, :] = -1
        if ys < max_y:
            admin[i, :xs, ys:] = -1
        y[i] = item["y"]
    cidx = torch.cat([item["census_idx"] for item in batch])
    ci = alloc("census_idx", tuple(cidx.shape), cidx.dtype)
    ci.copy_(cidx)
    out.update({
        "admin_mask": admin,
        "y": y,
        "img_coords": [item["img_coords"] for item in batch],
        "valid_coords": [item["valid_coords"] for item in batch],
        "season": torch.tensor([item["season"] for item in batch]),
        "census_idx": ci,
    })
    return out


def augment_geometric(inp, admin_mask, generator=None, p_flip=0.5, p_rot=0.75):
    """The joint geometric augmentation of the reference (utils/transform.py:54-200: RandomVerticalFlip /
    RandomHorizontalFlip (per sample) and RandomRotationTransform([90,180,270], p=.75) with expand=True), restated with
    torch.flip / torch.rot90 (exact for right angles; no torchvision).  inp: (B,C,H,W), admin_mask: (B,H,W).
    The rotation is batch-wide (expand=True changes the shape), as in the reference."""
    B = inp.shape[0]
    sel = torch.rand(B, generator=generator) < p_flip
    inp, admin_mask = inp.clone(), admin_mask.clone()
    inp[sel] = torch.flip(inp, dims=(-2,))[sel]
    admin_mask[sel] = torch.flip(admin_mask, dims=(-2,))[sel]
    sel = torch.rand(B, generator=generator) < p_flip
    inp[sel] = torch.flip(inp, dims=(-1,))[sel]
    admin_mask[sel] = torch.flip(admin_mask, dims=(-1,))[sel]
    if torch.rand(1, generator=generator) < p_rot:
        k = int(torch.randint(1, 4, (1,), generator=generator))
        inp = torch.rot90(inp, k, dims=(-2, -1))
        admin_mask = torch.rot90(admin_mask, k, dims=(-2, -1))
    return inp.contiguous(), admin_mask.contiguous()
