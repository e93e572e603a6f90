"""Synthetic PopulationDataset-shaped batches: 15-band 100x100 tiles, a census-region mask
per tile, a census count per tile.  Used by bench.py, the smoke test and the training counterpart; there is no
network for real Sentinel data in this environment."""
from __future__ import annotations

import torch

from . import stats


def make_raw_batch(B, H=100, W=100, seed=1600, device="cpu", region="full"):
    """Returns dict(raw (B,15,H,W) f32, admin_mask (B,H,W) f32, census_idx (B,) i64, y (B,) f32).
    S2 bands ~ U{0..9999} (the reference's own fake-data branch, data/PopulationDataset.py:581), S1 ~ N(mean, std) of
    the 'sen1' stats (dB).  region='full': the census region covers the tile (worst case: every pixel selected);
    'disc': a disc-shaped region inside a second region id, with collate-style -1 padding rows."""
    g = torch.Generator().manual_seed(seed)
    s2 = torch.randint(0, 10000, (B, 13, H, W), generator=g).float()
    s1 = torch.randn(B, 2, H, W, generator=g) * torch.tensor(stats.S1_STD).view(1, 2, 1, 1) \
        + torch.tensor(stats.S1_MEAN).view(1, 2, 1, 1)
    raw = torch.cat([s2, s1], 1)
    ids = torch.arange(1, B + 1, dtype=torch.int64)
    if region == "full":
        admin = ids.view(B, 1, 1).float().expand(B, H, W).contiguous()
    else:
        yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        admin = torch.empty(B, H, W)
        for b in range(B):
            r = 0.3 * min(H, W) + (b % 5)
            disc = ((yy - H / 2) ** 2 + (xx - W / 2) ** 2) < r * r
            admin[b] = torch.where(disc, float(ids[b]), float(ids[b] + B))
            admin[b, :2] = -1.0
    y = torch.rand(B, generator=g) * 500.0
    out = {"raw": raw, "admin_mask": admin, "census_idx": ids, "y": y}
    return {k: v.to(device) for k, v in out.items()}

