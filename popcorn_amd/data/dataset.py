"""Synthetic stand-in for the reference's ``Population_Dataset`` (data/PopulationDataset.py:30-37), shaped like its
two modes, for environments without the GeoTIFF archive (no rasterio / no data here):

  * mode="weaksup": one census region per item -- a variable-size crop with S2 (4,h,w) raw reflectances, S1 (2,h,w)
    backscatter in dB, ``admin_mask`` (h,w) of region ids, the census count ``y`` and ``census_idx`` (:425-470);
  * mode="test": a raster (seasons, 6, H, W) walked with 2048-px windows (:336-420) plus a census table / boundary map
    in the on-disk format written by utils/02_preprocess_rwa_shapefile.py:142-164 (idx, POP20, bbox, count).
"""
from __future__ import annotations

import torch
from torch.utils.data import Dataset

from . import stats


class SyntheticWeaksupDataset(Dataset):
    def __init__(self, n_regions=256, min_hw=64, max_hw=144, seed=1600, fixed_hw=None):
        self.n = n_regions
        g = torch.Generator().manual_seed(seed)
        if fixed_hw is not None:
            self.hw = [tuple(fixed_hw)] * n_regions
        else:
            hw = torch.randint(min_hw, max_hw + 1, (n_regions, 2), generator=g)
            self.hw = [tuple(int(v) for v in r) for r in hw]
        self.seed = seed
        self._cache = {}

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        # deterministic per index: generated once, then served from memory (the reference reads tiles from disk)
        if i not in self._cache:
            self._cache[i] = self._make(i)
        return dict(self._cache[i])

    def _make(self, i):
        h, w = self.hw[i]
        g = torch.Generator().manual_seed(self.seed * 7919 + i)
        s2 = torch.randint(0, 10000, (4, h, w), generator=g).float()
        s1 = torch.randn(2, h, w, generator=g) * torch.tensor(stats.S1_STD).view(2, 1, 1) + torch.tensor(stats.S1_MEAN).view(2, 1, 1)
        yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        cy, cx = h / 2 + (i % 5) - 2, w / 2 - (i % 3) + 1
        r = 0.35 * min(h, w)
        inside = ((yy - cy) ** 2 + (xx - cx) ** 2) < r * r
        cid = i + 1
        admin = torch.where(inside, float(cid), float(cid + self.n))
        y = torch.rand(1, generator=g).item() * 500.0
        return {"S2": s2, "S1": s1, "admin_mask": admin, "y": torch.tensor(y), "census_idx": torch.tensor([cid]),
                "img_coords": (0, 0), "valid_coords": (0, 0), "season": i % 4}


class SyntheticTestRaster:
    """A normalised (seasons,6,H,W) raster with a blocky census map (ids 1..n, 0 = outside) and a census table."""

    def __init__(self, h=2304, w=2560, seasons=1, n_regions=400, seed=1610, device="cpu"):
        g = torch.Generator().manual_seed(seed)
        self.raster = torch.randn(seasons, 6, h, w, generator=g).to(device)
        gy = int(n_regions ** 0.5)
        gx = (n_regions + gy - 1) // gy
        ys = torch.clamp((torch.arange(h) * gy) // h, max=gy - 1)
        xs = torch.clamp((torch.arange(w) * gx) // w, max=gx - 1)
        b = ys[:, None] * gx + xs[None, :] + 1
        b[b > n_regions] = 0
        self.boundary = b.to(torch.int32).to(device)
        self.census_idx = torch.arange(1, n_regions + 1)
        self.census_pop = torch.rand(n_regions, generator=g) * 2000
        self.shape = (h, w)
