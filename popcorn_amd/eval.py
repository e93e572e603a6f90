"""Evaluation counterpart of the reference's ``run_eval.py`` Trainer.test_target (run_eval.py:71-203) and the census
helpers of ``Population_Dataset`` (data/PopulationDataset.py:294-334, 656-672, 675-852):

  sliding windows (2048 px, 128 px overlap, interior-only write-back) over a raster  ->  ensemble forward (HIP)  ->
  device-resident (h,w) accumulators {sum, sum^2, scale sum, scale sum^2, count}  ->  mean / std  ->
  one-pass census aggregation (segment sum)  ->  metrics  ->  dasymetric adjustment  ->  metrics again.

Relative to the reference: the frozen building extractor (identical in every ensemble member, popcorn.py:96) runs ONCE
per window instead of once per member; accumulators never leave the device; the per-census-row Python loop is one
kernel.  Windows are independent, so multi-GPU evaluation shards them round-robin and sum-reduces the accumulators.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from .distributed import FlatReducer, shard_indices

INFERENCE_PATCH_SIZE = 2048      # utils/constants.py:12
OVERLAP = 128                    # utils/constants.py:13


def get_patch_indices(h, w, patchsize=INFERENCE_PATCH_SIZE, overlap=OVERLAP, fourseasons=False):
    """(x, y, season) window origins: regular grid of stride patchsize - 2*overlap plus a bottom row, a right column and
    the bottom-right corner, repeated per season.  data/PopulationDataset.py:294-334."""
    stride = patchsize - 2 * overlap
    x = torch.arange(0, h - patchsize, stride, dtype=int)
    y = torch.arange(0, w - patchsize, stride, dtype=int)
    main = torch.cartesian_prod(x, y).reshape(-1, 2)
    max_x, max_y = h - patchsize, w - patchsize
    bottom = torch.stack([torch.full((len(y),), max_x, dtype=int), y]).T
    right = torch.stack([x, torch.full((len(x),), max_y, dtype=int)]).T
    corner = torch.tensor([[max_x, max_y]])
    main = torch.cat([main, bottom, right, corner])
    seasons = range(4) if fourseasons else range(1)
    return torch.cat([torch.cat([main, torch.full((main.shape[0], 1), s, dtype=int)], dim=1) for s in seasons], dim=0)


def create_mask(patchsize_x, patchsize_y, overlap):
    """Interior of a window (data/PopulationDataset.py:656-672)."""
    m = torch.zeros(patchsize_x, patchsize_y, dtype=torch.bool)
    m[overlap:patchsize_x - overlap, overlap:patchsize_y - overlap] = True
    return m


class Stitcher:
    """Device-resident accumulators of run_eval.py:84-90 and the write-back / averaging of :127-154."""

    def __init__(self, h, w, device, with_scale=True):
        if torch.device(device).type != "cuda":
            raise L.PopcornHipError("Stitcher accumulates on a HIP device only")
        self.h, self.w = h, w
        # the fp32 accumulators are planes of ONE allocation: a multi-GPU run sums them with a single collective
        self.acc = torch.zeros(4 if with_scale else 2, h, w, dtype=torch.float32, device=device)
        self.out, self.out_sq = self.acc[0], self.acc[1]
        self.scale, self.scale_sq = (self.acc[2], self.acc[3]) if with_scale else (None, None)
        self.count = torch.zeros(h, w, dtype=torch.int16, device=device)

    def add_window(self, xl, yl, popdense, scale=None, overlap=OVERLAP):
        """popdense / scale: (M, ps, ps) member outputs of the window whose origin is row xl, column yl (the reference's
        ``img_coords``)."""
        L.require_device(popdense)
        M, psx, psy = popdense.shape
        popdense = popdense.contiguous().float()
        scale = scale.contiguous().float() if scale is not None and self.scale is not None else None
        L.check(L.lib().pc_stitch_accumulate(L.ptr(popdense), L.ptr(scale), M, psx, psy, overlap, int(xl), int(yl),
                                             L.ptr(self.out), L.ptr(self.out_sq), L.ptr(self.scale), L.ptr(self.scale_sq),
                                             L.ptr(self.count), self.h, self.w, L.stream_ptr()), "pc_stitch_accumulate")

    def all_reduce(self, reducer: FlatReducer):
        """Multi-GPU: windows were sharded over ranks; sum the accumulators (interiors of regular windows are disjoint,
        the bottom/right catch-up windows overlap them -- the count map handles both)."""
        if reducer.active:
            import torch.distributed as dist
            dist.all_reduce(self.acc, group=reducer.group)            # all fp32 planes in one ring pass
            c = self.count.to(torch.int32)                            # RCCL has no 16-bit integer sum
            dist.all_reduce(c, group=reducer.group)
            self.count.copy_(c)

    def finalize(self):
        n = self.h * self.w
        L.check(L.lib().pc_stitch_finalize(L.ptr(self.out), L.ptr(self.out_sq), L.ptr(self.scale), L.ptr(self.scale_sq),
                                           L.ptr(self.count), C.c_int64(n), L.stream_ptr()), "pc_stitch_finalize")
        return self.out, self.out_sq, self.scale, self.scale_sq


def census_sums(pred, boundary, num_ids, want_counts=False):
    """sums[id] = sum(pred[boundary == id]) for id in [0, num_ids) -- one pass (segment sum).  pred: (h,w) f32,
    boundary: (h,w) int32.  Returns float64 sums (and int32 counts)."""
    L.require_device(pred, boundary)
    assert pred.dtype == torch.float32 and boundary.dtype == torch.int32 and pred.is_contiguous() and boundary.is_contiguous()
    sums = torch.empty(num_ids, dtype=torch.float64, device=pred.device)
    counts = torch.empty(num_ids, dtype=torch.int32, device=pred.device) if want_counts else None
    L.check(L.lib().pc_census_sum(L.ptr(pred), L.ptr(boundary), C.c_int64(pred.numel()), num_ids, L.ptr(sums),
                                  L.ptr(counts), L.stream_ptr()), "pc_census_sum")
    return (sums, counts) if want_counts else sums


def convert_popmap_to_census(pred, boundary, census_idx, census_pop):
    """data/PopulationDataset.py:675-820 without the GeoTIFF/CSV I/O: returns (census_pred, census_gt) for the census
    rows ``census_idx`` (region ids) / ``census_pop`` (POP20)."""
    census_idx = torch.as_tensor(census_idx, dtype=torch.int64, device=pred.device)
    num_ids = int(census_idx.max().item()) + 1 if census_idx.numel() else 1
    sums = census_sums(pred.contiguous().float(), boundary.contiguous().to(torch.int32), num_ids)
    census_pred = sums[census_idx].to(torch.float32)
    census_gt = torch.as_tensor(census_pop, dtype=torch.float32, device=pred.device)
    return census_pred, census_gt


def adjust_map_to_census(pred, boundary, census_idx, census_pop):
    """Dasymetric rescale so that every census region sums to its census count (data/PopulationDataset.py:823-852).
    In place on ``pred`` like the reference; returns it."""
    L.require_device(pred, boundary)
    census_idx = torch.as_tensor(census_idx, dtype=torch.int64, device=pred.device)
    num_ids = int(census_idx.max().item()) + 1 if census_idx.numel() else 1
    b32 = boundary.contiguous().to(torch.int32)
    sums = census_sums(pred, b32, num_ids)
    pop = torch.zeros(num_ids, dtype=torch.float32, device=pred.device)
    has = torch.zeros(num_ids, dtype=torch.uint8, device=pred.device)
    pop[census_idx] = torch.as_tensor(census_pop, dtype=torch.float32, device=pred.device)
    has[census_idx] = 1
    L.check(L.lib().pc_census_adjust(L.ptr(pred), L.ptr(b32), C.c_int64(pred.numel()), num_ids, L.ptr(sums), L.ptr(pop),
                                     L.ptr(has), L.stream_ptr()), "pc_census_adjust")
    return pred


def evaluate_raster(models, raster, patchsize=INFERENCE_PATCH_SIZE, overlap=OVERLAP, fourseasons=False,
                    reducer: FlatReducer | None = None, rank=0):
    """Ensemble sliding-window inference over ``raster`` = callable (x, y, season, ps) -> normalised model input
    (1,6,ps,ps) on the device (the reference's Population_Dataset(mode="test") item, PopulationDataset.py:336-420), or a
    (S,6,h,w) device tensor of pre-normalised seasons.  Returns the finalised (mean map, std map, scale mean, scale std).

    Windows are assigned round-robin to ranks (no data-path collective); accumulators are summed once at the end."""
    reducer = reducer or FlatReducer()
    if torch.is_tensor(raster):
        h, w = raster.shape[-2:]
        tensor = raster
        raster = lambda x, y, s, ps: tensor[s:s + 1, :, x:x + ps, y:y + ps]  # noqa: E731
    else:
        h, w = raster.shape
    dev = next(models[0].parameters()).device
    st = Stitcher(h, w, dev)
    idx = get_patch_indices(h, w, patchsize, overlap, fourseasons)
    for i in shard_indices(idx.shape[0], rank, reducer.world):
        x, y, season = (int(v) for v in idx[i])
        inp = raster(x, y, season, patchsize).contiguous()
        sample = {"input": inp}
        pds, scs = [], []
        with torch.no_grad():
            for j, m in enumerate(models):
                m.eval()
                if j > 0 and m.sentinelbuildings and "building_counts" in sample:
                    # identical frozen extractor in every member: reuse the score of member 0
                    keep = m.sentinelbuildings
                    m.sentinelbuildings = False
                    o = m(sample, padding=False)
                    m.sentinelbuildings = keep
                else:
                    o = m(sample, padding=False)
                pds.append(o["popdensemap"][0])
                if o.get("scale") is not None:
                    scs.append(o["scale"][0])
        st.add_window(x, y, torch.stack(pds), torch.stack(scs) if scs else None, overlap)
    st.all_reduce(reducer)
    return st.finalize()
