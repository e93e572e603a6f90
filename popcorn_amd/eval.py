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
    """Device-resident accumulators of run_eval.py:84-90 and the write-back / averaging of :127-154.

    Multi-GPU (windows sharded over ranks): ``world`` pads the row count to a multiple of the world size so that
    ``reduce_scatter`` can hand every rank the SUM of one equal row band (one ring pass over the four fp32 planes -- half the
    bytes of an all-reduce); the visit-count map is not communicated at all: it depends only on the window list, so every
    rank adds the windows of the OTHER ranks to its own count map (``add_count_only``).  Each rank then finalises its band;
    ``gather_bands`` assembles the full maps where a caller wants them (the census sums do not: ``census_sums`` on the band
    plus one tiny all-reduce of the per-region sums)."""

    def __init__(self, h, w, device, with_scale=True, world=1):
        if torch.device(device).type != "cuda":
            raise L.PopcornHipError("Stitcher accumulates on a HIP device only")
        self.h, self.w, self.world = h, w, max(1, int(world))
        self.hb = -(-h // self.world)                      # rows per rank band
        self.hp = self.hb * self.world                     # padded row count (rows >= h are never written)
        # the fp32 accumulators are planes of ONE allocation
        self.acc = torch.zeros(4 if with_scale else 2, self.hp, w, dtype=torch.float32, device=device)
        self.out, self.out_sq = self.acc[0][:h], self.acc[1][:h]
        self.scale, self.scale_sq = (self.acc[2][:h], self.acc[3][:h]) if with_scale else (None, None)
        self.count = torch.zeros(self.hp, w, dtype=torch.int16, device=device)[:h]
        self.band = None                                   # (row0, row1) once the accumulators hold only this rank's band

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

    def add_count_only(self, xl, yl, M, ps, overlap=OVERLAP):
        """A window another rank computed: only its visit count (interior += M), no data."""
        x0, x1 = xl + overlap, min(xl + ps - overlap, self.h)
        y0, y1 = yl + overlap, min(yl + ps - overlap, self.w)
        if x1 > x0 and y1 > y0:
            self.count[x0:x1, y0:y1] += M

    def add_counts_only(self, windows, M, ps, overlap=OVERLAP):
        """The visit counts of MANY windows other ranks computed, in ONE launch of ``census.hip`` (pc_stitch_count_windows) instead of one
        slice-add per window from the host loop.  windows: iterable of (row origin, column origin)."""
        ws = [(int(x), int(y)) for x, y in windows]
        if not ws:
            return
        # clipped interiors {x0, x1, y0, y1} of all windows as ONE small device array; census.hip counts them into the map in one launch
        # (round 6: the difference-array form did this bookkeeping with stock torch ops -- index_put_, two cumsum_, a banded int16
        # conversion -- and a transient int32 plane of the raster's size)
        rows = []
        for x, y in ws:
            x0, x1 = min(x + overlap, self.h), min(x + ps - overlap, self.h)
            y0, y1 = min(y + overlap, self.w), min(y + ps - overlap, self.w)
            if x1 > x0 and y1 > y0:
                rows.append((x0, x1, y0, y1))
        if not rows:
            return
        win = torch.tensor(rows, dtype=torch.int32).to(self.count.device)
        L.check(L.lib().pc_stitch_count_windows(L.ptr(win), len(rows), int(M), L.ptr(self.count), self.h, self.w, L.stream_ptr()),
                "pc_stitch_count_windows")

    def all_reduce(self, reducer: FlatReducer):
        """Multi-GPU, simple form: sum the full accumulators on every rank (interiors of regular windows are disjoint, the
        bottom/right catch-up windows overlap them -- the count map handles both)."""
        if reducer.active:
            import torch.distributed as dist
            dist.all_reduce(self.acc, group=reducer.group)            # all fp32 planes in one ring pass
            c = self.count.to(torch.int32)                            # RCCL has no 16-bit integer sum
            dist.all_reduce(c, group=reducer.group)
            self.count.copy_(c)

    def reduce_scatter(self, reducer: FlatReducer, rank):
        """Multi-GPU, band form: this rank ends up with the summed accumulators of rows [rank * hb, (rank + 1) * hb) (clipped
        to h) in place; the other rows of its planes are stale afterwards.  The count map must already be complete
        (``add_count_only`` for the other ranks' windows).  Returns (row0, row1)."""
        r0, r1 = min(rank * self.hb, self.h), min((rank + 1) * self.hb, self.h)
        if reducer.active:
            import torch.distributed as dist
            assert reducer.world == self.world, "Stitcher(world=...) must equal the reducer's world size"
            if reducer.backend == "nccl":
                for p in range(self.acc.shape[0]):
                    plane = self.acc[p]                                # (hp, w) contiguous: chunk r = band r
                    dist.reduce_scatter_tensor(plane[rank * self.hb:(rank + 1) * self.hb], plane, group=reducer.group)
            else:
                # gloo (functional runs) has no reduce-scatter on device tensors: all-reduce, then make the result look like one --
                # the rows of the OTHER ranks' bands are poisoned, so anything that reads a stale row after this call (which the
                # RCCL branch leaves unsummed) shows up as NaN in the functional tests instead of passing by accident
                dist.all_reduce(self.acc, group=reducer.group)
                self.acc[:, :rank * self.hb] = float("nan")
                self.acc[:, (rank + 1) * self.hb:] = float("nan")
        self.band = (r0, r1)
        return self.band

    def finalize(self):
        """mean / std where count > 1 (run_eval.py:140-154) over the whole raster, or over this rank's band after
        ``reduce_scatter``.  Returns the four maps (full-raster views; after a reduce_scatter only the band rows are valid)."""
        r0, r1 = self.band if self.band is not None else (0, self.h)
        n = (r1 - r0) * self.w
        if n > 0:
            sl = lambda t: None if t is None else t[r0:r1]  # noqa: E731
            L.check(L.lib().pc_stitch_finalize(L.ptr(sl(self.out)), L.ptr(sl(self.out_sq)), L.ptr(sl(self.scale)),
                                               L.ptr(sl(self.scale_sq)), L.ptr(sl(self.count)), C.c_int64(n), L.stream_ptr()),
                    "pc_stitch_finalize")
        return self.out, self.out_sq, self.scale, self.scale_sq

    def gather_bands(self, reducer: FlatReducer):
        """After reduce_scatter + finalize: every rank receives every band (the full finalised maps)."""
        if reducer.active and self.band is not None:
            import torch.distributed as dist
            for p in range(self.acc.shape[0]):
                plane = self.acc[p]
                if reducer.backend == "nccl":
                    r = dist.get_rank(reducer.group)
                    dist.all_gather_into_tensor(plane, plane[r * self.hb:(r + 1) * self.hb].clone(), group=reducer.group)
                else:
                    parts = [torch.empty(self.hb, self.w, device=plane.device) for _ in range(self.world)]
                    r = dist.get_rank(reducer.group)
                    dist.all_gather(parts, plane[r * self.hb:(r + 1) * self.hb].contiguous(), group=reducer.group)
                    plane.copy_(torch.cat(parts, 0))
        self.band = None
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


def census_sums_sharded(st: "Stitcher", boundary, num_ids, reducer: FlatReducer):
    """Region sums of a map that is distributed by row band (``evaluate_raster(..., gather=False)``): each rank sums its own
    band, then ONE all-reduce of the float64 per-region sums (num_ids * 8 bytes) -- the full map never travels."""
    r0, r1 = st.band if st.band is not None else (0, st.h)
    if r1 > r0:
        # (a band starts at byte r0 * w * 4 of the plane: any 4-byte alignment -- pc_census_sum takes a scalar head)
        sums = census_sums(st.out[r0:r1], boundary[r0:r1].contiguous().to(torch.int32), num_ids)
    else:
        sums = torch.zeros(num_ids, dtype=torch.float64, device=st.out.device)
    if reducer.active:
        import torch.distributed as dist
        dist.all_reduce(sums, group=reducer.group)
    return sums


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
                    reducer: FlatReducer | None = None, rank=0, band_reduce=True, gather=True, return_stitcher=False):
    """Ensemble sliding-window inference over ``raster`` = callable (x, y, season, ps) -> normalised model input
    (1,6,ps,ps) on the device (the reference's Population_Dataset(mode="test") item, PopulationDataset.py:336-420), or a
    (S,6,h,w) device tensor of pre-normalised seasons.  Returns the finalised (mean map, std map, scale mean, scale std).

    Windows are assigned round-robin to ranks (no data-path collective).  Multi-GPU: ``band_reduce`` (default) sums the
    accumulators with one reduce-scatter by row band, every rank finalises its own band, and ``gather`` decides whether the
    bands are then all-gathered into full maps on every rank (True: the four maps are returned, as in the single-process
    case) or stay distributed (False: the ``Stitcher`` is returned; ``census_sums_sharded`` works on the band).
    ``band_reduce=False``: the round-2 form, an all-reduce of the full planes and of the count map.
    ``return_stitcher``: also return the ``Stitcher`` (its visit-count map)."""
    reducer = reducer or FlatReducer()
    if torch.is_tensor(raster):
        h, w = raster.shape[-2:]
        tensor = raster
        raster = lambda x, y, s, ps: tensor[s:s + 1, :, x:x + ps, y:y + ps]  # noqa: E731
    else:
        h, w = raster.shape
    dev = next(models[0].parameters()).device
    st = Stitcher(h, w, dev, world=reducer.world)
    idx = get_patch_indices(h, w, patchsize, overlap, fourseasons)
    mine = set(shard_indices(idx.shape[0], rank, reducer.world))
    if band_reduce and reducer.world > 1:
        # the visit count needs no collective: the windows of the OTHER ranks enter this rank's count map analytically, all at once
        st.add_counts_only([(int(idx[i][0]), int(idx[i][1])) for i in range(idx.shape[0]) if i not in mine], len(models), patchsize, overlap)
    for i in range(idx.shape[0]):
        x, y, season = (int(v) for v in idx[i])
        if i not in mine:
            continue
        inp = raster(x, y, season, patchsize).contiguous()
        sample = {"input": inp}
        pds, scs = [], []
        with torch.no_grad(), L.padded_rows():      # (16-byte aligned rows for the levels whose width is not a multiple of 4)
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
    if band_reduce and reducer.world > 1:
        # one reduce-scatter by row band (half the bytes of an all-reduce, no count collective), every rank finalises its band
        st.reduce_scatter(reducer, rank)
        st.finalize()
        if not gather:
            return st
        maps = st.gather_bands(reducer)
        return (maps, st) if return_stitcher else maps
    st.all_reduce(reducer)
    maps = st.finalize()
    return (maps, st) if return_stitcher else maps
