"""BASELINE config 5 at its real raster size: a synthetic raster of the pricp2 extent (reference README.md:233-234 at 10 m:
about 7,200 x 23,100 px), 2048-px windows with 128 px overlap (utils/constants.py:12-13), ``--fourseasons`` = 208 windows
(data/PopulationDataset.py:294-334), through the HIP forward, the device-resident stitcher and the census kernels.

The raster is a CALLABLE (window origin, season -> normalised input generated on the device): no 16 GB input tensor exists.
Size-independent properties (the oracle cannot run at this size): window count; the visit-count map equals the analytic
count everywhere; the far corner of the raster (plane offsets beyond 2^31 bytes) holds exactly the corner windows' forward;
census sum of the map == masked map sum; and, on a 1/8 crop, two ranks == one process."""
import os
import socket
import time

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

H, W, PS, OV = 7200, 23100, 2048, 128


class SyntheticRaster:
    """(x, y, season, ps) -> (1, 6, ps, ps) normalised input, a smooth deterministic function of the GLOBAL pixel coordinates
    and the season (so that any window / any rank sees the same raster)."""

    def __init__(self, h, w, device="cuda"):
        self.shape = (h, w)
        self.device = device

    def __call__(self, x, y, season, ps):
        i = torch.arange(x, x + ps, device=self.device, dtype=torch.float32).view(1, ps, 1)
        j = torch.arange(y, y + ps, device=self.device, dtype=torch.float32).view(1, 1, ps)
        c = torch.arange(6, device=self.device, dtype=torch.float32).view(6, 1, 1)
        v = torch.sin(0.0131 * (c + 1.0) * i + 0.0173 * j + 0.7 * season) + 0.5 * torch.cos(0.0071 * i * (1.0 + 0.1 * c) - 0.011 * j)
        return v.unsqueeze(0)


def _coverage(n, ps, ov):
    stride = ps - 2 * ov
    starts = sorted(set(list(range(0, n - ps, stride)) + [n - ps]))
    cov = np.zeros(n, dtype=np.int64)
    for s in starts:
        cov[s + ov:s + ps - ov] += 1
    return cov, starts


def _model():
    from popcorn_amd.model import POPCORN
    torch.manual_seed(1600)
    return POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()


def test_country_scale_fourseasons_raster_properties(capsys):
    from popcorn_amd import eval as E
    m = _model()
    raster = SyntheticRaster(H, W)
    idx = E.get_patch_indices(H, W, PS, OV, True)
    assert idx.shape[0] == 208                                     # 52 windows per season
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    (mean, std, smean, sstd), st = E.evaluate_raster([m], raster, PS, OV, True, return_stitcher=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    with capsys.disabled():
        print(f"\\n[config 5] {idx.shape[0]} windows of {PS}x{PS} over {H}x{W} px in {dt:.2f} s = {idx.shape[0] / dt:.1f} windows/s "
              f"({idx.shape[0] * PS * PS / dt / 1e6:.0f} Mpx/s), accumulators {st.acc.numel() * 4 / 2**30:.2f} GiB")
    assert st.acc.numel() * 4 > 2 ** 31                           # the planes span more than 2^31 bytes: offsets must be 64-bit
    # 1. visit count == analytic count (the window set is a cartesian product of row and column origins), everywhere
    rc, xs = _coverage(H, PS, OV)
    cc, ys = _coverage(W, PS, OV)
    want = torch.from_numpy(rc).cuda().view(-1, 1) * torch.from_numpy(cc).cuda().view(1, -1) * 4        # 4 seasons x 1 member
    assert torch.equal(st.count.to(torch.int64), want)
    del want
    # 2. nothing outside the visited pixels, finite inside
    assert bool(torch.isfinite(mean).all()) and float(mean[st.count == 0].abs().max()) == 0.0
    assert float(mean.min()) >= 0.0 and float(mean.max()) > 0.0
    # 3. the far corner (largest offsets): pixels only the corner windows cover hold the mean over the 4 seasons of that window's
    #    forward, and the unbiased std over them
    x0, y0 = xs[-1], ys[-1]
    outs = []
    with torch.no_grad():
        for s in range(4):
            outs.append(m({"input": raster(x0, y0, s, PS).contiguous()}, padding=False)["popdensemap"][0])
    outs = torch.stack(outs)
    only = (torch.from_numpy(rc[x0:x0 + PS] == 1).cuda().view(-1, 1) & torch.from_numpy(cc[y0:y0 + PS] == 1).cuda().view(1, -1))
    only[:OV] = False; only[-OV:] = False; only[:, :OV] = False; only[:, -OV:] = False
    assert int(only.sum()) > 10000
    got = mean[x0:x0 + PS, y0:y0 + PS][only]
    torch.testing.assert_close(got, outs.mean(0)[only], rtol=1e-5, atol=1e-6)
    # (the stitcher's std is sqrt of a cancelling fp32 difference, run_eval.py:143: compared where it is well conditioned)
    sd_ref, sd_got = outs.std(0)[only], std[x0:x0 + PS, y0:y0 + PS][only]
    ok = torch.isfinite(sd_got) & (sd_ref > 2e-2 * outs.mean(0)[only].clamp_min(1e-3))
    assert float(ok.float().mean()) > 0.5
    torch.testing.assert_close(sd_got[ok], sd_ref[ok], rtol=2e-2, atol=1e-4)
    # 4. census: segment sums over 1,280 block regions (+ an unlabelled margin) == masked sum of the map (fp64)
    ii = torch.arange(H, device="cuda").view(-1, 1) // 360
    jj = torch.arange(W, device="cuda").view(1, -1) // 361
    boundary = (ii * 64 + jj).to(torch.int32)
    boundary[:, -100:] = -1
    nreg = int(boundary.max().item()) + 1
    sums, counts = E.census_sums(mean, boundary, nreg, want_counts=True)
    total = mean[boundary >= 0].double().sum().item()
    assert abs(sums.sum().item() - total) <= 1e-9 * total
    assert int(counts.sum().item()) == int((boundary >= 0).sum().item())
    # one region re-summed directly
    rid = 64 * 7 + 20
    direct = mean[boundary == rid].double().sum().item()
    assert abs(sums[rid].item() - direct) <= 1e-9 * max(direct, 1.0)
    # 5. dasymetric adjustment at this size: every region with a non-zero prediction then sums to its census count
    pop = torch.rand(nreg, device="cuda") * 1000 + 1
    adj = E.adjust_map_to_census(mean.clone(), boundary, torch.arange(nreg), pop)
    s2 = E.census_sums(adj, boundary, nreg)
    nz = sums > 0
    torch.testing.assert_close(s2[nz].float(), pop[nz], rtol=3e-5, atol=1e-3)


# ---- 1/8 crop: two ranks (windows sharded, band reduce-scatter form) == one process -----------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _crop_rank(rank, world, port, q):
    import torch.distributed as dist
    from popcorn_amd import eval as E
    from popcorn_amd.distributed import FlatReducer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    m = _model()
    h, w = H // 2, W // 4
    maps = E.evaluate_raster([m], SyntheticRaster(h, w), PS, OV, True, reducer=FlatReducer(), rank=rank)
    torch.cuda.synchronize()
    if rank == 0:
        out = []
        for t in maps:
            nan = int(torch.isnan(t).sum())
            t = t.nan_to_num(0.0)
            out.append((float(t.double().sum()), float((t.double() ** 2).sum()), t[::53, ::59].cpu().numpy(), nan))
        q.put(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _launch(world):
    from tests.test_gpu_dp import _get
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_crop_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = _get(q, procs, timeout=900)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    return out


def test_eighth_crop_two_ranks_equal_one_process():
    one = _launch(1)
    two = _launch(2)
    for (s1, q1, a, n1), (s2, q2, b, n2), name in zip(one, two, ("mean", "std", "scale mean", "scale std")):
        # a std pixel whose cancelling difference lands on either side of zero is NaN in one run and ~0 in the other (the
        # reference's sqrt of a negative, run_eval.py:143): at most a handful of the 20 M pixels
        assert abs(n1 - n2) <= 20, (name, n1, n2)
        # the std maps are square roots of cancelling fp32 differences of the accumulated sums (run_eval.py:143): the two-rank sum
        # order moves them by a few 1e-6 relative; the mean maps are plain sums
        tol = 2e-5 if "std" in name else 1e-6
        assert abs(s1 - s2) <= tol * max(abs(s1), 1.0), (name, s1, s2)
        assert abs(q1 - q2) <= tol * max(abs(q1), 1.0), (name, q1, q2)
        np.testing.assert_allclose(b, a, rtol=1e-3 if "std" in name else 2e-5, atol=2e-4 if "std" in name else 2e-6, err_msg=name)
