"""BASELINE config 5 at its real window size: 2048 x 2048 sliding windows (utils/constants.py:12-13), the ``run_eval``
entry point on a raster larger than one window, and window sharding across ranks through ``Stitcher.all_reduce``
(run_eval.py:84-154; data/PopulationDataset.py:294-334,656-672).

The CPU oracle runs one 2048 x 2048 forward in a few seconds; the stitched-map check feeds ``O.stitch_loop`` (the restated
accumulation of run_eval.py:120-154) with per-window oracle forwards on a cropped sub-raster."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import popcorn_oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-6)


@pytest.fixture(scope="module")
def pair():
    from popcorn_amd.model import POPCORN
    torch.manual_seed(1600)
    m = POPCORN(input_channels=6, feature_extractor="DDA", occupancymodel=True, pretrained=True, biasinit=0.9407,
                sentinelbuildings=True).cuda().eval()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    return m, sd


def test_one_2048_window_vs_oracle(pair):
    """The eval-style call of run_eval.py:109 (``model(sample, padding=False)``, dense head, no admin_mask) on ONE full-size
    inference window: popdensemap / scale / building score <= 1e-4 relative against the CPU oracle."""
    m, sd = pair
    ps = 2048
    g = torch.Generator().manual_seed(50)
    x = torch.randn(1, 6, ps, ps, generator=g)
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    with torch.no_grad():
        ref = O.popcorn_forward(sd, {"input": x.clone()}, padding=False)
        inp = {"input": x.cuda()}
        out = m(inp, padding=False)
    torch.cuda.synchronize()
    assert tuple(out["popdensemap"].shape) == (1, ps, ps)
    assert _rel(out["popdensemap"].cpu(), ref["popdensemap"]) < 1e-4
    assert _rel(out["scale"].cpu(), ref["scale"]) < 1e-4
    assert abs(out["popcount"].item() - ref["popcount"].item()) <= 1e-4 * abs(ref["popcount"].item())
    # interior checksum: the part of the window the stitcher keeps (PopulationDataset.py:656-672)
    a = out["popdensemap"][0, 128:-128, 128:-128].double().sum().item()
    b = ref["popdensemap"][0, 128:-128, 128:-128].double().sum().item()
    assert abs(a - b) <= 1e-5 * abs(b)


def test_run_eval_cli_2048_windows_vs_oracle_stitch(pair, capsys):
    """``run_eval`` end to end (popcorn_amd/cli.py:run_eval <-> run_eval.py:71-203) on a 2304 x 2560 raster walked with
    2048-px windows / 128-px overlap: the window grid is the reference's (4 windows: origin, bottom, right, corner), the
    stitched mean map equals ``O.stitch_loop`` fed with per-window ORACLE forwards on a cropped sub-raster, the census
    metrics printed by the CLI equal metrics recomputed from the oracle-side census sums."""
    from popcorn_amd import cli, eval as E
    from popcorn_amd.data.dataset import SyntheticTestRaster
    from popcorn_amd.utils.metrics import get_test_metrics
    m, sd = pair
    h, w = 2304, 2560
    res = cli.run_eval(("-S2 -NIR -S1 -occmodel -senbuilds -pret --biasinit 0.9407 --raster_hw %d %d --seed 1600" % (h, w)).split())
    printed = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert set(printed) == set(res)
    idx = E.get_patch_indices(h, w, 2048, 128, False)
    assert idx.shape[0] == 4                                    # bottom row, right column, corner + one regular window
    assert np.array_equal(idx.numpy(), O.get_patch_indices(h, w, 2048, 128, False).numpy())
    # the CLI seeds the model with --seed and builds the raster from the eval default seed: rebuild both here
    data = SyntheticTestRaster(h, w, seasons=1, device="cuda")
    out, out_std, scale, scale_std = E.evaluate_raster([m], data.raster, 2048, 128, False)
    cp, cg = E.convert_popmap_to_census(out, data.boundary, data.census_idx, data.census_pop)
    tm = get_test_metrics(cp, cg, tag="MainCensus_synthetic_fine")
    for k, v in tm.items():
        assert abs(float(v) - res[k]) <= 1e-5 * max(1.0, abs(res[k])), k
    # oracle stitch on a sub-raster: windows of 512 with overlap 32 over a 704 x 640 crop (same code path, CPU-sized)
    ch, cw, ps, ov = 704, 640, 512, 32
    crop = data.raster[:, :, :ch, :cw].contiguous()
    o2, o2std, s2, s2std = E.evaluate_raster([m], crop, ps, ov, False)
    wins = []
    xc = crop.cpu()
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    with torch.no_grad():
        for x0, y0, s in E.get_patch_indices(ch, cw, ps, ov, False).tolist():
            r = O.popcorn_forward(sd, {"input": xc[s:s + 1, :, x0:x0 + ps, y0:y0 + ps].contiguous()}, padding=False)
            wins.append((x0, y0, r["popdensemap"], r["scale"]))
    r_out, r_sq, r_sc, r_scsq, r_cnt = O.stitch_loop(ch, cw, wins, ps, ov)
    assert _rel(o2.cpu(), r_out) < 1e-4
    assert _rel(s2.cpu(), r_sc) < 1e-4
    # census totals of the full-size run are consistent with the map (segment sum = checksum of checksums)
    inside = data.boundary > 0
    assert abs(cp.double().sum().item() - out[inside].double().sum().item()) <= 1e-6 * out[inside].double().sum().item()


# ---- window sharding across ranks --------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _eval_rank(rank, world, port, q, band=True):
    import torch.distributed as dist
    from popcorn_amd import eval as E
    from popcorn_amd.distributed import FlatReducer
    from popcorn_amd.model import POPCORN
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    torch.manual_seed(1600)
    ms = []
    for j in range(2):                              # a 2-member ensemble (the second with a perturbed head)
        m = POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda().eval()
        with torch.no_grad():
            m.head[6].bias.add_(0.05 * j)
        ms.append(m)
    g = torch.Generator().manual_seed(9)
    raster = torch.randn(2, 6, 300, 420, generator=g).cuda()       # two "seasons"
    maps = E.evaluate_raster(ms, raster, patchsize=128, overlap=16, fourseasons=False, reducer=FlatReducer(), rank=rank,
                              band_reduce=band)
    torch.cuda.synchronize()
    if rank == 0:
        q.put([t.cpu().numpy() for t in maps])
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _launch(world, band=True):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eval_rank, args=(r, world, port, q, band)) for r in range(world)]
    for p in procs:
        p.start()
    from tests.test_gpu_dp import _get
    out = _get(q, procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return out


@pytest.mark.parametrize("band,world", [(True, 2), (False, 2), (True, 4)])
def test_n_rank_evaluate_raster_equals_single_process(band, world):
    """Two processes (gloo group, both on the one GPU of the test box) shard the window list round-robin and accumulate into
    their own device stitchers.  band=True: the visit count comes from the window list (no count collective), the planes are
    reduced by row band, every rank finalises its band, the bands are gathered; band=False: the all-reduce form.  Either way the
    finalised maps equal the single-process run (300 rows over 2 ranks: bands of 150)."""
    one = _launch(1)
    two = _launch(world, band)           # (4 ranks: 30 windows -> shards of 8, 8, 7, 7; 300 rows -> bands of 75)
    for a, b, name, mean in zip(one, two, ("mean", "std", "scale mean", "scale std"), (None, one[0], None, one[2])):
        fin = np.isfinite(a)
        assert np.array_equal(fin, np.isfinite(b)), name
        if mean is None:
            np.testing.assert_allclose(b[fin], a[fin], rtol=2e-5, atol=2e-6, err_msg=name)
            continue
        # std = sqrt(sum x^2 / n - mean^2) (run_eval.py:140-154): what a different summation order across ranks moves is the
        # VARIANCE, by a few ulp of mean^2 -- a near-zero std then moves by sqrt of that (4e-6 absolute / 3.5e-4 relative at
        # std 1e-2, world 4).  The bound is therefore on the variance: 8 ulp of (mean^2 + var), fp32.
        va, vb, m2 = a[fin].astype(np.float64) ** 2, b[fin].astype(np.float64) ** 2, mean[fin].astype(np.float64) ** 2
        bound = 8 * 2.0 ** -23 * (m2 + va) + 1e-12
        assert np.all(np.abs(vb - va) <= bound), (name, float(np.max(np.abs(vb - va) / bound)))


def _band_census_rank(rank, world, port, q):
    """evaluate_raster(gather=False) + census_sums_sharded on a raster with an ODD width and an odd band height: rank 1's band
    starts at a byte offset that is not a multiple of 16 (pc_census_sum refused it and the other rank then hung in
    the all-reduce)."""
    import torch.distributed as dist
    from popcorn_amd import eval as E
    from popcorn_amd.distributed import FlatReducer
    from popcorn_amd.model import POPCORN
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    torch.manual_seed(1600)
    m = POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda().eval()
    g = torch.Generator().manual_seed(11)
    h, w = 301, 421
    raster = torch.randn(1, 6, h, w, generator=g).cuda()
    yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    boundary = ((yy // 43) * 10 + xx // 47).to(torch.int32).cuda()
    boundary[:3] = -1
    red = FlatReducer()
    if world > 1:
        st = E.evaluate_raster([m], raster, patchsize=128, overlap=16, reducer=red, rank=rank, band_reduce=True, gather=False)
        r0, r1 = st.band
        assert (r0 * w) % 4 != 0 or rank == 0
        # the emulated reduce-scatter poisons the rows of the other ranks' bands: nothing below may read them
        if rank == 1:
            assert torch.isnan(st.out[:r0]).all()
        sums = E.census_sums_sharded(st, boundary, 80, red)
        band = st.out[r0:r1].clone()
        maps = st.gather_bands(red)
        assert torch.equal(maps[0][r0:r1], band)
    else:
        maps = E.evaluate_raster([m], raster, patchsize=128, overlap=16, reducer=red, rank=0)
        sums = E.census_sums(maps[0].contiguous(), boundary, 80)
    torch.cuda.synchronize()
    if rank == 0:
        q.put([sums.cpu().numpy(), maps[0].cpu().numpy()])
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_two_rank_banded_census_sums_with_odd_width_and_poisoned_foreign_bands():
    """301 x 421 raster over 2 ranks: bands of 151 rows, rank 1's band starts at element 151 * 421 (odd -> 4-byte aligned only).
    Region sums from the distributed bands == the single-process sums; the gathered mean map == the single-process map; the
    reduce-scatter emulation on gloo leaves NaN in every row a rank does not own, so a stale-row read cannot pass."""
    from tests.test_gpu_dp import _get
    outs = []
    for world in (1, 2):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_band_census_rank, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        outs.append(_get(q, procs))
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    (s1, m1), (s2, m2) = outs
    assert np.isfinite(s2).all()
    np.testing.assert_allclose(s2, s1, rtol=1e-6, atol=1e-6)
    fin = np.isfinite(m1)
    assert np.array_equal(fin, np.isfinite(m2))
    np.testing.assert_allclose(m2[fin], m1[fin], rtol=2e-5, atol=2e-6)
