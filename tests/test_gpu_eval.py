"""Census aggregation, dasymetric adjustment and sliding-window stitching on the GPU vs the oracle's restatement of
the reference loops; plus size-independent properties at a larger raster (checksum of region sums, idempotence of the
adjustment, sharded == unsharded stitching)."""
import os

import numpy as np
import pytest
import torch

from oracle import popcorn_oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _regions(h, w, n, seed):
    """blocky census map with ids 0..n-1 plus a -1 (outside) band; returns boundary (int), bboxes"""
    rng = np.random.default_rng(seed)
    gy, gx = int(np.ceil(np.sqrt(n))), int(np.ceil(n / np.ceil(np.sqrt(n))))
    ys = np.minimum((np.arange(h) * gy) // h, gy - 1)
    xs = np.minimum((np.arange(w) * gx) // w, gx - 1)
    b = (ys[:, None] * gx + xs[None, :]).astype(np.int64)
    b[b >= n] = -1
    b[:2, :] = -1
    bbox = []
    for i in range(n):
        yy, xx = np.where(b == i)
        bbox.append((int(yy.min()), int(yy.max()) + 1, int(xx.min()), int(xx.max()) + 1) if len(yy) else (0, 0, 0, 0))
    return b, bbox


@pytest.mark.parametrize("shape,nreg", [((64, 80), 7), ((131, 77), 23), ((300, 260), 5000)])
def test_census_sum_and_adjust_vs_reference_loop(shape, nreg):
    from popcorn_amd import eval as E
    h, w = shape
    b, bbox = _regions(h, w, nreg, 1)
    rng = np.random.default_rng(2)
    pred = torch.from_numpy(rng.random((h, w)).astype(np.float32) * 3)
    idx = [i for i in range(nreg) if bbox[i][1] > 0]
    bb = [bbox[i] for i in idx]
    pop = torch.from_numpy(rng.random(len(idx)).astype(np.float32) * 100)
    bt = torch.from_numpy(b.astype(np.float32))
    ref = O.convert_popmap_to_census_loop(pred, bt, idx, bb)
    cp, cg = E.convert_popmap_to_census(pred.cuda(), torch.from_numpy(b).cuda(), idx, pop)
    torch.testing.assert_close(cp.cpu(), ref, rtol=2e-6, atol=1e-5)
    assert torch.equal(cg.cpu(), pop)
    sums, counts = E.census_sums(pred.cuda(), torch.from_numpy(b.astype(np.int32)).cuda(), nreg, want_counts=True)
    assert np.array_equal(counts.cpu().numpy(), np.bincount(b[b >= 0], minlength=nreg))       # index path: exact
    ref_adj = O.adjust_map_to_census_loop(pred, bt, idx, bb, pop)
    adj = E.adjust_map_to_census(pred.clone().cuda(), torch.from_numpy(b).cuda(), idx, pop)
    torch.testing.assert_close(adj.cpu(), ref_adj, rtol=3e-6, atol=1e-6)


@pytest.mark.parametrize("name", ["a", "b"])
def test_census_kernels_vs_reference_fixture_g9(name):
    """csrc/census.hip against outputs of the reference's own convert_popmap_to_census / adjust_map_to_census
    (fixture g9: the real functions behind a fake rasterio.open; data/PopulationDataset.py:675-852).  The reference sums
    each region in fp32 inside its bbox, the kernel accumulates fp64 over the whole raster in one pass: region sums
    agree to fp32 round-off of the reference's own reduction (3e-6), the index path (which pixels belong to a region,
    which regions are skipped) exactly."""
    from popcorn_amd import eval as E
    g = np.load(os.path.join(G, "g9_census.npz"))
    pred = torch.from_numpy(g[f"{name}/pred"]).cuda()
    boundary = torch.from_numpy(g[f"{name}/boundary"]).cuda()
    idx = g[f"{name}/census_idx"].tolist()
    pop = g[f"{name}/census_pop"]
    cp, cg = E.convert_popmap_to_census(pred, boundary, idx, pop)
    ref = g[f"{name}/census_pred"]
    np.testing.assert_allclose(cp.cpu().numpy(), ref, rtol=3e-6, atol=1e-5)
    assert np.array_equal(cp.cpu().numpy() == 0, ref == 0)            # absent ids / all-zero regions: exactly 0, as in the reference
    assert np.array_equal(cg.cpu().numpy(), g[f"{name}/census_gt"])
    adj = E.adjust_map_to_census(pred.clone(), boundary, idx, pop)
    radj = g[f"{name}/adjusted"]
    np.testing.assert_allclose(adj.cpu().numpy(), radj, rtol=4e-6, atol=1e-6)
    assert np.array_equal(adj.cpu().numpy() == g[f"{name}/pred"], radj == g[f"{name}/pred"])   # same pixels left untouched
    cp2, _ = E.convert_popmap_to_census(adj, boundary, idx, pop)
    np.testing.assert_allclose(cp2.cpu().numpy(), g[f"{name}/census_pred_adjusted"], rtol=1e-5, atol=1e-4)


def test_select_normalize_vs_reference_fixture_g10():
    """pc_select_normalize against the reference's apply_transformations_and_normalize(transform=None) output."""
    from popcorn_amd import ops
    from popcorn_amd.data import stats
    g = np.load(os.path.join(G, "g10_transform.npz"))
    raw = torch.cat([torch.from_numpy(g["pipe/S2"]), torch.from_numpy(g["pipe/S1"])], 1).contiguous().cuda()
    x = ops.select_normalize(raw, (0, 1, 2, 3, 4, 5), stats.MEAN6, stats.STD6)
    np.testing.assert_allclose(x.cpu().numpy(), g["pipe/none/input"], rtol=1e-6, atol=1e-6)


def test_census_properties_large_raster():
    """BASELINE-scale raster (2048 x 4096, 20k regions): checksum of checksums + idempotence."""
    from popcorn_amd import eval as E
    h, w, nreg = 2048, 4096, 20000
    g = torch.Generator(device="cuda").manual_seed(3)
    boundary = torch.randint(-1, nreg, (h, w), generator=g, device="cuda", dtype=torch.int32)
    pred = torch.rand(h, w, generator=g, device="cuda")
    sums = E.census_sums(pred, boundary, nreg)
    total_in = pred[boundary >= 0].double().sum().item()
    assert abs(sums.sum().item() - total_in) < 1e-9 * total_in
    pop = torch.rand(nreg, generator=g, device="cuda") * 50 + 1
    idx = torch.arange(nreg)
    adj = E.adjust_map_to_census(pred.clone(), boundary, idx, pop)
    sums2 = E.census_sums(adj, boundary, nreg)
    torch.testing.assert_close(sums2.float(), pop, rtol=2e-5, atol=1e-4)            # every region now sums to its census
    adj2 = E.adjust_map_to_census(adj.clone(), boundary, idx, pop)
    torch.testing.assert_close(adj2, adj, rtol=2e-5, atol=0)                        # idempotent
    assert torch.equal(adj[boundary < 0], pred[boundary < 0])                       # outside pixels untouched


def test_stitcher_vs_reference_loop():
    from popcorn_amd import eval as E
    h, w, ips, ov, M = 150, 170, 64, 8, 3
    idx = E.get_patch_indices(h, w, ips, ov, False)
    g = torch.Generator().manual_seed(4)
    wins = []
    st = E.Stitcher(h, w, "cuda")
    for x, y, s in idx.tolist():
        pd = torch.rand(M, ips, ips, generator=g)
        sc = torch.rand(M, ips, ips, generator=g)
        wins.append((x, y, pd, sc))
        st.add_window(x, y, pd.cuda(), sc.cuda(), ov)
    out, out_sq, sc_m, sc_sq = st.finalize()
    r_out, r_sq, r_sc, r_scsq, r_cnt = O.stitch_loop(h, w, wins, ips, ov)
    assert torch.equal(st.count.cpu(), r_cnt)
    torch.testing.assert_close(out.cpu(), r_out, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(sc_m.cpu(), r_sc, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(out_sq.cpu(), r_sq, rtol=1e-3, atol=2e-4, equal_nan=True)     # sqrt of a cancelling difference
    torch.testing.assert_close(sc_sq.cpu(), r_scsq, rtol=1e-3, atol=2e-4, equal_nan=True)


def test_evaluate_raster_sharded_equals_unsharded():
    """Window sharding (what each rank of an N-GPU evaluation does) reproduces the single-process maps: run the two
    halves of the window list into two stitchers, add them, finalize."""
    from popcorn_amd import eval as E
    from popcorn_amd.distributed import shard_indices
    from popcorn_amd.model import POPCORN
    torch.manual_seed(1600)
    m = POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
    h, w, ps, ov = 200, 232, 96, 16
    g = torch.Generator().manual_seed(5)
    raster = torch.randn(1, 6, h, w, generator=g).cuda()
    out, out_sq, sc, sc_sq = E.evaluate_raster([m], raster, patchsize=ps, overlap=ov)
    idx = E.get_patch_indices(h, w, ps, ov, False)
    parts = []
    for r in range(2):
        st = E.Stitcher(h, w, "cuda")
        for i in shard_indices(idx.shape[0], r, 2):
            x, y, s = idx[i].tolist()
            with torch.no_grad():
                o = m({"input": raster[s:s + 1, :, x:x + ps, y:y + ps].contiguous()}, padding=False)
            st.add_window(x, y, o["popdensemap"], o["scale"], ov)
        parts.append(st)
    tot = E.Stitcher(h, w, "cuda")
    for name in ("out", "out_sq", "scale", "scale_sq"):
        getattr(tot, name).copy_(getattr(parts[0], name) + getattr(parts[1], name))
    tot.count.copy_(parts[0].count + parts[1].count)
    o2, o2sq, s2, s2sq = tot.finalize()
    torch.testing.assert_close(o2, out, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(s2, sc, rtol=1e-6, atol=1e-7)
    # every pixel further than `ov` from the border is covered at least once
    assert (tot.count[ov:h - ov, ov:w - ov] >= 1).all()
    # and the stitched map equals a direct full-raster forward on interior pixels far from window seams is NOT expected
    # (receptive field 20 px < overlap 16 would leak) -- the reference has the same property with 128 px overlap.


def test_stitcher_and_census_vs_the_references_own_test_target_g12():
    """``pc_stitch_accumulate`` / ``pc_stitch_finalize`` + the census kernels against the maps the reference's OWN
    ``Trainer.test_target`` (run_eval.py:71-203) produced for the same windows (fixture g12: real patch grid, interior mask,
    ensemble sums, averaging / unbiased std where count > 1, plain sums where a pixel was visited once, census conversion,
    metrics, dasymetric adjustment).  The window inputs go through the HIP normalisation (``pc_select_normalize``)."""
    from popcorn_amd import eval as E
    from popcorn_amd import ops
    from popcorn_amd.data import stats
    from popcorn_amd.utils.metrics import get_test_metrics
    from tests.g12_case import CASES, load_case
    for name in CASES:
        c = load_case(name, lambda raw: ops.select_normalize(raw.cuda(), (0, 1, 2, 3, 4, 5), stats.MEAN6, stats.STD6))
        assert torch.equal(E.get_patch_indices(c["h"], c["w"], c["ips"], c["ov"], c["fourseasons"]),
                           torch.from_numpy(c["window_list"]))
        st = E.Stitcher(c["h"], c["w"], "cuda")
        for x, y, pd, sc in c["windows"]:
            st.add_window(x, y, pd, sc, c["ov"])
        out, out_sq, sc_m, sc_sq = st.finalize()
        r = c["ref"]
        assert torch.equal(st.count.cpu(), r["count"]), name
        torch.testing.assert_close(out.cpu(), r["map"], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(sc_m.cpu(), r["scale"], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(out_sq.cpu(), r["std"], rtol=1e-3, atol=3e-3, equal_nan=True)
        torch.testing.assert_close(sc_sq.cpu(), r["scale_std"], rtol=1e-3, atol=3e-3, equal_nan=True)
        # census conversion + metrics + adjustment on the reference's stitched map
        bnd = r["boundary"].cuda()
        cp, gt = E.convert_popmap_to_census(r["map"].cuda(), bnd, c["census_idx"], c["census_pop"])
        m = get_test_metrics(cp, gt, tag="MainCensus_uga_coarse")
        for k, v in m.items():
            ref = c["metrics"][k]
            assert abs(float(v) - ref) <= 5e-5 * max(1.0, abs(ref)), (name, k, float(v), ref)
        adj = E.adjust_map_to_census(r["map"].cuda().clone(), bnd, c["census_idx"], c["census_pop"])
        torch.testing.assert_close(adj.cpu(), r["adjusted"], rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("hw", [(172, 204), (90, 86), (256, 192)])
def test_inference_forward_with_aligned_rows_and_composed_up_vs_plain_and_oracle(hw, monkeypatch):
    """The inference path (eval.py) allocates activations with 16-byte aligned rows (L.padded_rows) and takes the composed Up
    convolution at any even width.  (1) With the composed path switched off, aligned rows only change which LOADER a layer takes,
    not a single arithmetic operation: bit-identical maps.  (2) With it on (the default) the maps agree to fp32 re-association
    (<= 2e-5) and with the oracle at the forward bound of this suite.  Geometries whose padded levels are not multiples of 4 / 32
    (the building extractor pads by 14: 200 x 232 -> 100 x 116 -> 50 x 58, 118 x 114 -> 59 x 57 -> 29 x 28, ...)."""
    from popcorn_amd import _lib as L
    from popcorn_amd import engine as E
    from popcorn_amd.data.synthetic import make_raw_batch
    from popcorn_amd.model import POPCORN
    H, W = hw
    torch.manual_seed(1600)
    m = POPCORN(6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda().eval()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x = O.select_normalize(make_raw_batch(1, H, W, seed=5, region="full")["raw"])

    def run(aligned):
        with torch.no_grad(), L.padded_rows(aligned):
            o = m({"input": x.cuda()}, padding=False)
        return o["popdensemap"].cpu(), o["scale"].cpu()

    monkeypatch.setattr(E, "COMPOSED_UP", False)
    plain = run(False)
    rows = run(True)
    assert torch.equal(plain[0], rows[0]) and torch.equal(plain[1], rows[1])
    monkeypatch.setattr(E, "COMPOSED_UP", True)
    comp = run(True)
    ref = O.popcorn_forward(sd, {"input": x}, padding=False, sparse=False)
    for got, base, want in ((comp[0], plain[0], ref["popdensemap"]), (comp[1], plain[1], ref["scale"])):
        scale = max(want.abs().max().item(), 1e-6)
        assert (got - base).abs().max().item() <= 2e-5 * scale
        assert (got - want).abs().max().item() <= 1e-4 * scale


def test_batched_visit_counts_equal_the_per_window_form():
    """Stitcher.add_counts_only (difference array + two prefix sums for all foreign windows at once) == add_count_only per window,
    including windows that stick out of the raster and windows with an empty interior."""
    from popcorn_amd import eval as E
    h, w, ps, ov = 301, 421, 128, 16
    idx = E.get_patch_indices(h, w, ps, ov, fourseasons=True)
    wins = [(int(r[0]), int(r[1])) for r in idx] + [(290, 410), (0, 400), (h - 20, 0)]
    a = E.Stitcher(h, w, "cuda")
    b = E.Stitcher(h, w, "cuda")
    for x, y in wins:
        a.add_count_only(x, y, 3, ps, ov)
    b.add_counts_only(wins, 3, ps, ov)
    assert a.count.dtype == b.count.dtype == torch.int16
    assert torch.equal(a.count, b.count) and int(a.count.max()) > 3
