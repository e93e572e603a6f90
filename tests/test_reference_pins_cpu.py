"""Host-side pieces pinned to outputs of the REFERENCE ITSELF (fixtures written by tests/golden/make_golden.py):

  g9_census.npz     the real Population_Dataset.convert_popmap_to_census / adjust_map_to_census
                    (data/PopulationDataset.py:675-852) behind a fake rasterio.open + a temporary census CSV
  g10_transform.npz the reference's augmentation classes (utils/transform.py:54-276) and
                    apply_transformations_and_normalize (utils/utils.py:105-214) with the trainer's transform set
                    (run_train.py:386-402); torchvision's five functionals restated from their published definitions
  g6_loss_metrics.npz  utils/losses.get_loss and utils/metrics.get_test_metrics

Checked here: the oracle's census loops, and the PRODUCT's host glue (popcorn_amd.utils.{transform,utils,losses,metrics}).
The HIP census kernels are checked against g9 in tests/test_gpu_eval.py.  CPU only."""
import os
import random

import numpy as np
import torch

from oracle import popcorn_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


def _census_case(g, name):
    pred = torch.from_numpy(g[f"{name}/pred"])
    boundary = torch.from_numpy(g[f"{name}/boundary"].astype(np.float32))
    idx = g[f"{name}/census_idx"].tolist()
    bbox = [tuple(int(v) for v in b) for b in g[f"{name}/census_bbox"]]
    pop = g[f"{name}/census_pop"]
    return pred, boundary, idx, bbox, pop


def test_oracle_census_loops_vs_reference_g9():
    g = np.load(os.path.join(G, "g9_census.npz"))
    for name in ("a", "b"):
        pred, boundary, idx, bbox, pop = _census_case(g, name)
        cp = O.convert_popmap_to_census_loop(pred, boundary, idx, bbox)
        assert np.array_equal(cp.numpy(), g[f"{name}/census_pred"]), name            # same fp32 reduction: bit-equal
        assert np.array_equal(g[f"{name}/census_gt"], pop.astype(np.float32))
        adj = O.adjust_map_to_census_loop(pred, boundary, idx, bbox, pop)
        assert np.array_equal(adj.numpy(), g[f"{name}/adjusted"]), name
        cp2 = O.convert_popmap_to_census_loop(adj, boundary, idx, bbox)
        assert np.array_equal(cp2.numpy(), g[f"{name}/census_pred_adjusted"])
        # the one-pass restatement (float64 segment sum) agrees with the reference's per-region fp32 sums
        s = O.census_sums(pred.numpy(), boundary.numpy(), max(idx) + 1)
        np.testing.assert_allclose(s[idx], g[f"{name}/census_pred"], rtol=3e-6, atol=1e-5)
        # regions that exist and have a non-zero prediction now sum to their census count
        ok = (g[f"{name}/census_pred"] > 0)
        np.testing.assert_allclose(g[f"{name}/census_pred_adjusted"][ok], pop[ok], rtol=2e-5)
        assert (~ok).any(), "fixture must contain an all-zero region and an absent id"


def _seeded(seed):
    torch.manual_seed(seed)
    random.seed(seed)


def test_augmentation_classes_vs_reference_g10():
    from popcorn_amd.utils import transform as T
    g = np.load(os.path.join(G, "g10_transform.npz"))
    s2, x6, mk = (torch.from_numpy(g[k]) for k in ("s2", "x6", "mask"))
    for seed in range(6):
        _seeded(seed)
        y = T.RandomBrightness(p=0.9, beta_limit=(0.666, 1.5))(s2.clone())
        assert np.array_equal(y.numpy(), g[f"brightness/{seed}"]), seed
        _seeded(seed)
        y = T.RandomGamma(p=0.9, gamma_limit=(0.6666, 1.5))(s2.clone())
        assert np.array_equal(y.numpy(), g[f"gamma/{seed}"]), seed
        _seeded(seed)
        y = T.RandomGamma(p=0.9, gamma_limit=(0.6666, 1.5))(s2[:, :3].clone())
        assert np.array_equal(y.numpy(), g[f"gamma3/{seed}"]), seed
        _seeded(seed)
        y = T.OwnCompose([T.RandomBrightness(p=0.9, beta_limit=(0.666, 1.5)),
                          T.RandomGamma(p=0.9, gamma_limit=(0.6666, 1.5))])(s2.clone())
        assert np.array_equal(y.numpy(), g[f"s2compose/{seed}"]), seed
        for cname, cls in (("vflip", T.RandomVerticalFlip), ("hflip", T.RandomHorizontalFlip)):
            for allsame in (True, False):
                _seeded(seed)
                a, b = cls(p=0.5, allsame=allsame)((x6.clone(), mk.clone()))
                assert np.array_equal(a.numpy(), g[f"{cname}/same{int(allsame)}/{seed}/x"]), (cname, allsame, seed)
                assert np.array_equal(b.numpy(), g[f"{cname}/same{int(allsame)}/{seed}/mask"])
        _seeded(seed)
        a, b = T.RandomRotationTransform(angles=[90, 180, 270], p=0.75)((x6.clone(), mk.clone()))
        assert np.array_equal(a.numpy(), g[f"rot/{seed}/x"]) and np.array_equal(b.numpy(), g[f"rot/{seed}/mask"])
    # some seeds must exercise both branches of every coin
    assert any(not np.array_equal(g[f"brightness/{s}"], g["s2"]) for s in range(6))
    assert any(g[f"rot/{s}/x"].shape != g["x6"].shape for s in range(6))


def test_apply_transformations_and_normalize_vs_reference_g10():
    from popcorn_amd.utils.transform import default_train_transform
    from popcorn_amd.utils.utils import apply_transformations_and_normalize, default_dataset_stats
    g = np.load(os.path.join(G, "g10_transform.npz"))
    s2, s1, admin = (torch.from_numpy(g[k]) for k in ("pipe/S2", "pipe/S1", "pipe/admin_mask"))
    tr = default_train_transform()
    st = default_dataset_stats()
    for seed in range(6):
        _seeded(seed + 40)
        r = apply_transformations_and_normalize({"S2": s2.clone(), "S1": s1.clone(), "admin_mask": admin.clone()}, tr, st)
        np.testing.assert_allclose(r["input"].numpy(), g[f"pipe/{seed}/input"], rtol=1e-6, atol=1e-6)
        assert np.array_equal(r["admin_mask"].numpy(), g[f"pipe/{seed}/admin_mask"]), seed
    r = apply_transformations_and_normalize({"S2": s2.clone(), "S1": s1.clone(), "admin_mask": admin.clone()}, None, st)
    np.testing.assert_allclose(r["input"].numpy(), g["pipe/none/input"], rtol=1e-6, atol=1e-6)
    # the oracle's band-select + normalise restatement (what tests/bench use as the checker of pc_select_normalize)
    raw = torch.cat([s2, s1], 1)
    np.testing.assert_allclose(O.select_normalize(raw, (0, 1, 2, 3, 4, 5)).numpy(), g["pipe/none/input"], rtol=1e-6, atol=1e-6)


def test_product_losses_and_metrics_vs_reference_g6():
    from popcorn_amd.utils.losses import get_loss
    from popcorn_amd.utils.metrics import get_test_metrics
    g = np.load(os.path.join(G, "g6_loss_metrics.npz"))
    pred, y, scale = (torch.from_numpy(g[k]) for k in ("pred", "y", "scale"))
    for lname in ["l1_loss", "log_l1_loss", "mse_loss", "log_mse_loss"]:
        loss, aux = get_loss({"popcount": pred.clone(), "popdensemap": torch.zeros(1, 2, 2), "scale": scale.clone()}, {"y": y},
                             scale=scale, loss=[lname], lam=[1.0], scale_regularization=0.01, tag="weak")
        assert abs(loss.item() - float(g[f"get_loss/{lname}/loss"])) <= 1e-6 * max(1, abs(loss.item()))
        ref_keys = [k for k in g.files if k.startswith(f"get_loss/{lname}/") and not k.endswith("/loss")]
        assert {k[len(f"get_loss/{lname}/"):].replace("|", "/") for k in ref_keys} == set(aux.keys())
        for k in ref_keys:
            kk = k[len(f"get_loss/{lname}/"):].replace("|", "/")
            assert abs(aux[kk] - float(g[k])) <= 1e-6 * max(1.0, abs(float(g[k]))), kk
    tm = get_test_metrics(pred, y, tag="coarse")
    ref_keys = [k for k in g.files if k.startswith("test_metrics/")]
    assert [k[len("test_metrics/"):].replace("|", "/") for k in ref_keys] == list(tm.keys())     # same keys, same order
    for k, v in tm.items():
        ref = float(g["test_metrics/" + k.replace("/", "|")])
        assert abs(v.item() - ref) <= 1e-6 * max(1.0, abs(ref)), k


def test_oracle_stitch_loop_vs_reference_g12():
    """``O.stitch_loop`` (and the oracle's patch grid, census loops and metrics behind it) against the maps the reference's own
    ``Trainer.test_target`` (run_eval.py:71-203) wrote for the same windows: fixture g12.  Case a: four seasons x three
    members (every visited pixel is averaged, unbiased std); case b: one member, one season (pixels visited once keep the
    SUM of squares in the std map -- run_eval.py:137,142 only touch count > 1); case c: two members, catch-up windows."""
    from tests.g12_case import CASES, load_case
    from popcorn_amd.utils.metrics import get_test_metrics
    for name in CASES:
        c = load_case(name, lambda raw: O.select_normalize(raw, (0, 1, 2, 3, 4, 5)))
        assert np.array_equal(O.get_patch_indices(c["h"], c["w"], c["ips"], c["ov"], c["fourseasons"]).numpy(), c["window_list"])
        out, out_sq, sc, sc_sq, cnt = O.stitch_loop(c["h"], c["w"], c["windows"], c["ips"], c["ov"])
        r = c["ref"]
        assert torch.equal(cnt, r["count"]), name
        torch.testing.assert_close(out, r["map"], rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(sc, r["scale"], rtol=1e-6, atol=1e-6)
        # sqrt of a cancelling difference of fp32 sums: absolute agreement only (NaN where the difference went negative)
        torch.testing.assert_close(out_sq, r["std"], rtol=1e-3, atol=2e-3, equal_nan=True)
        torch.testing.assert_close(sc_sq, r["scale_std"], rtol=1e-3, atol=2e-3, equal_nan=True)
        if name == "b":
            once = r["count"] == 1                     # visited once: left as plain sums (no division, no sqrt)
            assert once.any() and (r["count"] == 2).any() and (r["count"] == 0).any()
            torch.testing.assert_close(out_sq[once], (r["map"][once]) ** 2, rtol=1e-6, atol=1e-7)
        # census conversion + metrics + dasymetric adjustment on the stitched map, as test_target chains them
        bbox = []
        for cid in c["census_idx"]:
            xs, ys = torch.where(r["boundary"] == cid)
            bbox.append((int(xs.min()), int(xs.max()) + 1, int(ys.min()), int(ys.max()) + 1))
        bnd = r["boundary"].float()
        cp = O.convert_popmap_to_census_loop(r["map"], bnd, c["census_idx"], bbox)
        gt = torch.tensor(c["census_pop"], dtype=torch.float32)
        for fn, tag in ((O.get_test_metrics, "oracle"), (get_test_metrics, "product")):
            m = fn(cp, gt, tag="MainCensus_uga_coarse")
            for k, v in m.items():
                ref = c["metrics"][k]
                assert abs(float(v) - ref) <= 2e-5 * max(1.0, abs(ref)), (name, tag, k, float(v), ref)
        adj = O.adjust_map_to_census_loop(r["map"], bnd, c["census_idx"], bbox, c["census_pop"])
        torch.testing.assert_close(adj, r["adjusted"], rtol=1e-6, atol=1e-7)
