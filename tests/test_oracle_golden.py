"""Pin the CPU oracle (oracle/popcorn_oracle.py) against the golden vectors produced by importing the
reference itself (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import popcorn_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def sd():
    return O.load_golden_state(G)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def test_state_dict_keys(sd):
    lines = [l.rstrip("\n").split("\t") for l in open(os.path.join(G, "g1_state_dict_keys.txt"))]
    keys = [l[0] for l in lines]
    assert len(keys) == 324
    for k, shp, kind in lines:
        if k.endswith("num_batches_tracked"):
            continue
        assert k in sd, k
        assert ",".join(map(str, sd[k].shape)) == shp
    names = O.trainable_names(sd)
    assert len(names) == 56
    assert sum(sd[n].numel() for n in names) == 39298


def test_layers_g3(sd):
    g = np.load(os.path.join(G, "g3_layers.npz"))
    x = torch.from_numpy(g["input"])
    with torch.no_grad():
        for si, stream in enumerate(("sar_stream", "optical_stream")):
            xs = x[:, :2] if si == 0 else x[:, 2:]
            acts = O.unet_stream(sd, "unetmodel." + stream, xs, return_all=True)
            for mine, ref in (("inc", "inc.conv.conv.5"), ("down1", "down_seq.down1.mpconv.1.conv.5"),
                              ("down2", "down_seq.down2.mpconv.1.conv.5"), ("up2", "up_seq.up2.conv.conv.5"),
                              ("up1", "up_seq.up1.conv.conv.5")):
                np.testing.assert_allclose(acts[mine].numpy(), g[f"act/{stream}.{ref}"], rtol=0, atol=1e-5)
        feats = O.dualstream_features(sd, "unetmodel", x)
        ls, lo, lf = O.dualstream_logits(sd, "unetmodel", x)
    np.testing.assert_allclose(feats.numpy(), g["features"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(ls.numpy(), g["logits_sar"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(lo.numpy(), g["logits_optical"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(lf.numpy(), g["logits_fusion"], rtol=0, atol=1e-5)


@pytest.mark.parametrize("name", ["b2_100", "b1_131x77", "b2_64"])
@pytest.mark.parametrize("padding", [True, False])
@pytest.mark.parametrize("sparse", [True, False])
def test_forward_g2(sd, name, padding, sparse):
    g = np.load(os.path.join(G, "g2_forward.npz"))
    inp = {"input": torch.from_numpy(g[f"{name}/input"]), "admin_mask": torch.from_numpy(g[f"{name}/admin_mask"]),
           "census_idx": torch.from_numpy(g[f"{name}/census_idx"])}
    torch.manual_seed(1600)
    with torch.no_grad():
        o = O.popcorn_forward(sd, inp, padding=padding, sparse=sparse, return_features=True)
    tag = f"{name}/pad{int(padding)}_sp{int(sparse)}"
    assert tuple(o["features"].shape) == tuple(g[f"{tag}/feat_shape"])
    np.testing.assert_allclose(o["features"][:, :, ::7, ::5].numpy(), g[f"{tag}/feat_sample"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(inp["building_counts"].numpy(), g[f"{name}/building_counts"], rtol=0, atol=1e-6)
    assert rel_err(o["popdensemap"].numpy(), g[f"{tag}/popdensemap"]) < 1e-5
    assert rel_err(o["popcount"].numpy(), g[f"{tag}/popcount"]) < 1e-5
    assert o["scale"].shape == g[f"{tag}/scale"].shape          # sparse: Nsel is an index-path result -> exact
    assert rel_err(o["scale"].numpy(), g[f"{tag}/scale"]) < 1e-5


@pytest.mark.parametrize("name", ["b2_100", "b1_131x77"])
def test_forward_noadmin(sd, name):
    g = np.load(os.path.join(G, "g2_forward.npz"))
    inp = {"input": torch.from_numpy(g[f"{name}/input"])}
    with torch.no_grad():
        o = O.popcorn_forward(sd, inp, padding=False)
    assert rel_err(o["popcount"].numpy(), g[f"{name}/noadmin/popcount"]) < 1e-5
    assert rel_err(o["popdensemap"].numpy(), g[f"{name}/noadmin/popdensemap"]) < 1e-5


@pytest.mark.parametrize("name", ["b2_100", "b1_131x77", "b2_40x52"])
def test_mask_g4_bit_exact(name):
    g = np.load(os.path.join(G, "g4_mask.npz"))
    torch.manual_seed(1600)
    m = O.get_sparsity_mask(torch.from_numpy(g[f"{name}/building_counts"]), torch.from_numpy(g[f"{name}/admin_mask"]),
                            torch.from_numpy(g[f"{name}/census_idx"]))
    assert m.dtype == torch.bool
    assert np.array_equal(m.numpy(), g[f"{name}/mask"])


def test_train_step_g5(sd):
    g = np.load(os.path.join(G, "g5_train.npz"))
    sample = {"input": torch.from_numpy(g["input"]), "admin_mask": torch.from_numpy(g["admin_mask"]),
              "census_idx": torch.from_numpy(g["census_idx"]), "y": torch.from_numpy(g["y"])}
    params = dict(sd)
    state = {}
    traj = []
    for step in range(3):
        torch.manual_seed(1700 + step)
        loss, out, grads, aux = O.train_step_grads(params, dict(sample))
        traj.append(loss.item())
        if step == 0:
            assert set(grads.keys()) == set(g["step0/grad_names"].tolist())
            assert out["scale"].numel() == int(g["step0/nsel"])
            assert rel_err(out["popcount"].numpy(), g["step0/popcount"]) < 1e-5
            for n, gr in grads.items():
                ref = g["step0/grad/" + n]
                assert rel_err(gr.numpy(), ref) < 2e-4 or np.abs(gr.numpy() - ref).max() < 1e-6, n
            for k in [k for k in g.files if k.startswith("step0/lossdict/")]:
                kk = k[len("step0/lossdict/"):].replace("|", "/")
                assert abs(aux[kk] - float(g[k])) <= 1e-4 * max(1.0, abs(float(g[k]))), kk
        total, clipped = O.clip_grad_norm(grads, 0.01)
        if step == 0:
            assert abs(total.item() - float(g["step0/total_norm"])) < 1e-3 * float(g["step0/total_norm"])
        new = O.adam_step(params, clipped, state, lr=1e-4, weight_decay=1e-5)
        params.update(new)
        if step == 0:
            for n in grads:
                np.testing.assert_allclose(params[n].numpy(), g["step0/param_after/" + n], rtol=0, atol=2e-7)
    np.testing.assert_allclose(np.array(traj), g["loss_traj"], rtol=2e-5)
    for k in [k for k in g.files if k.startswith("step2/param_after/")]:
        np.testing.assert_allclose(params[k[len("step2/param_after/"):]].numpy(), g[k], rtol=0, atol=1e-6)


def test_loss_metrics_g6():
    g = np.load(os.path.join(G, "g6_loss_metrics.npz"))
    pred, y, scale = (torch.from_numpy(g[k]) for k in ("pred", "y", "scale"))
    for lname in ["l1_loss", "log_l1_loss", "mse_loss", "log_mse_loss"]:
        loss, aux = O.get_loss({"popcount": pred}, {"y": y}, scale=scale, loss=[lname], lam=[1.0],
                               scale_regularization=0.01, tag="weak")
        assert abs(loss.item() - float(g[f"get_loss/{lname}/loss"])) <= 1e-6 * max(1, abs(loss.item()))
        for k in [k for k in g.files if k.startswith(f"get_loss/{lname}/") and not k.endswith("/loss")]:
            kk = k[len(f"get_loss/{lname}/"):].replace("|", "/")
            assert abs(aux[kk] - float(g[k])) <= 1e-6 * max(1.0, abs(float(g[k]))), kk
    tm = O.get_test_metrics(pred, y, tag="coarse")
    for k, v in tm.items():
        ref = float(g["test_metrics/" + k.replace("/", "|")])
        assert abs(v.item() - ref) <= 1e-6 * max(1.0, abs(ref)), k


def test_census_sums_known_answer():
    rng = np.random.default_rng(5)
    b = rng.integers(0, 7, size=(50, 60)).astype(np.int32)
    p = rng.random((50, 60)).astype(np.float32)
    s = O.census_sums(p, b, 7)
    for i in range(7):
        assert abs(s[i] - p[b == i].astype(np.float64).sum()) < 1e-9


@pytest.mark.parametrize("ic", [2, 4])
def test_single_modality_g8(sd, ic):
    """S1-only / S2-only variants (popcorn.py:48-54,136-145): oracle vs reference fixture."""
    g = np.load(os.path.join(G, "g8_single_modality.npz"))
    sd2 = dict(sd)
    for k in g.files:
        if k.startswith(f"ic{ic}/head."):
            sd2[k[len(f"ic{ic}/"):]] = torch.from_numpy(g[k])
    sample = {"input": torch.from_numpy(g[f"ic{ic}/input"]), "admin_mask": torch.from_numpy(g[f"ic{ic}/admin_mask"]),
              "census_idx": torch.from_numpy(g[f"ic{ic}/census_idx"]), "y": torch.from_numpy(g[f"ic{ic}/y"])}
    torch.manual_seed(5)
    loss, out, grads, _ = O.train_step_grads(sd2, dict(sample))
    assert abs(loss.item() - float(g[f"ic{ic}/loss"])) < 1e-5
    assert rel_err(out["popdensemap"].numpy(), g[f"ic{ic}/popdensemap"]) < 1e-5
    ref_names = set(g[f"ic{ic}/grad_names"].tolist())
    assert {n for n, v in grads.items() if v is not None} == ref_names
    for n in ref_names:
        r = g[f"ic{ic}/grad/{n}"]
        assert np.abs(grads[n].numpy() - r).max() <= 2e-4 * max(np.abs(r).max(), 1e-3), n


def test_force_decisions_with_the_oracles_own_decisions_changes_nothing_and_foreign_ones_are_followed(sd):
    """ForceDecisions (the mode the GPU tests use to compare gradients under SHARED ReLU / arg-max decisions): (1) forcing the decisions
    the oracle took itself reproduces its gradients bit for bit, 0 flips; (2) forcing the decisions of a perturbed evaluation makes the
    fp32 oracle's gradients equal that evaluation's to rounding, although the two unforced gradient sets differ by a flipped tie."""
    g = np.load(os.path.join(G, "g5_train.npz"))
    sample = {"input": torch.from_numpy(g["input"]), "admin_mask": torch.from_numpy(g["admin_mask"]),
              "census_idx": torch.from_numpy(g["census_idx"]), "y": torch.from_numpy(g["y"])}
    torch.manual_seed(1700)
    with O.TieProbe() as probe:
        _, _, grads, _ = O.train_step_grads(sd, dict(sample))
    torch.manual_seed(1700)
    with O.ForceDecisions(probe.acts, probe.pools) as f:
        _, _, forced, _ = O.train_step_grads(sd, dict(sample))
    assert f.flips == {"relu": 0, "pool": 0} and f.i == len(probe.acts) and f.j == len(probe.pools)
    for n in grads:
        assert torch.equal(grads[n], forced[n]), n
    # a foreign decision set: flip the mask at the site with the smallest |pre-activation| of the first layer by handing in a map
    # whose value there has the other sign
    acts = [a.clone() for a in probe.acts]
    a0 = acts[0]
    pos = a0[a0 > 0]
    site = (a0 == pos.min()).nonzero()[0]
    acts[0][tuple(site)] = 0.0                      # "the other implementation" decided this unit is off
    torch.manual_seed(1700)
    with O.ForceDecisions(acts, probe.pools) as f2:
        _, _, forced2, _ = O.train_step_grads(sd, dict(sample))
    assert f2.flips["relu"] == 1
    # the forced evaluation's gradients are those of a network in which that unit is off: they differ from the unforced ones exactly
    # where that unit fed gradient (first layer of its stream), and are finite everywhere
    changed = [n for n in grads if not torch.equal(grads[n], forced2[n])]
    assert changed and all(torch.isfinite(forced2[n]).all() for n in forced2)
