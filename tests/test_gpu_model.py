"""Model-level parity on the GPU: POPCORN.forward (HIP) vs the golden vectors produced by the reference itself
(g2 forward, g5 train step) and vs the CPU oracle on fresh seeded inputs.  Tolerances (north_star): forward
<= 1e-4 relative in fp32; mask / Nsel index paths bit-exact."""
import os

import numpy as np
import pytest
import random

import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.fixture(scope="module")
def model():
    from popcorn_amd.model import POPCORN
    torch.manual_seed(1600)
    m = POPCORN(input_channels=6, feature_extractor="DDA", occupancymodel=True, pretrained=True, biasinit=0.9407,
                sentinelbuildings=True)
    return m.cuda()


@pytest.mark.parametrize("name", ["b2_100", "b1_131x77", "b2_64"])
@pytest.mark.parametrize("padding", [True, False])
@pytest.mark.parametrize("sparse", [True, False])
def test_forward_vs_reference_golden(model, name, padding, sparse):
    g = np.load(os.path.join(G, "g2_forward.npz"))
    inp = {"input": torch.from_numpy(g[f"{name}/input"]).cuda(),
           "admin_mask": torch.from_numpy(g[f"{name}/admin_mask"]).cuda(),
           "census_idx": torch.from_numpy(g[f"{name}/census_idx"]).cuda()}
    model.eval()
    torch.manual_seed(1600)
    with torch.no_grad():
        o = model(inp, train=False, padding=padding, sparse=sparse)
    tag = f"{name}/pad{int(padding)}_sp{int(sparse)}"
    np.testing.assert_allclose(inp["building_counts"].cpu().numpy(), g[f"{name}/building_counts"], rtol=0, atol=1e-5)
    assert rel_err(o["popdensemap"].cpu().numpy(), g[f"{tag}/popdensemap"]) < 1e-4
    assert rel_err(o["popcount"].cpu().numpy(), g[f"{tag}/popcount"]) < 1e-4
    assert o["scale"].shape == g[f"{tag}/scale"].shape                  # Nsel: index path, exact
    assert rel_err(o["scale"].cpu().numpy(), g[f"{tag}/scale"]) < 1e-4
    # feature map (padded domain) through the engine directly
    from popcorn_amd.model.popcorn import pad_geometry
    H, W = inp["input"].shape[2:]
    pt, pb, pl, pr = pad_geometry(H, W, padding)
    feats, _ = model.engines()[0].forward(inp["input"], pt, pl, H + pt + pb, W + pl + pr)
    assert tuple(feats.shape) == tuple(g[f"{tag}/feat_shape"])
    np.testing.assert_allclose(feats[:, :, ::7, ::5].cpu().numpy(), g[f"{tag}/feat_sample"], rtol=0, atol=3e-5)
    assert abs(feats.double().sum().item() - float(g[f"{tag}/feat_sum64"])) < 1e-5 * abs(float(g[f"{tag}/feat_sum64"])) + 1e-2


@pytest.mark.parametrize("name", ["b2_100", "b1_131x77"])
def test_forward_noadmin_eval_call(model, name):
    g = np.load(os.path.join(G, "g2_forward.npz"))
    inp = {"input": torch.from_numpy(g[f"{name}/input"]).cuda()}
    with torch.no_grad():
        o = model(inp, padding=False)
    assert rel_err(o["popcount"].cpu().numpy(), g[f"{name}/noadmin/popcount"]) < 1e-4
    assert rel_err(o["popdensemap"].cpu().numpy(), g[f"{name}/noadmin/popdensemap"]) < 1e-4


def test_layer_activations_vs_reference_golden(model):
    """Per-layer bisecting aid (fixture g3): un-padded 36x28 tile straight through both streams."""
    g = np.load(os.path.join(G, "g3_layers.npz"))
    X6 = torch.from_numpy(g["input"])                       # already in [VV,VH,B,G,R,NIR] order
    # engine expects the dataset order [R,G,B,NIR,VV,VH]: invert the reorder
    X = torch.cat([X6[:, 4:5], X6[:, 3:4], X6[:, 2:3], X6[:, 5:6], X6[:, 0:2]], 1).cuda()
    eng = model.engines()[0]
    feats, saved = eng.forward(X, 0, 0, 36, 28, save=True)
    np.testing.assert_allclose(feats.cpu().numpy(), g["features"], rtol=0, atol=2e-5)
    ref = {"a1": "inc.conv.conv.2", "a2": "inc.conv.conv.5", "b1": "down_seq.down1.mpconv.1.conv.2",
           "b2": "down_seq.down1.mpconv.1.conv.5", "c1": "down_seq.down2.mpconv.1.conv.2",
           "c2": "down_seq.down2.mpconv.1.conv.5", "u2": "up_seq.up2.up", "e1": "up_seq.up2.conv.conv.2",
           "e2": "up_seq.up2.conv.conv.5", "u1": "up_seq.up1.up", "f1": "up_seq.up1.conv.conv.2"}
    for s in ("sar_stream", "optical_stream"):
        for k, r in ref.items():
            np.testing.assert_allclose(saved[s][k].cpu().numpy(), g[f"act/{s}.{r}"], rtol=0, atol=2e-5, err_msg=f"{s}.{k}")


def test_train_step_vs_reference_golden(model):
    """Reference recipe (run_train.py:201-238) through the drop-in module + torch autograd + torch Adam: loss,
    all 56 gradients, clipped norm and post-Adam parameters vs fixture g5."""
    from torch.nn.utils import clip_grad_norm_
    from popcorn_amd.model import POPCORN
    from popcorn_amd.utils.losses import get_loss
    g = np.load(os.path.join(G, "g5_train.npz"))
    torch.manual_seed(1600)
    m = POPCORN(input_channels=6, feature_extractor="DDA", occupancymodel=True, pretrained=True, biasinit=0.9407,
                sentinelbuildings=True).cuda()
    m.train()
    head_name = ["head.6.weight", "head.6.bias"]
    named = list(m.named_parameters())
    opt = torch.optim.Adam([
        {"params": [p for n, p in named if n not in head_name and "unetmodel" not in n], "weight_decay": 1e-5},
        {"params": [p for n, p in named if n not in head_name and "unetmodel" in n], "weight_decay": 1e-5},
        {"params": [p for n, p in named if n in head_name and "unetmodel" not in n], "weight_decay": 0.0}], lr=1e-4)
    sample0 = {k: torch.from_numpy(g[k]).cuda() for k in ("input", "admin_mask", "census_idx", "y")}
    sd_before = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    traj = []
    for step in range(3):
        torch.manual_seed(1700 + step)
        sample = dict(sample0)
        o = m(sample, train=True, padding=False, sparse=True)
        loss, ld = get_loss(o, sample, scale=o["scale"], loss=["log_l1_loss"], lam=[1.0], scale_regularization=0.01,
                            tag="weak")
        opt.zero_grad()
        (loss * 100.0).backward()
        if step == 0:
            assert o["scale"].numel() == int(g["step0/nsel"])
            assert rel_err(o["popcount"].detach().cpu().numpy(), g["step0/popcount"]) < 1e-4
            got = {n for n, p in named if p.grad is not None}
            assert got == set(g["step0/grad_names"].tolist())
            worst, hip_g, ref_g = 0.0, {}, {}
            for n, p in named:
                if p.grad is not None:
                    ref = g["step0/grad/" + n]
                    hip_g[n], ref_g[n] = p.grad.detach().cpu().clone(), torch.from_numpy(ref)
                    worst = max(worst, np.abs(p.grad.cpu().numpy() - ref).max() / max(np.abs(ref).max(), 1e-3))
            tie = worst > 2e-4
            if tie:
                # the reference's gradients (fixture g5) against the HIP path: above the bar only through a PROVEN ReLU / arg-max tie
                # flip (tests/tie_adjudication.py; the oracle restates the reference and equals the fixture to 1e-5, test_oracle_golden)
                from tests.tie_adjudication import assert_tie_flip
                cpu_s = {k: torch.from_numpy(g[k]) for k in ("input", "admin_mask", "census_idx", "y")}
                assert worst < 5e-3, worst
                assert_tie_flip(sd_before, cpu_s, sample0["input"], hip_g, ref_g, 1700, worst)
            # and without any adjudication: the fp64 oracle made to take the HIP forward's side at every ReLU mask / pooling arg-max
            # (O.ForceDecisions) is the HIP gradients' neighbour -- what separates them from the fixture is those decisions alone
            from tests.tie_adjudication import forced_decision_distance
            cpu_s = {k: torch.from_numpy(g[k]) for k in ("input", "admin_mask", "census_idx", "y")}
            wf, wname, flips, _ = forced_decision_distance(sd_before, cpu_s, sample0["input"], hip_g, 1700)
            print(f"\n[shared decisions] golden train step: HIP vs fp64 oracle under the HIP forward's decisions {wf:.2e} ({wname}); flips {flips}")
            assert wf < 1e-4, (wf, wname, flips)
            for k in [k for k in g.files if k.startswith("step0/lossdict/")]:
                kk = k[len("step0/lossdict/"):].replace("|", "/")
                assert abs(ld[kk] - float(g[k])) <= 1e-4 * max(1.0, abs(float(g[k]))), kk
        total = clip_grad_norm_(m.parameters(), 0.01)
        if step == 0:
            assert abs(total.item() - float(g["step0/total_norm"])) < 2e-4 * float(g["step0/total_norm"])
        opt.step()
        if step == 0:
            for n, p in named:
                if p.grad is not None:
                    d = np.abs(p.detach().cpu().numpy() - g["step0/param_after/" + n])
                    if not tie:
                        assert d.max() <= 1e-6, (n, d.max())
                    else:
                        # Adam's first step moves every element by lr * g / (|g| + eps) = +-lr: an element whose gradient is smaller than
                        # the (proven) tie-flip perturbation can land on the other side -- a few elements, by at most 2 lr
                        assert d.max() <= 2.0e-4 + 1e-6 and (d > 1e-6).mean() < 0.02, (n, d.max(), (d > 1e-6).mean())
        traj.append(loss.item())
    np.testing.assert_allclose(np.array(traj), g["loss_traj"], rtol=1e-4)


@pytest.mark.parametrize("flags", [dict(encoder_no_grad=True), dict(unet_no_grad=True, encoder_no_grad=True)])
def test_grad_truncation_modes_vs_oracle(model, flags):
    """limit1/limit2 regimes (run_train.py:191-198): encoder under no_grad / whole U-Net under no_grad."""
    from oracle import popcorn_oracle as O
    from popcorn_amd.utils.losses import get_loss
    g = np.load(os.path.join(G, "g5_train.npz"))
    sample = {k: torch.from_numpy(g[k]) for k in ("input", "admin_mask", "census_idx", "y")}
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    torch.manual_seed(5)
    _, _, ref_grads, _ = O.train_step_grads(sd, dict(sample), **flags)
    model.train()
    model.zero_grad()
    torch.manual_seed(5)
    s = {k: v.cuda() for k, v in sample.items()}
    o = model(s, train=True, padding=False, sparse=True, **flags)
    loss, _ = get_loss(o, s, scale=o["scale"], loss=["log_l1_loss"], lam=[1.0], scale_regularization=0.01, tag="weak")
    (loss * 100.0).backward()
    got = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    assert set(got) == set(ref_grads)
    worst = max((got[n].cpu() - r).abs().max().item() / max(r.abs().max().item(), 1e-3) for n, r in ref_grads.items())
    if worst > 2e-4:
        from tests.tie_adjudication import assert_tie_flip        # above the bar only through a proven ReLU / arg-max tie flip
        assert worst < 5e-3, worst
        assert_tie_flip(sd, dict(sample), s["input"], {n: got[n].cpu() for n in ref_grads}, ref_grads, 5, worst, **flags)
    # under shared decisions (the fp64 oracle on the HIP side of every ReLU mask / arg-max, tests/tie_adjudication.py) nothing is left
    from tests.tie_adjudication import forced_decision_distance
    wf, wname, flips, _ = forced_decision_distance(sd, dict(sample), s["input"], {n: got[n].cpu() for n in ref_grads}, 5, **flags)
    print(f"\n[shared decisions] truncation regime {flags}: HIP vs fp64 oracle under the HIP forward's decisions {wf:.2e} ({wname}); flips {flips}")
    assert wf < 1e-4, (wf, wname, flips)
    model.zero_grad()


def test_run_train_cli_loss_decreases_and_resume(tmp_path):
    """Counterpart of run_train.py end to end (fused HIP step, variable-size synthetic regions) + checkpoint resume."""
    from popcorn_amd.cli import Trainer, train_parser
    argv = ("-S2 -NIR -S1 -occmodel -senbuilds -pret -wd 1e-5 --biasinit 0.9407 -lr 1e-3 -e 2 --synthetic_regions 16 "
            f"-wb 4 --save_dir {tmp_path} -lt 2").split()
    t = Trainer(train_parser().parse_args(argv))
    first = t.train_step(next(iter(t.loader))).item()
    losses = t.train()
    assert torch.isfinite(torch.stack(losses)).all()
    ck = os.path.join(t.exp, "last_model.pth")
    assert os.path.exists(ck)
    d = torch.load(ck, weights_only=False)
    assert set(d) == {"model", "epoch", "iter", "optimizer", "scheduler"} and d["epoch"] == 2      # run_train.py:445-456
    t2 = Trainer(train_parser().parse_args(argv + ["-r", ck]))
    for (k1, v1), (k2, v2) in zip(t.model.state_dict().items(), t2.model.state_dict().items()):
        assert k1 == k2 and torch.equal(v1.cpu(), v2.cpu())
    assert torch.equal(t.fused.m.cpu(), t2.fused.m.cpu()) and t2.info["epoch"] == 2
    assert first == first


def test_torch_optimizer_path_equals_fused_path(tmp_path):
    """The drop-in module + torch autograd + torch.optim.Adam and the fused HIP step take the same optimisation step."""
    from popcorn_amd.cli import Trainer, train_parser
    base = ("-S2 -NIR -S1 -occmodel -senbuilds -pret -wd 1e-5 --biasinit 0.9407 --synthetic_regions 8 -wb 4 "
            f"--save_dir {tmp_path}").split()
    ta = Trainer(train_parser().parse_args(base))
    tb = Trainer(train_parser().parse_args(base + ["--torch_optimizer"]))
    sample = next(iter(ta.loader))
    for t in (ta, tb):
        torch.manual_seed(11)
        random.seed(11)                      # the augmentations draw from both generators (utils/transform.py)
        t.train_step({k: (v.clone() if torch.is_tensor(v) else v) for k, v in sample.items()})
    for (k1, v1), (k2, v2) in zip(ta.model.state_dict().items(), tb.model.state_dict().items()):
        torch.testing.assert_close(v1, v2, rtol=0, atol=2e-6, msg=lambda m, k=k1: f"{k}: {m}")


@pytest.mark.parametrize("ic", [2, 4])
def test_single_modality_vs_reference_golden(ic):
    """S1-only (input_channels=2) / S2-only (4) variants, popcorn.py:48-54,136-145,301-314: forward, loss and all 32
    gradients vs fixture g8; plus the fused training step on the same sample."""
    from popcorn_amd.model import POPCORN
    from popcorn_amd.train import FusedTrainStep
    from popcorn_amd.utils.losses import get_loss
    g = np.load(os.path.join(G, "g8_single_modality.npz"))
    torch.manual_seed(1600)
    m = POPCORN(input_channels=ic, feature_extractor="DDA", occupancymodel=True, pretrained=True, biasinit=0.9407,
                sentinelbuildings=True).cuda()
    for k in g.files:
        if k.startswith(f"ic{ic}/head."):
            assert np.array_equal(m.state_dict()[k[len(f"ic{ic}/"):]].cpu().numpy(), g[k])     # same seeded init
    m.train()
    s = {k: torch.from_numpy(g[f"ic{ic}/{k}"]).cuda() for k in ("input", "admin_mask", "census_idx", "y")}
    torch.manual_seed(5)
    o = m(dict(s), train=True, padding=False, sparse=True)
    loss, _ = get_loss(o, s, scale=o["scale"], loss=["log_l1_loss"], lam=[1.0], scale_regularization=0.01, tag="weak")
    (loss * 100.0).backward()
    assert abs(loss.item() - float(g[f"ic{ic}/loss"])) < 1e-4
    assert rel_err(o["popdensemap"].detach().cpu().numpy(), g[f"ic{ic}/popdensemap"]) < 1e-4
    assert o["scale"].shape == g[f"ic{ic}/scale"].shape
    got = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}
    assert set(got) == set(g[f"ic{ic}/grad_names"].tolist())
    for n, gr in got.items():
        r = g[f"ic{ic}/grad/{n}"]
        assert np.abs(gr.cpu().numpy() - r).max() <= 2e-4 * max(np.abs(r).max(), 1e-3), n
    with torch.no_grad():
        o2 = m({"input": s["input"]}, padding=True)
    assert rel_err(o2["popdensemap"].cpu().numpy(), g[f"ic{ic}/dense_pad1/popdensemap"]) < 1e-4
    # fused path: same gradients
    tr = FusedTrainStep(m, lr=1e-4, weight_decay=1e-5)
    torch.manual_seed(5)
    l2 = tr.step(dict(s))
    assert abs(l2[0].item() - float(g[f"ic{ic}/loss"])) < 1e-4
    for n in got:
        r = g[f"ic{ic}/grad/{n}"]
        assert np.abs(tr.grads[n].cpu().numpy() - r).max() <= 2e-4 * max(np.abs(r).max(), 1e-3), n


def test_fused_step_skips_frozen_groups_like_torch_adam():
    """limit1 / limit2 steps (run_train.py:191-198) leave the encoder / the whole U-Net without a gradient, and
    torch.optim.Adam skips such parameters entirely (no weight decay, no moment decay, no per-parameter step).  The fused
    step must do the same: frozen parameters bit-identical across the step, and after a mixed sequence of regimes every
    parameter equals the drop-in module + torch autograd + torch.optim.Adam run (per-group bias-correction steps)."""
    from torch.nn.utils import clip_grad_norm_
    from popcorn_amd import engine as E
    from popcorn_amd.model import POPCORN
    from popcorn_amd.train import FusedTrainStep
    from popcorn_amd.utils.losses import get_loss
    g = np.load(os.path.join(G, "g5_train.npz"))
    sample0 = {k: torch.from_numpy(g[k]).cuda() for k in ("input", "admin_mask", "census_idx", "y")}
    kw = dict(input_channels=6, feature_extractor="DDA", occupancymodel=True, pretrained=True, biasinit=0.9407,
              sentinelbuildings=True)
    torch.manual_seed(1600)
    ma = POPCORN(**kw).cuda()
    torch.manual_seed(1600)
    mb = POPCORN(**kw).cuda()
    wd = 1e-3                                    # large, so that a wrongly decayed frozen weight shows at once
    fused = FusedTrainStep(ma, lr=1e-3, weight_decay=wd, gradient_clip=0.01)
    head_name = ["head.6.weight", "head.6.bias"]
    named = list(mb.named_parameters())
    opt = torch.optim.Adam([
        {"params": [p for n, p in named if n not in head_name], "weight_decay": wd},
        {"params": [p for n, p in named if n in head_name], "weight_decay": 0.0}], lr=1e-3)
    is_enc = lambda n: n.startswith("unetmodel.") and any(("." + E.CONVS[t][0] + ".") in n for t in E.ENCODER)  # noqa: E731
    regimes = [dict(), dict(encoder_no_grad=True), dict(encoder_no_grad=True, unet_no_grad=True), dict(),
               dict(encoder_no_grad=True)]
    for it, flags in enumerate(regimes):
        before = {n: p.detach().clone() for n, p in ma.named_parameters()}
        m_before, v_before = fused.m.clone(), fused.v.clone()
        torch.manual_seed(40 + it)
        fused.step(dict(sample0), **flags)
        torch.manual_seed(40 + it)
        mb.train()
        s = dict(sample0)
        o = mb(s, train=True, padding=False, sparse=True, **flags)
        loss, _ = get_loss(o, s, scale=o["scale"], loss=["log_l1_loss"], lam=[1.0], scale_regularization=0.01, tag="weak")
        opt.zero_grad()                            # set_to_none: parameters outside the graph keep grad None -> skipped
        (loss * 100.0).backward()
        clip_grad_norm_(mb.parameters(), 0.01)
        opt.step()
        # frozen groups: bit-identical parameters and moments
        off = 0
        for n, p in zip(fused.names, [dict(ma.named_parameters())[n] for n in fused.names]):
            k = p.numel()
            frozen = (flags.get("unet_no_grad") and n.startswith("unetmodel.")) or (flags.get("encoder_no_grad") and is_enc(n))
            if frozen:
                assert torch.equal(p.detach(), before[n]), (it, n)
                assert torch.equal(fused.m[off:off + k], m_before[off:off + k]) and torch.equal(fused.v[off:off + k], v_before[off:off + k])
            else:
                assert not torch.equal(p.detach(), before[n]), (it, n)
            off += k
    assert fused.step_count.tolist()[:3] == [2, 4, 5]          # encoder, decoder, head updates in the sequence above
    pa, pb = dict(ma.named_parameters()), dict(mb.named_parameters())
    for n in fused.names:
        torch.testing.assert_close(pa[n].detach(), pb[n].detach(), rtol=0, atol=5e-6, msg=lambda m_, n=n: f"{n}: {m_}")


def test_trainer_validation_test_loops_guards_and_checkpoint_interop(tmp_path):
    """Trainer behaviour around the step (run_train.py:111-141,224-227,289-370,445-476): weak validation and the
    in-training target test run every --val_every_n_epochs and log their metrics; a non-finite loss raises on the fused
    path; a fused-step checkpoint carries a torch.optim.Adam state dict that the torch-optimizer trainer (= the reference
    recipe) loads, and vice versa; resuming recomputes the learning rate from the epoch."""
    import json as _json
    from popcorn_amd.cli import Trainer, train_parser
    base = ("-S2 -NIR -S1 -occmodel -senbuilds -pret -wd 1e-5 --biasinit 0.9407 -lr 1e-3 --synthetic_regions 8 -wb 4 "
            f"--save_dir {tmp_path} -lt 1 -wv -val 1 -lrs 1 -lrg 0.5 --fixed_hw 64 64").split()
    t = Trainer(train_parser().parse_args(base + ["-e", "2"]))
    t.train()
    recs = [_json.loads(l) for l in open(os.path.join(t.exp, "train_log.jsonl"))]
    val = [r for r in recs if any(k.endswith("/val") for k in r)]
    tst = [r for r in recs if any(k.endswith("/targettest") for k in r)]
    assert len(val) == 2 and len(tst) == 2
    assert {"Population_MainCensus_synthetic_fine/r2/val", "Population_MainCensus_synthetic_fine/mape/val"} <= set(val[0])
    assert all(np.isfinite(v) for r in tst for k, v in r.items() if k.endswith("/targettest"))
    assert any("Population_weak/r2" in r for r in recs)
    assert os.path.exists(os.path.join(t.exp, "synthetic_predictions.pt"))
    assert abs(t.fused.lr - 1e-3 * 0.25) < 1e-12                     # two epochs of StepLR(1, 0.5)
    ck = os.path.join(t.exp, "last_model.pth")
    d = torch.load(ck, weights_only=False)
    assert set(d) == {"model", "epoch", "iter", "optimizer", "scheduler"}
    assert {"state", "param_groups"} <= set(d["optimizer"]) and [len(g["params"]) for g in d["optimizer"]["param_groups"]] == [
        sum(1 for n, _ in t.model.named_parameters() if n not in ("head.6.weight", "head.6.bias") and "unetmodel" not in n),
        sum(1 for n, _ in t.model.named_parameters() if "unetmodel" in n), 2]
    # fused checkpoint -> fused trainer: moments, per-group steps, lr recomputed from the epoch
    t2 = Trainer(train_parser().parse_args(base + ["-e", "3", "-r", ck]))
    assert torch.equal(t2.fused.m.cpu(), t.fused.m.cpu()) and torch.equal(t2.fused.step_count.cpu(), t.fused.step_count.cpu())
    assert abs(t2.fused.lr - 1e-3 * 0.25) < 1e-12 and t2.info["epoch"] == 2
    # fused checkpoint -> torch.optim.Adam (the reference's resume path), and both continue identically for one step
    t3 = Trainer(train_parser().parse_args(base + ["-e", "3", "-r", ck, "--torch_optimizer"]))
    st = t3.optimizer.state_dict()["state"]
    assert len(st) == 56 and all(int(v["step"]) == 4 for v in st.values())
    # (the torch path resumes with the checkpoint's scheduler state, which -- as in the reference, run_train.py:123-141 --
    # was saved before the epoch's scheduler.step(); the fused path recomputes lr from the epoch instead)
    sample = next(iter(t3.loader))
    torch.manual_seed(11)
    random.seed(11)
    t3.model.train()
    t3.train_step({k: (v.clone() if torch.is_tensor(v) else v) for k, v in sample.items()})
    # torch-Adam checkpoint -> fused trainer
    ck3 = t3.save_model("torch")
    t4 = Trainer(train_parser().parse_args(base + ["-e", "3", "-r", ck3]))
    assert t4.fused.step_count.cpu().tolist()[:3] == [5, 5, 5]
    off = 0
    idx, _ = t4.fused._torch_adam_index([n for n, _ in t4.model.named_parameters()])
    st3 = t3.optimizer.state_dict()["state"]
    for n_, p in zip(t4.fused.names, [dict(t4.model.named_parameters())[n] for n in t4.fused.names]):
        k = p.numel()
        assert torch.equal(t4.fused.m[off:off + k].cpu(), st3[idx[n_]]["exp_avg"].reshape(-1).cpu()), n_
        off += k
    # non-finite loss: raised at the next log step of the fused path
    t5 = Trainer(train_parser().parse_args(base + ["-e", "1"]))
    with torch.no_grad():
        t5.model.head[6].bias.fill_(float("nan"))
    with pytest.raises(Exception, match="NaN/Inf"):
        t5.train()


@pytest.mark.parametrize("occ,senb", [(False, True), (False, False), (True, False)])
def test_fused_step_without_occupancy_model_or_with_its_own_building_layer_vs_oracle(occ, senb):
    """The non-default model variants in TRAINING (popcorn.py:113-114,177-181; utils/losses.py:63-76): occupancymodel = False (popdensemap
    = relu(out), scale = None -> NO scale regulariser in the loss whatever the flag says: the fused step used to add it, found by
    tools/sweep_train_variants.py) and sentinelbuildings = False with a building layer in the sample (used instead of the frozen
    extractor's score: the fused step used to ignore it).  Loss and all gradients against the oracle."""
    from oracle import popcorn_oracle as O
    from popcorn_amd.data.synthetic import make_raw_batch
    from popcorn_amd.model import POPCORN
    from popcorn_amd.train import FusedTrainStep
    torch.manual_seed(1600)
    m = POPCORN(input_channels=6, occupancymodel=occ, pretrained=True, biasinit=0.9407, sentinelbuildings=senb).cuda()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    b = make_raw_batch(3, 100, 100, seed=3, region="disc")
    cpu = {"input": O.select_normalize(b["raw"]), "admin_mask": b["admin_mask"], "census_idx": b["census_idx"], "y": b["y"]}
    if not senb:
        cpu["building_counts"] = torch.rand(3, 1, 100, 100, generator=torch.Generator().manual_seed(4))
    for use_graph in (False, True):
        tr = FusedTrainStep(m, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01, scale_regularization=0.01, use_graph=use_graph)
        torch.manual_seed(5)
        loss = tr.step({k: v.cuda() for k, v in cpu.items()}).clone()
        torch.cuda.synchronize()
        names = O.trainable_names(sd)
        wsd = dict(sd)
        for nm in names:
            wsd[nm] = sd[nm].detach().clone().requires_grad_(True)
        work = {k: v.clone() for k, v in cpu.items()}
        torch.manual_seed(5)
        out = O.popcorn_forward(wsd, work, padding=False, sparse=True, occupancymodel=occ, sentinelbuildings=senb)
        rl, _ = O.get_loss(out, work, scale=out["scale"], loss=("log_l1_loss",), lam=(1.0,), scale_regularization=0.01, tag="weak")
        (rl * 100.0).backward()
        assert abs(loss[0].item() - rl.item()) < 1e-5 * max(1.0, abs(rl.item())), (use_graph, loss, rl)
        if not occ:
            assert loss[1].item() == 0.0                                                    # no regulariser term
        for nm in names:
            g = wsd[nm].grad
            assert g is not None
            e = (tr.grads[nm].cpu() - g).abs().max().item()
            assert e <= 2e-4 * max(g.abs().max().item(), 1e-3), (use_graph, nm, e)
        m.load_state_dict(sd)
        tr.sync_from_model()


@pytest.mark.parametrize("with_buffer", [True, False])
def test_captured_step_on_static_buffers_reads_the_given_building_layer(with_buffer):
    """a sentinelbuildings = False model fed its own building layer THROUGH ``static_buffers()`` (the loader-facing static
    set of a captured step).  The layer must be an input of the captured graph: the loss follows it when it changes, equals the eager
    step's loss for the same layer, and differs from the frozen extractor's path."""
    from popcorn_amd.data.synthetic import make_raw_batch
    from popcorn_amd.model import POPCORN
    from popcorn_amd.train import FusedTrainStep
    from popcorn_amd import ops
    from popcorn_amd.data import stats
    B, H, W = 2, 100, 100
    b = make_raw_batch(B, H, W, seed=3, device="cuda", region="disc")
    x = ops.select_normalize(b["raw"], stats.BAND6, stats.MEAN6, stats.STD6)
    layers = [torch.rand(B, 1, H, W, generator=torch.Generator().manual_seed(s)).cuda() for s in (4, 5)]

    def fresh(use_graph):
        torch.manual_seed(1600)
        m = POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=False).cuda()
        return FusedTrainStep(m, lr=0.0, weight_decay=0.0, gradient_clip=0.01, use_graph=use_graph)      # lr 0: the steps are comparable

    eager = []
    for lay in layers:
        tr = fresh(False)
        torch.manual_seed(5)
        eager.append(tr.step({"input": x, "admin_mask": b["admin_mask"], "census_idx": b["census_idx"], "y": b["y"], "building_counts": lay})[0].item())
    assert abs(eager[0] - eager[1]) > 1e-6 * abs(eager[0])
    tr = fresh(True)
    st = tr.static_buffers(B, H, W, building=with_buffer)
    st["input"].copy_(x); st["admin_mask"].copy_(b["admin_mask"]); st["census_idx"].copy_(b["census_idx"]); st["y"].copy_(b["y"])
    got = []
    for lay in layers + layers[:1]:
        if with_buffer:
            st["building_counts"].copy_(lay)
            smp = st
        else:
            smp = dict(st)
            smp["building_counts"] = lay          # the static set plus the user's own tensor for the layer
        torch.manual_seed(5)
        got.append(tr.step(smp)[0].item())
    torch.cuda.synchronize()
    for g, e in zip(got, eager + eager[:1]):
        assert abs(g - e) <= 1e-6 * abs(e), (got, eager)
