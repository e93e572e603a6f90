"""The one-launch 32 x 32 level (pc_level2_fwd_group: down2's DoubleConv + up2's ConvTranspose2d with the maps resident in LDS;
reference model/DDA_model/utils/networks.py:253-271,284-295,302-306) against torch-CPU on the same operands, and the U-Net
forward with the fused level against the layer-by-layer launches."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _bn(c, g):
    return (torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1,
            torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5)          # conv bias, gamma, beta, mean, var


def _ref_layer(x, w, p):
    bias, gamma, beta, mean, var = p
    y = F.conv2d(x.double(), w.double(), bias.double(), padding=1)
    y = (y - mean.double().view(1, -1, 1, 1)) / torch.sqrt(var.double().view(1, -1, 1, 1) + 1e-5) * gamma.double().view(1, -1, 1, 1) \
        + beta.double().view(1, -1, 1, 1)
    return torch.relu(y)


@pytest.mark.parametrize("nprob,save", [(1, True), (4, True), (2, False)])
def test_level2_fwd_group_vs_torch(nprob, save):
    from popcorn_amd import _lib as L
    from popcorn_amd import ops
    g = torch.Generator().manual_seed(11 + nprob)
    B = 3
    probs, refs, keep = [], [], []
    for i in range(nprob):
        # a strided view (channel slice of a wider tensor) for the input: descriptors carry the strides
        xw = torch.relu(torch.randn(B, 20, 32, 32, generator=g))
        x = xw[:, 2:18]
        w1, w2 = torch.randn(16, 16, 3, 3, generator=g) * 0.1, torch.randn(16, 16, 3, 3, generator=g) * 0.1
        p1, p2 = _bn(16, g), _bn(16, g)
        wt, bt = torch.randn(16, 16, 2, 2, generator=g) * 0.2, torch.randn(16, generator=g) * 0.1
        c1 = _ref_layer(x, w1, p1)
        c2 = _ref_layer(c1, w2, p2)
        u2 = F.conv_transpose2d(c2, wt.double(), bt.double(), stride=2)
        refs.append((c1.float(), c2.float(), u2.float()))
        dv = [t.cuda() for t in (w1, w2, wt, bt)]
        d1, d2 = [t.cuda() for t in p1], [t.cuda() for t in p2]
        xd = xw.cuda()[:, 2:18]
        pr = {"x": xd, "w1": dv[0], "w2": dv[1], "wt": dv[2], "bt": dv[3],
              "bn1": L.bn(d1[0], d1[1], d1[2], d1[3], d1[4], 1e-5), "bn2": L.bn(d2[0], d2[1], d2[2], d2[3], d2[4], 1e-5),
              "u2": torch.full((B, 16, 64, 64), float("nan"), device="cuda")}
        if save:
            pr["c1"] = torch.full((B, 16, 32, 32), float("nan"), device="cuda")
            pr["c2"] = torch.full((B, 16, 32, 32), float("nan"), device="cuda")
        assert ops.level2_fwd_ok(pr["x"], pr["u2"])
        keep.append((dv, d1, d2))
        probs.append(pr)
    ops.level2_fwd_group(probs)
    torch.cuda.synchronize()
    for pr, (c1, c2, u2) in zip(probs, refs):
        names = [("u2", u2)] + ([("c1", c1), ("c2", c2)] if save else [])
        for name, ref in names:
            got = pr[name].cpu()
            err = (got - ref).abs().max().item() / ref.abs().max().item()
            assert err < 2e-5, (name, err)                      # fp32 MFMA accumulation vs float64: a few ulp of the largest value


def test_level2_refuses_other_geometries():
    from popcorn_amd import ops
    x = torch.zeros(2, 16, 32, 32, device="cuda")
    assert ops.level2_fwd_ok(x, torch.zeros(2, 16, 64, 64, device="cuda"))
    assert not ops.level2_fwd_ok(torch.zeros(2, 16, 30, 32, device="cuda"), torch.zeros(2, 16, 60, 64, device="cuda"))
    assert not ops.level2_fwd_ok(torch.zeros(2, 16, 32, 33, device="cuda")[..., 1:], torch.zeros(2, 16, 64, 64, device="cuda"))   # rows not 16-byte aligned
    assert not ops.level2_fwd_ok(x.bfloat16(), torch.zeros(2, 16, 64, 64, device="cuda"))


def test_unet_forward_fused_level_equals_layerwise(monkeypatch):
    """forward_multi with the one-launch level against the three launches it replaces: same saved activations (c1, c2, u2) and
    features to fp32 summation-order accuracy, for the trainable U-Net next to the frozen extractor (4 problems, 2 saved)."""
    from popcorn_amd import engine as E
    from popcorn_amd.model import POPCORN
    torch.manual_seed(1600)
    m = POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
    eng_u, eng_b = m.engines()
    X = torch.randn(2, 6, 100, 100, generator=torch.Generator().manual_seed(2)).cuda()
    outs = {}
    for flag in (True, False):
        monkeypatch.setattr(E, "FUSED_LEVEL2", flag)
        (f_b, f_u), (_, saved) = E.forward_multi([eng_b, eng_u], X, 14, 14, 128, 128, [False, True], logit_only=[True, False])
        torch.cuda.synchronize()
        outs[flag] = (f_b.clone(), f_u.clone(), {s: {k: saved[s][k].clone() for k in ("c1", "c2", "e1")} for s in ("sar_stream", "optical_stream")})
    for a, b in ((outs[True][0], outs[False][0]), (outs[True][1], outs[False][1])):
        assert (a - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item())
    for s in ("sar_stream", "optical_stream"):
        for k in ("c1", "c2", "e1"):
            a, b = outs[True][2][s][k], outs[False][2][s][k]
            assert (a - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item()), (s, k)


@pytest.mark.parametrize("use_graph", [False, True])
def test_fused_step_from_raw_tiles_equals_step_from_normalised_input(use_graph):
    """FusedTrainStep fed the RAW 15-band tile (its first launch = pc_select_normalize_pad: band select + normalise + reflect
    padding in one pass, the unpadded input never exists) against the same step fed ``ops.select_normalize``'s output:
    identical arithmetic, so losses and parameters agree bit for bit over three steps."""
    from popcorn_amd import ops
    from popcorn_amd.data import stats
    from popcorn_amd.data.synthetic import make_raw_batch
    from popcorn_amd.model import POPCORN
    from popcorn_amd.train import FusedTrainStep
    batch = make_raw_batch(3, 100, 100, seed=5, device="cuda", region="disc")
    x = ops.select_normalize(batch["raw"], stats.BAND6, stats.MEAN6, stats.STD6)
    runs = []
    for key, data in (("input", x), ("raw", batch["raw"])):
        torch.manual_seed(1600)
        m = POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
        tr = FusedTrainStep(m, lr=1e-3, weight_decay=1e-5, gradient_clip=0.01, use_graph=use_graph)
        s = {key: data, "admin_mask": batch["admin_mask"], "census_idx": batch["census_idx"], "y": batch["y"]}
        losses = []
        for it in range(3):
            torch.manual_seed(40 + it)
            losses.append(tr.step(dict(s))[0].item())
        torch.cuda.synchronize()
        runs.append((losses, tr.flat_p.clone()))
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    assert torch.equal(runs[0][1], runs[1][1])


def test_select_normalize_pad_equals_normalise_then_pad():
    from popcorn_amd import ops
    from popcorn_amd.data import stats
    g = torch.Generator().manual_seed(3)
    raw = (torch.rand(2, 15, 37, 52, generator=g) * 9000).cuda()
    order = [4, 5, 2, 1, 0, 3]
    got = ops.select_normalize_pad(raw, [stats.BAND6[c] for c in order], [stats.MEAN6[c] for c in order], [stats.STD6[c] for c in order], 5, 6, 7, 5)
    x = ops.select_normalize(raw, stats.BAND6, stats.MEAN6, stats.STD6)
    want = torch.nn.functional.pad(x[:, order], (7, 5, 5, 6), mode="reflect")
    assert torch.equal(got, want)


def test_level2_bwd_group_vs_autograd():
    """pc_level2_bwd_group (both weight / bias gradients, the data-gradient chain through the two convolutions and the pooling
    scatter in one launch) against torch autograd (float64) of  pool -> conv+BN+ReLU -> conv+BN+ReLU  on the same operands."""
    from popcorn_amd import _lib as L
    from popcorn_amd import ops
    g = torch.Generator().manual_seed(23)
    B, nprob = 3, 2
    wb = ops.WgradBatch(torch.device("cuda"))
    probs, refs, keep = [], [], []
    for i in range(nprob):
        b2 = torch.relu(torch.randn(B, 16, 64, 64, generator=g)).double().requires_grad_(True)       # post-ReLU activations (some zeros)
        w1 = (torch.randn(16, 16, 3, 3, generator=g) * 0.1).double().requires_grad_(True)
        w2 = (torch.randn(16, 16, 3, 3, generator=g) * 0.1).double().requires_grad_(True)
        p1, p2 = _bn(16, g), _bn(16, g)
        bias1 = p1[0].double().requires_grad_(True)
        bias2 = p2[0].double().requires_grad_(True)

        def layer(x, w, bias, p):
            _, gamma, beta, mean, var = p
            y = F.conv2d(x, w, bias, padding=1)
            return torch.relu((y - mean.double().view(1, -1, 1, 1)) / torch.sqrt(var.double().view(1, -1, 1, 1) + 1e-5)
                              * gamma.double().view(1, -1, 1, 1) + beta.double().view(1, -1, 1, 1))
        x = F.max_pool2d(b2, 2)
        c1 = layer(x, w1, bias1, p1)
        c2 = layer(c1, w2, bias2, p2)
        gout = torch.randn(B, 16, 32, 32, generator=g).double()          # dL/dc2 (post-ReLU)
        c2.backward(gout)
        # what the kernel receives: dL/d(conv2 output) = gout * relu'(c2) * bn2 scale
        s2 = (p2[1] / torch.sqrt(p2[4] + 1e-5)).double().view(1, -1, 1, 1)
        g2 = (gout * (c2.detach() > 0) * s2).float()
        # the producer of b2 is a conv+BN+ReLU layer: the scatter multiplies by relu'(b2) * its BN scale -- emulate with a BN of scale 1.7
        pa = (torch.zeros(16), torch.full((16,), 1.7), torch.zeros(16), torch.zeros(16), torch.ones(16) - 1e-5)
        out0 = torch.randn(B, 16, 64, 64, generator=g)
        refs.append((w1.grad.float(), bias1.grad.float(), w2.grad.float(), bias2.grad.float(),
                     out0 + (b2.grad * 1.7).float() * (b2.detach() > 0)))
        d1 = [t.cuda() for t in p1]
        da = [t.cuda() for t in pa]
        pr = {"g2": g2.cuda(), "c1": c1.detach().float().cuda(), "x": x.detach().float().cuda(), "w1": w1.detach().float().cuda(),
              "w2": w2.detach().float().cuda(), "bn1": L.bn(None, d1[1], d1[2], d1[3], d1[4], 1e-5), "act": b2.detach().float().cuda(),
              "act_bn": L.bn(None, da[1], da[2], da[3], da[4], 1e-5), "out": out0.cuda(),
              "dw1": torch.full((16, 16, 3, 3), float("nan"), device="cuda"), "db1": torch.full((16,), float("nan"), device="cuda"),
              "dw2": torch.full((16, 16, 3, 3), float("nan"), device="cuda"), "db2": torch.full((16,), float("nan"), device="cuda")}
        assert ops.level2_bwd_ok(pr["g2"], pr["c1"], pr["x"], pr["act"], pr["out"])
        keep.append((d1, da))
        probs.append(pr)
    wb.level2_bwd_group(probs)
    wb.finish()
    torch.cuda.synchronize()
    for pr, (rw1, rb1, rw2, rb2, rout) in zip(probs, refs):
        for name, ref in (("dw2", rw2), ("db2", rb2), ("dw1", rw1), ("db1", rb1), ("out", rout)):
            got = pr[name].cpu()
            err = (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)
            assert err < 5e-5, (name, err)


def test_unet_backward_fused_level_equals_layerwise(monkeypatch):
    """Whole train-step gradients with the one-launch 32 x 32 level (forward + backward) against the layer-by-layer launches."""
    from popcorn_amd import engine as E
    from popcorn_amd import ops
    from popcorn_amd.data import stats
    from popcorn_amd.data.synthetic import make_raw_batch
    from popcorn_amd.model import POPCORN
    from popcorn_amd.train import FusedTrainStep
    batch = make_raw_batch(3, 100, 100, seed=9, device="cuda", region="disc")
    x = ops.select_normalize(batch["raw"], stats.BAND6, stats.MEAN6, stats.STD6)
    grads = {}
    for flag in (True, False):
        monkeypatch.setattr(E, "FUSED_LEVEL2", flag)
        torch.manual_seed(1600)
        m = POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
        tr = FusedTrainStep(m, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01)
        torch.manual_seed(4)
        tr.step({"input": x, "admin_mask": batch["admin_mask"], "census_idx": batch["census_idx"], "y": batch["y"]})
        torch.cuda.synchronize()
        grads[flag] = {k: v.clone() for k, v in tr.grads.items()}
    for k in grads[True]:
        a, b = grads[True][k], grads[False][k]
        assert (a - b).abs().max().item() <= 1e-4 * max(b.abs().max().item(), 1e-6), k


@pytest.mark.parametrize("Cs,hw", [(8, (32, 64)), (16, (36, 32)), (8, (128, 128)), (8, (44, 60)), (16, (20, 76)), (8, (12, 38))])
def test_conv3x3_up_fwd_group_vs_convt_then_conv(Cs, hw):
    """pc_conv3x3_up_fwd_group (the first conv of an Up block from the LOW-resolution map: composed 2x2-neighbourhood weights per
    output parity, transposed-conv bias through the in-image taps; no up-sampled tensor) against torch float64
    conv3x3(cat[skip, conv_transpose2d(z)]) + BN + ReLU -- borders, corners and interior."""
    from popcorn_amd import _lib as L
    from popcorn_amd import ops
    g = torch.Generator().manual_seed(31 + Cs)
    H, W = hw
    B, Cz, nprob = 2, Cs, 3
    probs, refs, keep = [], [], []
    for i in range(nprob):
        skip = torch.randn(B, Cs, H, W, generator=g)
        z = torch.randn(B, Cz, H // 2, W // 2, generator=g)
        w = torch.randn(8, Cs + Cz, 3, 3, generator=g) * 0.1
        wt = torch.randn(Cz, Cz, 2, 2, generator=g) * 0.2
        bt = torch.randn(Cz, generator=g)
        p = _bn(8, g)
        u = F.conv_transpose2d(z.double(), wt.double(), bt.double(), stride=2)
        refs.append(_ref_layer(torch.cat([skip.double(), u], 1), w, p).float())
        dv = [t.cuda() for t in (skip, z, w, wt, bt)]
        dp = [t.cuda() for t in p]
        out = torch.full((B, 8, H, W), float("nan"), device="cuda")
        if W % 4:                  # widths that are not a multiple of 4 need 16-byte aligned rows (L.padded_rows: the inference path)
            with L.padded_rows():
                sk = L.empty_act(B, Cs, H, W, "cuda")
                out = L.empty_act(B, 8, H, W, "cuda")
            sk.copy_(dv[0]); out.fill_(float("nan"))
            dv[0] = sk
        pr = {"skip": dv[0], "z": dv[1], "w": dv[2], "wt": dv[3], "bt": dv[4], "bn": L.bn(dp[0], dp[1], dp[2], dp[3], dp[4], 1e-5),
              "out": out}
        assert ops.conv3x3_up_fwd_ok(pr["skip"], pr["z"], pr["out"])
        keep.append((dv, dp))
        probs.append(pr)
    ops.conv3x3_up_fwd_group(probs)
    torch.cuda.synchronize()
    for pr, ref in zip(probs, refs):
        got = pr["out"].cpu()
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        assert err < 3e-5, err
        # the image border separately (bias-through-taps corrections)
        for sl in ((slice(None), slice(None), 0), (slice(None), slice(None), -1), (slice(None), slice(None), slice(None), 0),
                   (slice(None), slice(None), slice(None), -1)):
            assert (got[sl] - ref[sl]).abs().max().item() < 3e-5 * ref.abs().max().item()


@pytest.mark.parametrize("Cs,hw", [(8, (128, 128)), (16, (64, 64)), (8, (64, 64)),
                                   # round 5: column tiles (interior halos from the neighbouring tile, ragged last tile, narrow maps)
                                   (8, (40, 256)), (8, (36, 416)), (16, (28, 208)), (16, (24, 136)), (8, (16, 24)), (16, (12, 16)), (8, (20, 96))])
def test_conv3x3_up_bwd_group_vs_autograd(Cs, hw):
    """pc_conv3x3_up_bwd_group (backward of the up-sampled half of an Up block's first conv from the LOW-resolution map: data
    gradient through the composed 4 x 4 stride-2 window, weight gradient of the composed weights + parity / border sums, chain
    rule to the module's own parameters) against torch autograd (float64) of conv3x3(conv_transpose2d(z, Wt, bt), W[:, Cs:])."""
    from popcorn_amd import _lib as L
    from popcorn_amd import ops
    g = torch.Generator().manual_seed(41 + Cs + hw[0])
    H, W = hw
    B, Cz, nprob = 2, Cs, 2
    probs, refs, keep = [], [], []
    fwd = []
    for i in range(nprob):
        z = torch.relu(torch.randn(B, Cz, H // 2, W // 2, generator=g)).double().requires_grad_(True)
        w = (torch.randn(8, Cs + Cz, 3, 3, generator=g) * 0.1).double().requires_grad_(True)
        wt = (torch.randn(Cz, Cz, 2, 2, generator=g) * 0.2).double().requires_grad_(True)
        bt = torch.randn(Cz, generator=g).double().requires_grad_(True)
        G = torch.randn(B, 8, H, W, generator=g)
        u = F.conv_transpose2d(z, wt, bt, stride=2)
        y = F.conv2d(u, w[:, Cs:], None, padding=1)
        (y * G.double()).sum().backward()
        pz = _bn(Cz, g)
        sz = (pz[1] / torch.sqrt(pz[4] + 1e-5)).double().view(1, -1, 1, 1)
        refs.append(((z.grad * (z.detach() > 0) * sz).float(), w.grad[:, Cs:].float(), wt.grad.float(), bt.grad.float()))
        dz = [t.cuda() for t in pz]
        skip = torch.randn(B, Cs, H, W, generator=g).cuda()
        pb = _bn(8, g)
        db = [t.cuda() for t in pb]
        f = {"skip": skip, "z": z.detach().float().cuda(), "w": w.detach().float().cuda(), "wt": wt.detach().float().cuda(),
             "bt": bt.detach().float().cuda(), "bn": L.bn(db[0], db[1], db[2], db[3], db[4], 1e-5), "out": torch.empty(B, 8, H, W, device="cuda")}
        fwd.append(f)
        keep.append((dz, db))
        probs.append({"g": G.cuda(), "z": f["z"], "z_bn": L.bn(None, dz[1], dz[2], dz[3], dz[4], 1e-5),
                      "gz": torch.full((B, Cz, H // 2, W // 2), float("nan"), device="cuda"), "w": f["w"], "wt": f["wt"], "bt": f["bt"],
                      "dw": torch.full((8, Cs + Cz, 3, 3), 7.0, device="cuda"), "dwt": torch.full((Cz, Cz, 2, 2), float("nan"), device="cuda"),
                      "dbt": torch.full((Cz,), float("nan"), device="cuda")})
    slots = ops.conv3x3_up_fwd_group(fwd)
    for pr, sl in zip(probs, slots):
        pr["fwd_ws"] = sl
        assert ops.conv3x3_up_bwd_ok(pr["g"], pr["z"], pr["gz"])
    ops.conv3x3_up_bwd_group(probs)
    torch.cuda.synchronize()
    for pr, (rgz, rdw, rdwt, rdbt) in zip(probs, refs):
        assert torch.equal(pr["dw"][:, :Cs].cpu(), torch.full((8, Cs, 3, 3), 7.0))        # the skip half is not this call's business
        for name, got, ref in (("gz", pr["gz"], rgz), ("dw", pr["dw"][:, Cs:], rdw), ("dwt", pr["dwt"], rdwt), ("dbt", pr["dbt"], rdbt)):
            err = (got.cpu() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)
            assert err < 5e-5, (name, err)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("use_graph", [False, True])
def test_fused_step_from_the_loaders_uint16_s2_and_fp32_s1_equals_the_step_from_raw_tiles(precision, use_graph):
    """pc_ingest_split: the step fed {raw_s2: uint16 digital numbers of the 4 selected S2 bands, raw_s1: fp32 S1} -- what a loader reads
    from disk, two thirds of the host-to-device bytes of fp32 -- against the step fed the fp32 15-band tile: the uint16 -> fp32 conversion
    is exact, so losses and parameters agree bit for bit over three steps (both arithmetic modes, eager and replayed)."""
    from popcorn_amd.data import stats
    from popcorn_amd.data.synthetic import make_raw_batch
    from popcorn_amd.model import POPCORN
    from popcorn_amd.train import FusedTrainStep
    batch = make_raw_batch(3, 100, 100, seed=5, device="cuda", region="disc")
    b6 = list(stats.BAND6)
    s2 = batch["raw"][:, b6[:4]].to(torch.int32).cpu().to(torch.uint16).cuda().contiguous()
    s1 = batch["raw"][:, b6[4:]].contiguous()
    assert torch.equal(s2.cpu().to(torch.int32).float().cuda(), batch["raw"][:, b6[:4]])           # the synthetic S2 values are integers
    runs = []
    for data in ({"raw": batch["raw"]}, {"raw_s2": s2, "raw_s1": s1}):
        torch.manual_seed(1600)
        m = POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
        m.set_precision(precision)
        tr = FusedTrainStep(m, lr=1e-3, weight_decay=1e-5, gradient_clip=0.01, use_graph=use_graph)
        losses = []
        for step in range(3):
            torch.manual_seed(50 + step)
            losses.append(tr.step({**data, "admin_mask": batch["admin_mask"], "census_idx": batch["census_idx"], "y": batch["y"]}).tolist())
        torch.cuda.synchronize()
        runs.append((losses, tr.flat_p.clone()))
    assert runs[0][0] == runs[1][0]
    assert torch.equal(runs[0][1], runs[1][1])


def test_ingest_split_on_an_unfused_geometry_falls_back_to_the_two_step_form():
    """A tile whose padding differs between the two networks (64 x 48: no shared padded domain) takes the convert + select_normalize
    path; same result as the fp32-fed step."""
    from popcorn_amd.data import stats
    from popcorn_amd.data.synthetic import make_raw_batch
    from popcorn_amd.model import POPCORN
    from popcorn_amd.train import FusedTrainStep
    batch = make_raw_batch(2, 64, 48, seed=6, device="cuda", region="disc")
    b6 = list(stats.BAND6)
    s2 = batch["raw"][:, b6[:4]].to(torch.int32).cpu().to(torch.uint16).cuda().contiguous()
    s1 = batch["raw"][:, b6[4:]].contiguous()
    out = []
    for data in ({"raw": batch["raw"]}, {"raw_s2": s2, "raw_s1": s1}):
        torch.manual_seed(1600)
        m = POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
        tr = FusedTrainStep(m, lr=1e-3, weight_decay=1e-5, gradient_clip=0.01)
        torch.manual_seed(9)
        out.append((tr.step({**data, "admin_mask": batch["admin_mask"], "census_idx": batch["census_idx"], "y": batch["y"]}).tolist(), tr.flat_p.clone()))
    assert out[0][0] == out[1][0] and torch.equal(out[0][1], out[1][1])
