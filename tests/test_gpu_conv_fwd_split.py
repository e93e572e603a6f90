"""Kernel parity of the split-operand FORWARD conv (csrc/conv3x3_fwd_s3.h: fp32 tensors, every operand split exactly into three bf16
numbers when a strip is staged, six partial products on the bf16 matrix pipe) against torch float64, next to the fp32-MFMA form it
replaces (pc_set_conv_split: 1 = split form wherever the tensors are aligned, 0 = fp32 MFMA everywhere).
Shapes: the DoubleConv layers (8 -> 8, 8 -> 16, 16 -> 16, 16 -> 8 incl. the pooled second output and the partial 1x1 logit;
networks.py:259-294) and the first conv of an Up block taken from the low-resolution map (networks.py:302-318).  Tolerance: 2e-5 abs on O(1) activations, and the split form's distance from float64
must stay within a small factor of the fp32-MFMA form's."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _mk(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _bn(c, g):
    return (torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1,
            torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5)          # conv bias, gamma, beta, mean, var


def _ref_layer(x, w, p, relu=True):
    bias, gamma, beta, mean, var = p
    y = F.conv2d(x.double(), w.double(), bias.double(), padding=1)
    y = (y - mean.double().view(1, -1, 1, 1)) / torch.sqrt(var.double().view(1, -1, 1, 1) + 1e-5) * gamma.double().view(1, -1, 1, 1) \
        + beta.double().view(1, -1, 1, 1)
    return torch.relu(y) if relu else y


class _form:
    def __init__(self, v):
        self.v = v

    def __enter__(self):
        from popcorn_amd import _lib as L
        self.prev = L.lib().pc_set_conv_split(self.v)

    def __exit__(self, *a):
        from popcorn_amd import _lib as L
        L.lib().pc_set_conv_split(self.prev)


@pytest.mark.parametrize("cin,cout", [(8, 16), (16, 16), (16, 8), (8, 8)])
@pytest.mark.parametrize("shape", [(2, 64, 64), (1, 36, 52), (3, 16, 32), (2, 20, 8), (1, 5, 4), (1, 128, 160)])
def test_forward_split_form_vs_float64(cin, cout, shape):
    from popcorn_amd import _lib as L
    from popcorn_amd import ops
    B, H, W = shape
    g = torch.Generator().manual_seed(100 + cin + cout + H)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.2
    p = _bn(cout, g)
    ref = _ref_layer(x, w, p)
    dp = [t.cuda() for t in p]
    xd, wd = x.cuda(), w.cuda()
    err = {}
    outs = {}
    for form in (1, 0):
        with _form(form):
            out = torch.full((B, cout, H, W), float("nan"), device="cuda")
            ops.conv3x3_fwd_group([{"a": xd, "w": wd, "bn": L.bn(dp[0], dp[1], dp[2], dp[3], dp[4], 1e-5), "out": out}])
            torch.cuda.synchronize()
        outs[form] = out
        err[form] = (out.cpu().double() - ref).abs().max().item()
    assert err[0] < 2e-5 and err[1] < 2e-5, err
    assert err[1] <= 4 * err[0] + 1e-6, err
    # a different kernel did run: the two forms round differently somewhere (and nowhere by more than a few ulp)
    if H * W >= 1024:
        assert not torch.equal(outs[1], outs[0])
    assert (outs[1] - outs[0]).abs().max().item() < 2e-5


def test_forward_split_form_single_tap_is_exact():
    """one non-zero tap of weight 1: the three parts of an operand sum to it exactly, whatever the accumulation order"""
    from popcorn_amd import ops
    x = _mk(1, 16, 20, 40, seed=5)
    with _form(1):
        for (co, ci, dy, dx) in [(1, 6, 0, 2), (15, 0, 2, 0), (3, 11, 1, 0), (9, 15, 2, 2)]:
            w = torch.zeros(16, 16, 3, 3)
            w[co, ci, dy, dx] = 1.0
            ref = F.conv2d(x, w, None, padding=1)
            out = ops.conv3x3_bn_relu(x.cuda(), w.cuda(), None, relu=False)
            assert torch.equal(out.cpu(), ref), (co, ci, dy, dx)


@pytest.mark.parametrize("cin,cout,shape", [(16, 16, (3, 32, 64)), (8, 16, (2, 64, 64)), (8, 8, (1, 128, 32)), (16, 16, (2, 64, 96))])
def test_forward_split_form_pooled_second_output(cin, cout, shape):
    """pool_out == MaxPool2d(2) of the ordinary output (networks.py:289), bit for bit, next to a problem of the same launch without one"""
    from popcorn_amd import _lib as L
    from popcorn_amd import ops
    B, H, W = shape
    with _form(1):
        probs, refs = [], []
        for i in range(2):
            x = _mk(B, cin, H, W, seed=10 + i).cuda()
            w = (_mk(cout, cin, 3, 3, seed=20 + i, scale=0.2)).cuda()
            b = _mk(cout, seed=30 + i, scale=0.1).cuda()
            out = torch.full((B, cout, H, W), float("nan"), device="cuda")
            pr = {"a": x, "w": w, "bn": L.bn(b), "out": out, "_k": b}
            if i == 0:
                pr["pool_out"] = ops.pool_out_like(out)
                pr["pool_out"].fill_(float("nan"))
            probs.append(pr)
            refs.append(F.relu(F.conv2d(x.cpu().double(), w.cpu().double(), b.cpu().double(), padding=1)))
        ops.conv3x3_fwd_group(probs)
        torch.cuda.synchronize()
    for pr, ref in zip(probs, refs):
        assert (pr["out"].cpu().double() - ref).abs().max().item() < 2e-5
    assert torch.equal(probs[0]["pool_out"], F.max_pool2d(probs[0]["out"], 2))


@pytest.mark.parametrize("Cs,hw", [(8, (32, 64)), (16, (36, 32)), (8, (128, 128)), (8, (44, 56)), (16, (20, 72)), (8, (12, 40)), (16, (64, 64)),
                                   (8, (4, 8))])
def test_forward_split_form_composed_up_vs_convt_then_conv(Cs, hw):
    """the first conv of an Up block from the LOW-resolution map in split form against torch float64 conv3x3(cat[skip, conv_transpose2d(z)])
    + BN + ReLU (borders, corners, interior), and against the fp32-MFMA form of the same composed weights"""
    from popcorn_amd import _lib as L
    from popcorn_amd import ops
    g = torch.Generator().manual_seed(31 + Cs + hw[1])
    H, W = hw
    B, Cz, nprob = 2, Cs, 3
    res = {}
    data = []
    for i in range(nprob):
        skip = torch.randn(B, Cs, H, W, generator=g)
        z = torch.randn(B, Cz, H // 2, W // 2, generator=g)
        w = torch.randn(8, Cs + Cz, 3, 3, generator=g) * 0.1
        wt = torch.randn(Cz, Cz, 2, 2, generator=g) * 0.2
        bt = torch.randn(Cz, generator=g)
        p = _bn(8, g)
        u = F.conv_transpose2d(z.double(), wt.double(), bt.double(), stride=2)
        data.append(([t.cuda() for t in (skip, z, w, wt, bt)], [t.cuda() for t in p], _ref_layer(torch.cat([skip.double(), u], 1), w, p)))
    for form in (1, 0):
        with _form(form):
            probs = []
            for dv, dp, _ in data:
                pr = {"skip": dv[0], "z": dv[1], "w": dv[2], "wt": dv[3], "bt": dv[4], "bn": L.bn(dp[0], dp[1], dp[2], dp[3], dp[4], 1e-5),
                      "out": torch.full((B, 8, H, W), float("nan"), device="cuda")}
                assert ops.conv3x3_up_fwd_ok(pr["skip"], pr["z"], pr["out"])
                probs.append(pr)
            ops.conv3x3_up_fwd_group(probs)
            torch.cuda.synchronize()
        res[form] = [pr["out"].cpu().double() for pr in probs]
    for i, (_, _, ref) in enumerate(data):
        scale = ref.abs().max().item()
        e1 = (res[1][i] - ref).abs().max().item() / scale
        e0 = (res[0][i] - ref).abs().max().item() / scale
        assert e1 < 3e-5 and e0 < 3e-5, (e1, e0)
        assert e1 <= 4 * e0 + 1e-6, (e1, e0)
        got = res[1][i]
        for sl in ((slice(None), slice(None), 0), (slice(None), slice(None), -1), (slice(None), slice(None), slice(None), 0),
                   (slice(None), slice(None), slice(None), -1)):
            assert (got[sl] - ref[sl]).abs().max().item() < 3e-5 * scale
    if H * W >= 1024:
        assert not all(torch.equal(a, b) for a, b in zip(res[1], res[0]))


def test_forward_split_form_partial_logit_output():
    """dot_w / dot_out (sum_co dot_w[co] * relu(bn(conv))[co] as a one-channel map) for one problem of a group, ragged height, next to a
    problem that writes its feature map: both multiplication forms against float64"""
    from popcorn_amd import _lib as L
    from popcorn_amd import ops
    B, H, W = 2, 30, 48
    g = torch.Generator().manual_seed(77)
    xs = [torch.randn(B, 8, H, W, generator=g) for _ in range(2)]
    ws = [torch.randn(8, 8, 3, 3, generator=g) * 0.2 for _ in range(2)]
    ps = [_bn(8, g) for _ in range(2)]
    dw = torch.randn(8, generator=g)
    ref0 = (_ref_layer(xs[0], ws[0], ps[0]) * dw.double().view(1, 8, 1, 1)).sum(1)
    ref1 = _ref_layer(xs[1], ws[1], ps[1])
    dps = [[t.cuda() for t in p] for p in ps]
    dwd = dw.cuda()
    for form in (1, 0):
        with _form(form):
            logits = torch.full((B, 2, H, W), float("nan"), device="cuda")
            out1 = torch.full((B, 8, H, W), float("nan"), device="cuda")
            bns = [L.bn(d[0], d[1], d[2], d[3], d[4], 1e-5) for d in dps]
            ops.conv3x3_fwd_group([{"a": xs[0].cuda(), "w": ws[0].cuda(), "bn": bns[0], "dot_w": dwd, "dot_out": logits[:, 1:2]},
                                   {"a": xs[1].cuda(), "w": ws[1].cuda(), "bn": bns[1], "out": out1}])
            torch.cuda.synchronize()
        assert (logits[:, 1].cpu().double() - ref0).abs().max().item() < 5e-5, form
        assert torch.isnan(logits[:, 0]).all()
        assert (out1.cpu().double() - ref1).abs().max().item() < 2e-5, form


def test_forward_split_form_fuzzed_geometries_and_views():
    """40 seeded cases: random (B, H, W % 4 == 0), channel counts 8 / 16, inputs and outputs that are CHANNEL WINDOWS of larger tensors
    (what the engine passes: feats[:, f0:f0 + 8], halves of a 16-channel map) and row-padded tensors (L.padded_rows), optional pooled
    second output where the geometry allows it; split form against float64, and nothing outside the output window is written"""
    import random
    from popcorn_amd import _lib as L
    from popcorn_amd import ops
    rnd = random.Random(20261003)
    for case in range(40):
        cin, cout = rnd.choice([(8, 8), (8, 16), (16, 16), (16, 8)])
        B = rnd.randint(1, 3)
        H = rnd.randint(3, 70)
        W = 4 * rnd.randint(1, 40)
        g = torch.Generator().manual_seed(1000 + case)
        x_full = torch.randn(B, cin + 8, H, W, generator=g)
        c0 = rnd.choice([0, 8])
        w = torch.randn(cout, cin, 3, 3, generator=g) * 0.2
        p = _bn(cout, g)
        ref = _ref_layer(x_full[:, c0:c0 + cin], w, p)
        xd = x_full.cuda()
        out_full = torch.full((B, cout + 16, H, W), float("nan"), device="cuda")
        o0 = rnd.choice([0, 8, 16])
        dp = [t.cuda() for t in p]
        pr = {"a": xd[:, c0:c0 + cin], "w": w.cuda(), "bn": L.bn(dp[0], dp[1], dp[2], dp[3], dp[4], 1e-5), "out": out_full[:, o0:o0 + cout]}
        want_pool = W % 32 == 0 and H % 4 == 0 and o0 == 0 and rnd.random() < 0.7
        if want_pool:
            # (the pooled copy needs a dense output tensor: ops.pool_out_like decides)
            dense = torch.full((B, cout, H, W), float("nan"), device="cuda")
            po = ops.pool_out_like(dense)
            if po is not None:
                pr["out"], pr["pool_out"] = dense, po
        with _form(1):
            ops.conv3x3_fwd_group([pr])
            torch.cuda.synchronize()
        got = pr["out"].cpu().double()
        err = (got - ref).abs().max().item()
        assert err < 2e-5, (case, cin, cout, B, H, W, err)
        if "pool_out" in pr:
            assert torch.equal(pr["pool_out"], F.max_pool2d(pr["out"], 2)), case
        else:
            rest = torch.ones(cout + 16, dtype=torch.bool)
            rest[o0:o0 + cout] = False
            assert torch.isnan(out_full[:, rest]).all(), (case, "wrote outside its channel window")
