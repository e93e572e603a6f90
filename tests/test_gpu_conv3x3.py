"""Kernel parity: HIP conv3x3 (+BN+ReLU, fused pool / concat / reflect loaders, dgrad epilogues) vs stock
PyTorch-CPU fp32 ops (the oracle's building blocks).  Tolerance: 1e-5 abs on O(1) activations (fp32 MFMA is an
exact fp32 fma chain; only the summation order differs from oneDNN)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _mk(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _bn_params(c, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1, torch.randn(c, generator=g) * 0.2,
            torch.rand(c, generator=g) + 0.3)


def _ref_cbr(x, w, b, gamma, beta, mean, var, relu=True):
    y = F.conv2d(x, w, b, padding=1)
    y = F.batch_norm(y, mean, var, gamma, beta, training=False, eps=1e-5)
    return F.relu(y) if relu else y


@pytest.mark.parametrize("cin,cout", [(2, 8), (4, 8), (8, 8), (16, 8), (32, 8), (8, 16), (16, 16)])
@pytest.mark.parametrize("shape", [(2, 64, 64), (1, 37, 53), (3, 16, 32)])
def test_conv_fwd_direct(cin, cout, shape):
    from popcorn_amd import ops
    B, H, W = shape
    x = _mk(B, cin, H, W, seed=1)
    w = _mk(cout, cin, 3, 3, seed=2, scale=0.2)
    b = _mk(cout, seed=3, scale=0.1)
    gamma, beta, mean, var = _bn_params(cout, 4)
    ref = _ref_cbr(x, w, b, gamma, beta, mean, var)
    dev = "cuda"
    out = ops.conv3x3_bn_relu(x.to(dev), w.to(dev), b.to(dev), gamma.to(dev), beta.to(dev), mean.to(dev), var.to(dev))
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=2e-5)


def test_conv_fwd_asymmetric_weights_catch_transpose():
    """Single non-zero tap, asymmetric in (dy,dx) and (co,ci): catches any row/col or tap flip."""
    from popcorn_amd import ops
    x = _mk(1, 8, 20, 40, seed=5)
    for (co, ci, dy, dx) in [(1, 6, 0, 2), (7, 0, 2, 0), (3, 3, 1, 0)]:
        w = torch.zeros(8, 8, 3, 3)
        w[co, ci, dy, dx] = 1.0
        ref = F.conv2d(x, w, None, padding=1)
        out = ops.conv3x3_bn_relu(x.cuda(), w.cuda(), None, relu=False)
        torch.testing.assert_close(out.cpu(), ref, rtol=0, atol=1e-6)


def test_conv_fwd_pool_and_concat_with_up_pad():
    from popcorn_amd import ops
    from popcorn_amd import _lib as L
    # pool fused: source 2x resolution (odd size -> floor)
    xs = _mk(2, 8, 45, 67, seed=6)
    w = _mk(16, 8, 3, 3, seed=7, scale=0.2)
    b = _mk(16, seed=8, scale=0.1)
    gamma, beta, mean, var = _bn_params(16, 9)
    ref = _ref_cbr(F.max_pool2d(xs, 2), w, b, gamma, beta, mean, var)
    out = ops.conv3x3_bn_relu(xs.cuda(), w.cuda(), b.cuda(), gamma.cuda(), beta.cuda(), mean.cuda(), var.cuda(),
                              a_mode=L.PC_SRC_POOL2)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=2e-5)
    # concat [skip(16), up(16)] with the up tensor smaller than the skip (Up's zero F.pad, networks.py:309-312)
    skip = _mk(2, 16, 23, 35, seed=10)
    upt = _mk(2, 16, 22, 34, seed=11)
    w = _mk(8, 32, 3, 3, seed=12, scale=0.1)
    b = _mk(8, seed=13, scale=0.1)
    gamma, beta, mean, var = _bn_params(8, 14)
    dy, dx = 1, 1
    upp = F.pad(upt, (dx // 2, dx - dx // 2, dy // 2, dy - dy // 2))
    ref = _ref_cbr(torch.cat([skip, upp], 1), w, b, gamma, beta, mean, var)
    out = ops.conv3x3_bn_relu(skip.cuda(), w.cuda(), b.cuda(), gamma.cuda(), beta.cuda(), mean.cuda(), var.cuda(),
                              b=upt.cuda(), b_offset=(dy // 2, dx // 2))
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=2e-5)


def test_conv_fwd_reflect_reorder():
    """add_padding(force) + channel reorder fused into the first conv (popcorn.py:243-245,130-134)."""
    from popcorn_amd import ops
    from popcorn_amd import _lib as L
    X = _mk(2, 6, 30, 41, seed=15)
    Xp = F.pad(X, (14, 14, 14, 14), mode="reflect")
    Xr = torch.cat([Xp[:, 4:6], torch.flip(Xp[:, :3], dims=(1,)), Xp[:, 3:4]], 1)
    for lo, hi, chmap in [(0, 2, (4, 5, 0, 0)), (2, 6, (2, 1, 0, 3))]:
        cin = hi - lo
        w = _mk(8, cin, 3, 3, seed=16 + lo, scale=0.3)
        b = _mk(8, seed=17, scale=0.1)
        gamma, beta, mean, var = _bn_params(8, 18)
        ref = _ref_cbr(Xr[:, lo:hi], w, b, gamma, beta, mean, var)
        out = ops.conv3x3_bn_relu(X.cuda(), w.cuda(), b.cuda(), gamma.cuda(), beta.cuda(), mean.cuda(), var.cuda(),
                                  a_mode=L.PC_SRC_REFLECT, a_pad=(14, 14), chmap=chmap, out_hw=(58, 69), a_channels=cin)
        torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("cin_total,c0,cn,cg", [(8, 0, 8, 8), (16, 8, 8, 8), (32, 0, 16, 8), (32, 16, 16, 8),
                                                (8, 0, 8, 16), (16, 0, 16, 16)])
@pytest.mark.parametrize("shape", [(2, 29, 47), (2, 32, 64)])            # ragged: scalar epilogues; aligned: the 16-byte ones
def test_conv_dgrad_plain_and_masked(cin_total, c0, cn, cg, shape):
    from popcorn_amd import ops
    from popcorn_amd import _lib as L
    B, H, W = shape
    x = _mk(B, cin_total, H, W, seed=20).requires_grad_(True)
    w = _mk(cg, cin_total, 3, 3, seed=21, scale=0.2)
    g = _mk(B, cg, H, W, seed=22)
    F.conv2d(x, w, None, padding=1).backward(g)
    ref = x.grad[:, c0:c0 + cn]
    out = torch.full((B, cn, H, W), 7.0, device="cuda")
    ops.conv3x3_dgrad(g.cuda(), w.cuda(), c0, cn, out)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=2e-5)
    # masked (+accumulate): out += dgrad * (act>0) * gamma/sqrt(var+eps)
    act = F.relu(_mk(B, cn, H, W, seed=23))
    gamma, beta, mean, var = _bn_params(cn, 24)
    scale = gamma / torch.sqrt(var + 1e-5)
    base = _mk(B, cn, H, W, seed=25)
    ref2 = base + ref * (act > 0) * scale.view(1, -1, 1, 1)
    out2 = base.clone().cuda()
    bnd = L.bn(None, gamma.cuda(), beta.cuda(), mean.cuda(), var.cuda())
    keep = (gamma, beta, mean, var)
    g_, b_, m_, v_ = (t.cuda() for t in keep)
    bnd = L.bn(None, g_, b_, m_, v_)
    ops.conv3x3_dgrad(g.cuda(), w.cuda(), c0, cn, out2, act=act.cuda(), act_bn=bnd, accumulate=True)
    torch.testing.assert_close(out2.cpu(), ref2, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("shape", [(2, 38, 53), (2, 64, 128), (3, 16, 64)])
@pytest.mark.parametrize("cn", [8, 16])                                   # d1a (8-channel input) / d2a (16: the im2col mapping)
def test_conv_dgrad_pool_scatter_matches_autograd(shape, cn):
    """d1a-style: y = conv(maxpool(relu_bn_out)); gradient w.r.t. the pre-pool activation's conv output.  The aligned
    shapes take the 16-byte epilogue, the ragged one the scalar path."""
    from popcorn_amd import ops
    from popcorn_amd import _lib as L
    B, Hs, Ws = shape
    pre = _mk(B, cn, Hs, Ws, seed=30).requires_grad_(True)        # conv output of the producing layer (pre-BN)
    gamma, beta, mean, var = _bn_params(cn, 31)
    act = F.relu(F.batch_norm(pre, mean, var, gamma, beta, training=False, eps=1e-5))
    w = _mk(16, cn, 3, 3, seed=32, scale=0.2)
    y = F.conv2d(F.max_pool2d(act, 2), w, None, padding=1)
    g = _mk(*y.shape, seed=33)
    y.backward(g)
    base = _mk(B, cn, Hs, Ws, seed=34)
    ref = base + pre.grad
    out = base.clone().cuda()
    g_, b_, m_, v_ = (t.cuda() for t in (gamma, beta, mean, var))
    bnd = L.bn(None, g_, b_, m_, v_)
    ops.conv3x3_dgrad(g.cuda(), w.cuda(), 0, cn, out, act=act.detach().cuda(), act_bn=bnd, pool=True, accumulate=True)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("cin,cout", [(2, 8), (4, 8), (8, 8), (16, 8), (32, 8), (8, 16), (16, 16)])
@pytest.mark.parametrize("shape", [(3, 64, 64), (1, 37, 53), (40, 32, 32)])
def test_conv_wgrad(cin, cout, shape):
    from popcorn_amd import ops
    B, H, W = shape
    x = _mk(B, cin, H, W, seed=40)
    w = _mk(cout, cin, 3, 3, seed=41, scale=0.2).requires_grad_(True)
    bias = torch.zeros(cout, requires_grad=True)
    g = _mk(B, cout, H, W, seed=42)
    F.conv2d(x, w, bias, padding=1).backward(g)
    dw, db = ops.conv3x3_wgrad(x.cuda(), g.cuda(), cout)
    scale = w.grad.abs().max().item()
    torch.testing.assert_close(dw.cpu(), w.grad, rtol=1e-5, atol=2e-6 * max(scale, 1.0))
    torch.testing.assert_close(db.cpu(), bias.grad, rtol=1e-5, atol=2e-6 * max(bias.grad.abs().max().item(), 1.0))
    # accumulate + determinism
    dw2, db2 = ops.conv3x3_wgrad(x.cuda(), g.cuda(), cout, dw=dw.clone(), db=db.clone(), accumulate=True)
    torch.testing.assert_close(dw2.cpu(), 2 * dw.cpu(), rtol=1e-6, atol=0)
    dw3, db3 = ops.conv3x3_wgrad(x.cuda(), g.cuda(), cout)
    assert torch.equal(dw3, dw) and torch.equal(db3, db)


def test_conv_wgrad_fused_loaders():
    from popcorn_amd import ops
    from popcorn_amd import _lib as L
    xs = _mk(2, 8, 45, 67, seed=43)
    w = _mk(16, 8, 3, 3, seed=44, scale=0.2).requires_grad_(True)
    y = F.conv2d(F.max_pool2d(xs, 2), w, None, padding=1)
    g = _mk(*y.shape, seed=45)
    y.backward(g)
    dw, _ = ops.conv3x3_wgrad(xs.cuda(), g.cuda(), 16, a_mode=L.PC_SRC_POOL2)
    torch.testing.assert_close(dw.cpu(), w.grad, rtol=1e-5, atol=1e-4)
    skip = _mk(2, 16, 23, 35, seed=46)
    upt = _mk(2, 16, 22, 34, seed=47)
    w = _mk(8, 32, 3, 3, seed=48, scale=0.1).requires_grad_(True)
    y = F.conv2d(torch.cat([skip, F.pad(upt, (0, 1, 0, 1))], 1), w, None, padding=1)
    g = _mk(*y.shape, seed=49)
    y.backward(g)
    dw, _ = ops.conv3x3_wgrad(skip.cuda(), g.cuda(), 8, b=upt.cuda(), b_offset=(0, 0))
    torch.testing.assert_close(dw.cpu(), w.grad, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("cin,cout,shape", [(8, 8, (2, 64, 64)), (16, 16, (3, 32, 64)), (8, 8, (1, 128, 32))])
def test_conv_fwd_pooled_second_output(cin, cout, shape):
    """pool_out of the grouped forward == MaxPool2d(2) of the ordinary output (networks.py:289), bit for bit, next to a
    problem of the same launch that does not ask for it."""
    from popcorn_amd import ops
    from popcorn_amd import _lib as L
    B, H, W = shape
    probs, refs = [], []
    for i in range(2):
        x = _mk(B, cin, H, W, seed=10 + i).cuda()
        w = (_mk(cout, cin, 3, 3, seed=20 + i, scale=0.2)).cuda()
        b = _mk(cout, seed=30 + i, scale=0.1).cuda()
        out = torch.empty(B, cout, H, W, device="cuda")
        pr = {"a": x, "w": w, "bn": L.bn(b), "out": out, "_k": b}
        if i == 0:
            pr["pool_out"] = ops.pool_out_like(out)
            assert pr["pool_out"] is not None and pr["pool_out"].shape == (B, cout, H // 2, W // 2)
        probs.append(pr)
        refs.append(F.relu(F.conv2d(x.cpu(), w.cpu(), b.cpu(), padding=1)))
    ops.conv3x3_fwd_group(probs)
    for pr, ref in zip(probs, refs):
        torch.testing.assert_close(pr["out"].cpu(), ref, rtol=1e-5, atol=2e-5)
    assert torch.equal(probs[0]["pool_out"], F.max_pool2d(probs[0]["out"], 2))


def test_conv_fwd_pooled_output_needs_full_strips():
    from popcorn_amd import ops
    assert ops.pool_out_like(torch.empty(1, 8, 36, 52, device="cuda")) is None      # W % 32 != 0
    assert ops.pool_out_like(torch.empty(1, 8, 30, 64, device="cuda")) is None      # H % 4 != 0


def test_conv_fwd_partial_logit_output():
    """dot_w / dot_out: sum_co dot_w[co] * relu(bn(conv))[co] as a one-channel map, for one problem of a group; the
    other problem writes its feature map as usual."""
    from popcorn_amd import ops
    from popcorn_amd import _lib as L
    B, H, W = 2, 32, 64
    gamma, beta, mean, var = _bn_params(8, 41)
    xs = [_mk(B, 8, H, W, seed=50 + i) for i in range(2)]
    ws = [_mk(8, 8, 3, 3, seed=60 + i, scale=0.2) for i in range(2)]
    bs = [_mk(8, seed=70 + i, scale=0.1) for i in range(2)]
    dw = _mk(8, seed=80)
    logits = torch.full((B, 2, H, W), float("nan"), device="cuda")
    out1 = torch.empty(B, 8, H, W, device="cuda")
    keep = [t.cuda() for t in (gamma, beta, mean, var)]
    bsd = [b.cuda() for b in bs]                       # L.bn() stores raw pointers: the tensors must stay alive
    bns = [L.bn(b, *keep) for b in bsd]
    dwd = dw.cuda()
    probs = [{"a": xs[0].cuda(), "w": ws[0].cuda(), "bn": bns[0], "dot_w": dwd, "dot_out": logits[:, 1:2]},
             {"a": xs[1].cuda(), "w": ws[1].cuda(), "bn": bns[1], "out": out1}]
    ops.conv3x3_fwd_group(probs)
    ref0 = _ref_cbr(xs[0], ws[0], bs[0], gamma, beta, mean, var)
    ref1 = _ref_cbr(xs[1], ws[1], bs[1], gamma, beta, mean, var)
    torch.testing.assert_close(logits[:, 1].cpu(), (ref0 * dw.view(1, 8, 1, 1)).sum(1), rtol=1e-5, atol=5e-5)
    assert torch.isnan(logits[:, 0]).all()                      # the other channel is untouched
    torch.testing.assert_close(out1.cpu(), ref1, rtol=1e-5, atol=2e-5)


@pytest.fixture(params=[1, 0], ids=["split", "fp32mfma"])
def conv_form(request):
    """The fused backward's multiplication form (popcorn_hip.h: pc_set_conv_split): 1 = operands split exactly into three bf16 numbers
    when a strip is staged, six partial products on the bf16 matrix pipe (conv3x3_bwd_s3_kernel); 0 = v_mfma_f32_16x16x4_f32."""
    from popcorn_amd import _lib as L
    prev = L.lib().pc_set_conv_split(int(request.param))
    yield int(request.param)
    L.lib().pc_set_conv_split(prev)


@pytest.mark.parametrize("shape", [(2, 64, 64), (1, 36, 52), (3, 128, 128), (2, 20, 8)])
@pytest.mark.parametrize("case", ["plain_masked", "concat_up_half", "concat_skip_half_accumulate"])
def test_conv_backward_fused_op_fp32(shape, case, conv_form):
    """pc_conv3x3_bwd_group in fp32 mode (planar tensors, 8 -> 8 channels): data gradient (+ ReLU / BN factor, += form) and
    weight / bias gradient of a layer, or of one column block of a concat layer, in one launch, against torch autograd (fp64)."""
    from popcorn_amd import ops, _lib as L
    B, H, W = shape
    cin_total, c0 = {"plain_masked": (8, 0), "concat_up_half": (16, 8), "concat_skip_half_accumulate": (16, 0)}[case]
    masked = case != "concat_up_half"
    x = F.relu(_mk(B, 8, H, W, seed=80)) if masked else _mk(B, 8, H, W, seed=80)
    w = _mk(8, cin_total, 3, 3, seed=81, scale=0.2)
    g = _mk(B, 8, H, W, seed=82)
    gamma, beta, mean, var = _bn_params(8, 83)
    scale = gamma / torch.sqrt(var + 1e-5)
    wd = w.double().requires_grad_(True)
    xd = torch.zeros(B, cin_total, H, W, dtype=torch.double)
    xd[:, c0:c0 + 8] = x.double()
    xd.requires_grad_(True)
    bias = torch.zeros(8, dtype=torch.double, requires_grad=True)
    F.conv2d(xd, wd, bias, padding=1).backward(g.double())
    gx = xd.grad[:, c0:c0 + 8]
    ref = (gx * (x > 0) * scale.view(1, 8, 1, 1).double()) if masked else gx
    acc = case == "concat_skip_half_accumulate"
    prev = _mk(B, 8, H, W, seed=84)
    out = prev.cuda() if acc else torch.empty(B, 8, H, W, device="cuda")
    dw = torch.full((8, cin_total, 3, 3), 7.0, device="cuda")
    db = torch.empty(8, device="cuda")
    wb = ops.WgradBatch(torch.device("cuda"))
    wb.conv3x3_bwd_group([{"g": g.cuda(), "x": x.cuda(), "w": w.cuda(), "out": out, "dw": dw, "db": db,
                           "x_bn": L.bn(None, gamma.cuda(), beta.cuda(), mean.cuda(), var.cuda()) if masked else None}],
                         cin_total, c0, accumulate=acc)
    wb.finish()
    torch.testing.assert_close(out.cpu().double(), ref + (prev.double() if acc else 0), rtol=1e-5, atol=2e-5)
    gw = wd.grad[:, c0:c0 + 8]
    assert (dw[:, c0:c0 + 8].cpu().double() - gw).abs().max().item() <= 2e-5 * gw.abs().max().item()
    other = [c for c in range(cin_total) if not c0 <= c < c0 + 8]
    assert bool((dw[:, other] == 7.0).all())
    assert (db.cpu().double() - bias.grad).abs().max().item() <= 2e-5 * bias.grad.abs().max().item()


@pytest.mark.parametrize("shape", [(2, 64, 64), (1, 36, 52), (5, 32, 96), (2, 20, 8)])
def test_conv_backward_fused_op_fp32_16_channels(shape):
    """Split-operand form only: a 16 -> 16 layer (down1's second conv, networks.py:284-295) as the two 8-channel column blocks of its
    input over the same 16-channel gradient, ONE launch -- data gradient (masked), weight and bias gradient against autograd in fp64."""
    from popcorn_amd import ops, _lib as L
    B, H, W = shape
    x = F.relu(_mk(B, 16, H, W, seed=180))
    w = _mk(16, 16, 3, 3, seed=181, scale=0.2)
    g = _mk(B, 16, H, W, seed=182)
    gamma, beta, mean, var = _bn_params(16, 183)
    scale = (gamma / torch.sqrt(var + 1e-5)).view(1, 16, 1, 1).double()
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    bias = torch.zeros(16, dtype=torch.double, requires_grad=True)
    F.conv2d(xd, wd, bias, padding=1).backward(g.double())
    ref = xd.grad * (x > 0) * scale
    out = torch.empty(B, 16, H, W, device="cuda")
    dw, db = torch.full((16, 16, 3, 3), 7.0, device="cuda"), torch.empty(16, device="cuda")
    gd, xc = g.cuda(), x.cuda()
    wb = ops.WgradBatch(torch.device("cuda"))
    wb.conv3x3_bwd_group([{"g": gd, "x": xc[:, 8 * i:8 * i + 8], "w": w.cuda(), "out": out[:, 8 * i:8 * i + 8], "dw": dw,
                           "db": db if i == 0 else None, "c0_add": 8 * i,
                           "x_bn": L.bn(None, gamma[8 * i:8 * i + 8].cuda(), beta[8 * i:8 * i + 8].cuda(), mean[8 * i:8 * i + 8].cuda(),
                                        var[8 * i:8 * i + 8].cuda())} for i in (0, 1)], 16, 0)
    wb.finish()
    torch.testing.assert_close(out.cpu().double(), ref, rtol=1e-5, atol=4e-5)
    assert (dw.cpu().double() - wd.grad).abs().max().item() <= 2e-5 * wd.grad.abs().max().item()
    assert (db.cpu().double() - bias.grad).abs().max().item() <= 2e-5 * bias.grad.abs().max().item()


@pytest.mark.parametrize("shape,cg", [((2, 32, 32), 16), ((1, 18, 28), 16), ((3, 64, 64), 16), ((2, 32, 64), 8)])
def test_conv_backward_fused_op_fp32_pool_scatter(shape, cg):
    """Split-operand form only: the first conv of a Down block (MaxPool2d(2) -> conv, networks.py:289) -- x is the saved pooled copy of
    pool_act; the data gradient is scattered (+=) to the first arg-max of every 2 x 2 window of the full-resolution gradient, times the
    ReLU / BN factor of pool_act's producer; against autograd through F.max_pool2d in fp64."""
    from popcorn_amd import ops, _lib as L
    B, H, W = shape
    act = F.relu(_mk(B, 8, 2 * H, 2 * W, seed=190))
    w = _mk(cg, 8, 3, 3, seed=191, scale=0.2)
    g = _mk(B, cg, H, W, seed=192)
    prev = _mk(B, 8, 2 * H, 2 * W, seed=193)
    gamma, beta, mean, var = _bn_params(8, 194)
    scale = (gamma / torch.sqrt(var + 1e-5)).view(1, 8, 1, 1).double()
    ad, wd = act.double().requires_grad_(True), w.double().requires_grad_(True)
    bias = torch.zeros(cg, dtype=torch.double, requires_grad=True)
    F.conv2d(F.max_pool2d(ad, 2), wd, bias, padding=1).backward(g.double())
    ref = prev.double() + ad.grad * (act > 0) * scale
    pooled = F.max_pool2d(act, 2)
    out = prev.cuda()
    dw, db = torch.empty(cg, 8, 3, 3, device="cuda"), torch.empty(cg, device="cuda")
    wb = ops.WgradBatch(torch.device("cuda"))
    wb.conv3x3_bwd_group([{"g": g.cuda(), "x": pooled.cuda(), "w": w.cuda(), "out": out, "dw": dw, "db": db, "pool_act": act.cuda(),
                           "x_bn": L.bn(None, gamma.cuda(), beta.cuda(), mean.cuda(), var.cuda())}], 8, 0, accumulate=True)
    wb.finish()
    # (ties inside a window: random data has none; the first-arg-max rule itself is pinned by test_conv_dgrad_pool_scatter_matches_autograd)
    torch.testing.assert_close(out.cpu().double(), ref, rtol=1e-5, atol=4e-5)
    assert (dw.cpu().double() - wd.grad).abs().max().item() <= 2e-5 * wd.grad.abs().max().item()
    assert (db.cpu().double() - bias.grad).abs().max().item() <= 2e-5 * bias.grad.abs().max().item()


def test_conv_backward_fused_split_form_has_the_error_of_fp32_arithmetic():
    """The split form is fp32 arithmetic, not a reduced-precision mode: measured against autograd in FLOAT64 on a 3 x 128 x 128 batch
    its data gradient, weight gradient and bias gradient are as close as the v_mfma_f32_16x16x4_f32 form's (same bound for both;
    the printed ratio is the evidence)."""
    from popcorn_amd import ops, _lib as L
    B, H, W = 3, 128, 128
    x = F.relu(_mk(B, 8, H, W, seed=170))
    w = _mk(8, 8, 3, 3, seed=171, scale=0.2)
    g = _mk(B, 8, H, W, seed=172)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    bias = torch.zeros(8, dtype=torch.double, requires_grad=True)
    F.conv2d(xd, wd, bias, padding=1).backward(g.double())
    err = {}
    for form in (1, 0):
        prev = L.lib().pc_set_conv_split(form)
        try:
            out = torch.empty(B, 8, H, W, device="cuda")
            dw, db = torch.empty(8, 8, 3, 3, device="cuda"), torch.empty(8, device="cuda")
            wb = ops.WgradBatch(torch.device("cuda"))
            wb.conv3x3_bwd_group([{"g": g.cuda(), "x": x.cuda(), "w": w.cuda(), "out": out, "dw": dw, "db": db}], 8, 0)
            wb.finish()
            err[form] = ((out.cpu().double() - xd.grad).abs().max().item() / xd.grad.abs().max().item(),
                         (dw.cpu().double() - wd.grad).abs().max().item() / wd.grad.abs().max().item(),
                         (db.cpu().double() - bias.grad).abs().max().item() / bias.grad.abs().max().item())
        finally:
            L.lib().pc_set_conv_split(prev)
    print("fused conv backward vs float64 (gx, dw, db max-rel): split", err[1], "fp32 mfma", err[0])
    for a, b in zip(err[1], err[0]):
        assert a <= 2e-6 and b <= 2e-6 and a <= 4 * b + 2e-7


def test_conv_backward_fused_op_fp32_refuses_unaligned():
    from popcorn_amd import ops, _lib as L
    x = torch.zeros(1, 8, 10, 10, device="cuda")
    wb = ops.WgradBatch(torch.device("cuda"))
    with pytest.raises(L.PopcornHipError):
        wb.conv3x3_bwd_group([{"g": x, "x": x, "w": torch.zeros(8, 8, 3, 3, device="cuda"), "out": torch.empty_like(x),
                               "dw": torch.empty(8, 8, 3, 3, device="cuda"), "db": torch.empty(8, device="cuda")}], 8, 0)


@pytest.mark.parametrize("geom", [(2, 6, 20, 31, 3, 5, 2, 4), (1, 6, 100, 100, 14, 14, 14, 14), (3, 6, 9, 12, 0, 3, 4, 0)])
def test_reflect_pad_select_matches_torch(geom):
    """pc_reflect_pad_select: reflect padding + channel gather of the model input (popcorn.py:231-258,130-134) in one pass."""
    from popcorn_amd import ops
    B, Cc, H, W, t, b, l, r = geom
    x = _mk(B, Cc, H, W, seed=95)
    sel = [4, 5, 2, 1, 0, 3]
    ref = F.pad(x[:, sel], (l, r, t, b), mode="reflect")
    out = ops.reflect_pad_select(x.cuda(), sel, t, b, l, r)
    assert torch.equal(out.cpu(), ref)


def test_conv_backward_fused_both_concat_blocks_in_one_launch():
    """c0_add: the skip block (masked) and the up-sampled block (unmasked, no second bias gradient) of a concat layer as two
    problems of ONE pc_conv3x3_bwd_group launch, against autograd through torch.cat."""
    from popcorn_amd import ops, _lib as L
    B, H, W = 2, 64, 64
    skip, up = F.relu(_mk(B, 8, H, W, seed=60)), _mk(B, 8, H, W, seed=61)
    w = _mk(8, 16, 3, 3, seed=62, scale=0.2)
    g = _mk(B, 8, H, W, seed=63)
    gamma, beta, mean, var = _bn_params(8, 64)
    xd = torch.cat([skip, up], 1).double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    bias = torch.zeros(8, dtype=torch.double, requires_grad=True)
    F.conv2d(xd, wd, bias, padding=1).backward(g.double())
    scale = (gamma / torch.sqrt(var + 1e-5)).view(1, 8, 1, 1).double()
    o_skip, o_up = torch.empty(B, 8, H, W, device="cuda"), torch.empty(B, 8, H, W, device="cuda")
    dw, db = torch.empty(8, 16, 3, 3, device="cuda"), torch.empty(8, device="cuda")
    wb = ops.WgradBatch(torch.device("cuda"))
    gd = g.cuda()
    wb.conv3x3_bwd_group([{"g": gd, "x": skip.cuda(), "w": w.cuda(), "out": o_skip, "dw": dw, "db": db,
                           "x_bn": L.bn(None, gamma.cuda(), beta.cuda(), mean.cuda(), var.cuda())},
                          {"g": gd, "x": up.cuda(), "w": w.cuda(), "out": o_up, "dw": dw, "db": None, "c0_add": 8}], 16, 0)
    wb.finish()
    torch.testing.assert_close(o_skip.cpu().double(), xd.grad[:, :8] * (skip > 0) * scale, rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(o_up.cpu().double(), xd.grad[:, 8:], rtol=1e-5, atol=2e-5)
    assert (dw.cpu().double() - wd.grad).abs().max().item() <= 2e-5 * wd.grad.abs().max().item()
    assert (db.cpu().double() - bias.grad).abs().max().item() <= 2e-5 * bias.grad.abs().max().item()


@pytest.mark.gpu
def test_cross_lane_helpers_are_bit_identical_to_the_shuffle_butterflies():
    """common.h: pc_sum8 / pc_lane_xor* / pc_xor16_sum / pc_xor32_sum (DPP controls and gfx950's v_permlane swaps) replaced every
    __shfl_xor (= ds_bpermute_b32, an LDS round trip) of the epilogues: same values, same association order, bit for bit."""
    import ctypes as C
    from popcorn_amd import _lib as L
    rng = np.random.default_rng(5)
    for trial in range(20):
        x = (rng.standard_normal(64) * 10.0 ** rng.integers(-3, 4)).astype(np.float32)
        xin = torch.from_numpy(x).cuda()
        out = torch.empty(384, device="cuda")
        L.check(L.lib().pc_debug_lane_ops(L.ptr(xin), L.ptr(out), L.stream_ptr()), "pc_debug_lane_ops")
        o = out.cpu().numpy()
        lane = np.arange(64)
        s = x.copy()
        for k in (1, 2, 4):
            s = (s + s[lane ^ k]).astype(np.float32)
        assert np.array_equal(o[0:64].view(np.uint32), s.view(np.uint32)), trial
        assert np.array_equal(o[64:128], np.maximum(x, x[lane ^ 8]))
        assert np.array_equal(o[128:192].view(np.uint32), (x + x[lane ^ 16]).astype(np.float32).view(np.uint32))
        assert np.array_equal(o[192:256].view(np.uint32), (x + x[lane ^ 32]).astype(np.float32).view(np.uint32))
        assert np.array_equal(o[256:320], x[lane ^ 1]) and np.array_equal(o[320:384], x[lane ^ 2])
