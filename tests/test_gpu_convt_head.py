"""Kernel parity: ConvTranspose2d 2x2/s2 (fwd, dgrad, wgrad), building-score tail, sparsity mask, sparse head
forward and ordered compaction -- HIP vs stock PyTorch-CPU ops / the oracle.  fp32 tolerance 1e-5 rel on O(1) data;
mask / index paths bit-exact."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _mk(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize("C_", [8, 16])
@pytest.mark.parametrize("shape", [(2, 32, 32), (3, 9, 21), (1, 40, 16)])
def test_convt_fwd_dgrad_wgrad(C_, shape):
    from popcorn_amd import ops
    from popcorn_amd import _lib as L
    B, H, W = shape
    x = _mk(B, C_, H, W, seed=1).requires_grad_(True)
    w = _mk(C_, C_, 2, 2, seed=2, scale=0.3).requires_grad_(True)
    bias = _mk(C_, seed=3, scale=0.1).requires_grad_(True)
    y = F.conv_transpose2d(x, w, bias, stride=2)
    g = _mk(*y.shape, seed=4)
    y.backward(g)
    out = ops.convt2x2(x.detach().cuda(), w.detach().cuda(), bias.detach().cuda())
    torch.testing.assert_close(out.cpu(), y.detach(), rtol=1e-5, atol=1e-5)
    gx = torch.empty(B, C_, H, W, device="cuda")
    ops.convt2x2_dgrad(g.cuda(), w.detach().cuda(), gx)
    torch.testing.assert_close(gx.cpu(), x.grad, rtol=1e-5, atol=1e-5)
    # fused ReLU/BN backward of the producer
    act = F.relu(_mk(B, C_, H, W, seed=5))
    gen = torch.Generator().manual_seed(6)
    gamma, var = torch.rand(C_, generator=gen) + 0.5, torch.rand(C_, generator=gen) + 0.3
    beta, mean = torch.zeros(C_), torch.zeros(C_)
    keep = [t.cuda() for t in (gamma, beta, mean, var)]
    bnd = L.bn(None, *keep)
    ops.convt2x2_dgrad(g.cuda(), w.detach().cuda(), gx, act=act.cuda(), act_bn=bnd)
    ref = x.grad * (act > 0) * (gamma / torch.sqrt(var + 1e-5)).view(1, -1, 1, 1)
    torch.testing.assert_close(gx.cpu(), ref, rtol=1e-5, atol=1e-5)
    dw, db = ops.convt2x2_wgrad(x.detach().cuda(), g.cuda())
    torch.testing.assert_close(dw.cpu(), w.grad, rtol=1e-5, atol=2e-6 * max(1.0, w.grad.abs().max().item()))
    torch.testing.assert_close(db.cpu(), bias.grad, rtol=1e-5, atol=2e-6 * max(1.0, bias.grad.abs().max().item()))
    dw2, db2 = ops.convt2x2_wgrad(x.detach().cuda(), g.cuda())
    assert torch.equal(dw, dw2) and torch.equal(db, db2)


@pytest.mark.parametrize("C_", [8, 16])
@pytest.mark.parametrize("shape", [(2, 32, 32), (3, 16, 64), (1, 40, 16)])
@pytest.mark.parametrize("masked", [True, False])
def test_convt_backward_fused_fp32(C_, shape, masked):
    """pc_convt2x2_bwd_group (fp32): weight / bias gradient AND data gradient (+ ReLU / BN factor of x's producer) of a transposed
    conv in one launch, against autograd; the fused form is for aligned tensors with W % 16 == 0 and refuses others."""
    from popcorn_amd import ops
    from popcorn_amd import _lib as L
    B, H, W = shape
    x = F.relu(_mk(B, C_, H, W, seed=11)).requires_grad_(True)
    w = _mk(C_, C_, 2, 2, seed=12, scale=0.3).requires_grad_(True)
    bias = _mk(C_, seed=13, scale=0.1).requires_grad_(True)
    y = F.conv_transpose2d(x, w, bias, stride=2)
    g = _mk(*y.shape, seed=14)
    y.backward(g)
    gen = torch.Generator().manual_seed(16)
    gamma, var = torch.rand(C_, generator=gen) + 0.5, torch.rand(C_, generator=gen) + 0.3
    keep = [t.cuda() for t in (gamma, torch.zeros(C_), torch.zeros(C_), var)]
    gx = torch.full((B, C_, H, W), 7.0, device="cuda")
    dw, db = torch.empty(C_, C_, 2, 2, device="cuda"), torch.empty(C_, device="cuda")
    wb = ops.WgradBatch(torch.device("cuda"))
    wb.convt2x2_bwd_group([{"x": x.detach().cuda(), "g": g.cuda(), "w": w.detach().cuda(), "out": gx, "dw": dw, "db": db,
                            "x_bn": L.bn(None, *keep) if masked else None}])
    wb.finish()
    ref = x.grad * (x > 0) * (gamma / torch.sqrt(var + 1e-5)).view(1, -1, 1, 1) if masked else x.grad
    torch.testing.assert_close(gx.cpu(), ref.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dw.cpu(), w.grad, rtol=1e-5, atol=2e-6 * max(1.0, w.grad.abs().max().item()))
    torch.testing.assert_close(db.cpu(), bias.grad, rtol=1e-5, atol=2e-6 * max(1.0, bias.grad.abs().max().item()))


def test_convt_backward_fused_fp32_refuses_ragged_rows():
    from popcorn_amd import ops
    from popcorn_amd import _lib as L
    x, g = torch.zeros(1, 8, 9, 21, device="cuda"), torch.zeros(1, 8, 18, 42, device="cuda")
    wb = ops.WgradBatch(torch.device("cuda"))
    with pytest.raises(L.PopcornHipError):
        wb.convt2x2_bwd_group([{"x": x, "g": g, "w": torch.zeros(8, 8, 2, 2, device="cuda"), "out": torch.empty_like(x),
                                "dw": torch.empty(8, 8, 2, 2, device="cuda"), "db": torch.empty(8, device="cuda")}])


def test_outconv_sigmoid_crop():
    from popcorn_amd import ops
    feat = _mk(2, 16, 58, 69, seed=7)
    w = _mk(1, 16, 1, 1, seed=8, scale=0.4)
    b = _mk(1, seed=9)
    ref = torch.sigmoid(F.conv2d(feat, w, b))[:, :, 14:-14, 14:-14]
    out = ops.outconv_sigmoid_crop(feat.cuda(), w.cuda(), b.cuda(), 30, 41, 14, 14)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["b2_100", "b1_131x77", "b2_40x52"])
def test_sparsity_mask_bit_exact_vs_golden(name):
    """Mask is an index path: bit-exact against the reference's own output (fixture g4)."""
    from popcorn_amd import ops
    g = np.load(os.path.join(G, "g4_mask.npz"))
    bc = torch.from_numpy(g[f"{name}/building_counts"])
    admin = torch.from_numpy(g[f"{name}/admin_mask"])
    census = torch.from_numpy(g[f"{name}/census_idx"])
    B, H, W = admin.shape
    rowsel = torch.zeros(H, dtype=torch.uint8)
    colsel = torch.zeros(W, dtype=torch.uint8)
    rowsel[torch.from_numpy(g[f"{name}/xindices"])] = 1
    colsel[torch.from_numpy(g[f"{name}/yindices"])] = 1
    mask, counts = ops.sparsity_mask(bc.cuda(), admin.cuda(), census.cuda(), rowsel.cuda(), colsel.cuda())
    ref = g[f"{name}/mask"]
    assert np.array_equal(mask.cpu().numpy().astype(bool), ref)
    assert counts[0].item() == int(ref.sum())
    # empty selection -> falls back to the region (popcorn.py:374-375)
    mask2, counts2 = ops.sparsity_mask(torch.zeros_like(bc).cuda(), admin.cuda(), census.cuda(),
                                       torch.zeros(H, dtype=torch.uint8).cuda(), torch.zeros(W, dtype=torch.uint8).cuda())
    region = (admin == census.view(-1, 1, 1)).numpy()
    assert np.array_equal(mask2.cpu().numpy().astype(bool), region)
    assert counts2[0].item() == int(region.sum())


def test_compact_masked_order():
    from popcorn_amd import ops
    gen = torch.Generator().manual_seed(10)
    for n in (1, 1000, 1024, 5000, 300007):
        src = torch.randn(n, generator=gen)
        mask = (torch.rand(n, generator=gen) < 0.37)
        out, cnt = ops.compact_masked(src.cuda(), mask.to(torch.uint8).cuda())
        k = cnt.item()
        assert k == int(mask.sum())
        assert torch.equal(out[:k].cpu(), src[mask])


@pytest.fixture
def head_form(request):
    """The head kernels' multiplication form (popcorn_hip.h: pc_set_head_split): 1 = exact 3-way bf16 operand splits on the bf16 matrix
    pipe (default), 0 = fp32 MFMA."""
    from popcorn_amd import _lib as L
    prev = L.lib().pc_set_head_split(int(request.param))
    yield int(request.param)
    L.lib().pc_set_head_split(prev)


@pytest.mark.parametrize("head_form", [1, 0], indirect=True)
@pytest.mark.parametrize("sparse", [True, False])
@pytest.mark.parametrize("shape", [(2, 100, 100, 128, 128, 14, 14), (1, 37, 29, 64, 64, 13, 17)])
def test_head_fwd_vs_oracle(sparse, shape, head_form):
    from oracle import popcorn_oracle as O
    from popcorn_amd import ops
    B, H, W, Hp, Wp, py, px = shape
    sd = O.load_golden_state(G)
    feat = _mk(B, 16, Hp, Wp, seed=11)
    gen = torch.Generator().manual_seed(12)
    building = torch.rand(B, 1, H, W, generator=gen)
    admin = (torch.rand(B, H, W, generator=gen) < 0.6).float() * 5.0
    census = torch.full((B,), 5, dtype=torch.int64)
    mask = (torch.rand(B, H, W, generator=gen) < 0.5) & (admin == 5.0)
    headin = feat[:, :, py:py + H, px:px + W]
    with torch.no_grad():
        if sparse:
            out = O.sparse_head_forward(sd, headin, mask)[:, 0]
        else:
            out = O.head_forward(sd, headin)[:, 0]
        scale = F.relu(out)
        pd = scale * building[:, 0]
        pc = (pd * (admin == census.view(-1, 1, 1))).sum((1, 2))
    ht = [sd[f"head.{i}.{n}"].cuda() for i in (0, 2, 4, 6) for n in ("weight", "bias")]
    s_map, pd_gpu, pc_gpu = ops.head_fwd(feat.cuda(), py, px, H, W, ht, building.cuda(),
                                         mask=mask.to(torch.uint8).cuda() if sparse else None,
                                         admin_mask=admin.cuda(), census_idx=census.cuda())
    torch.testing.assert_close(s_map.cpu(), scale, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(pd_gpu.cpu(), pd, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(pc_gpu.cpu(), pc, rtol=2e-5, atol=1e-4)
    # no admin mask: plain sum (popcorn.py:189-190)
    _, _, pc2 = ops.head_fwd(feat.cuda(), py, px, H, W, ht, building.cuda())
    with torch.no_grad():
        ref2 = (F.relu(O.head_forward(sd, headin)[:, 0]) * building[:, 0]).sum((1, 2))
    torch.testing.assert_close(pc2.cpu(), ref2, rtol=2e-5, atol=1e-4)


def test_head_split_products_have_the_error_of_fp32_arithmetic():
    """The split form (six bf16 x bf16 partial products of exact 3-way operand splits, fp32 accumulation) against the head evaluated in
    FLOAT64, next to the fp32-MFMA form on the same inputs: forward outputs, the feature gradient and the weight gradients of both forms
    sit at the distance of fp32 rounding from the exact values -- the split form is not a reduced-precision mode."""
    from oracle import popcorn_oracle as O
    from popcorn_amd import ops, _lib as L
    B, H, W, Hp, Wp, py, px = 3, 100, 100, 128, 128, 14, 14
    sd = O.load_golden_state(G)
    names = [f"head.{i}.{n}" for i in (0, 2, 4, 6) for n in ("weight", "bias")]
    work = {n: sd[n].double().clone().requires_grad_(True) for n in names}
    feat = _mk(B, 16, Hp, Wp, seed=31)
    fd = feat.double().requires_grad_(True)
    gen = torch.Generator().manual_seed(32)
    building = torch.rand(B, 1, H, W, generator=gen)
    g_pc = torch.randn(B, generator=gen)
    x = fd[:, :, py:py + H, px:px + W]
    h = x
    for i in (0, 2, 4, 6):
        h = F.conv2d(h, work[f"head.{i}.weight"], work[f"head.{i}.bias"])
        if i != 6:
            h = F.relu(h)
    scale = F.relu(h[:, 0])
    pc = (scale * building[:, 0].double()).sum((1, 2))
    (pc * g_pc.double()).sum().backward()
    ht = [sd[n].cuda() for n in names]
    rel = lambda a, r: ((a.double() - r).abs().max() / r.abs().max()).item()  # noqa: E731
    err = {}
    for form in (1, 0):
        prev = L.lib().pc_set_head_split(form)
        try:
            s_map, _, pc_gpu = ops.head_fwd(feat.cuda(), py, px, H, W, ht, building.cuda())
            grads, g_feat = ops.head_bwd(feat.cuda(), py, px, H, W, ht, building.cuda(), g_popcount=g_pc.cuda())
        finally:
            L.lib().pc_set_head_split(prev)
        err[form] = {"scale": rel(s_map.cpu(), scale.detach()), "popcount": rel(pc_gpu.cpu(), pc.detach()),
                     "g_feat": rel(g_feat.cpu(), fd.grad), "dW": max(rel(g.cpu()[:1] if n.startswith("head.6") else g.cpu(),
                                                                       work[n].grad[:1] if n.startswith("head.6") else work[n].grad)
                                                                   for n, g in zip(names, grads))}
    print(f"\n[head forms vs float64] split: {err[1]}   fp32 MFMA: {err[0]}")
    for form in (1, 0):
        assert err[form]["scale"] < 3e-6 and err[form]["popcount"] < 3e-6, err          # (fp32 epsilon = 6e-8; four layers of 16 / 64-term sums)
        assert err[form]["g_feat"] < 1e-5 and err[form]["dW"] < 1e-5, err              # (sums over 30,000 pixels)
    for k in err[1]:
        assert err[1][k] < 4 * err[0][k] + 1e-7, (k, err)                                # the same class, not merely inside the bar


@pytest.mark.parametrize("head_form", [1, 0], indirect=True)
@pytest.mark.parametrize("sparse", [True, False])
@pytest.mark.parametrize("shape", [(3, 100, 100, 128, 128, 14, 14), (1, 37, 29, 64, 64, 13, 17)])
def test_head_bwd_vs_oracle_autograd(sparse, shape, head_form):
    """All four upstream-gradient routes at once; reference = torch autograd through the oracle head."""
    from oracle import popcorn_oracle as O
    from popcorn_amd import ops
    B, H, W, Hp, Wp, py, px = shape
    sd = O.load_golden_state(G)
    names = [f"head.{i}.{n}" for i in (0, 2, 4, 6) for n in ("weight", "bias")]
    work = dict(sd)
    for n in names:
        work[n] = sd[n].clone().requires_grad_(True)
    feat = _mk(B, 16, Hp, Wp, seed=21).requires_grad_(True)
    gen = torch.Generator().manual_seed(22)
    building = torch.rand(B, 1, H, W, generator=gen)
    admin = (torch.rand(B, H, W, generator=gen) < 0.6).float() * 5.0
    census = torch.full((B,), 5, dtype=torch.int64)
    mask = (torch.rand(B, H, W, generator=gen) < 0.5) & (admin == 5.0)
    g_pc = torch.randn(B, generator=gen)
    g_pd = torch.randn(B, H, W, generator=gen) * 0.1
    g_sm = torch.randn(B, H, W, generator=gen) * 0.1
    g_const = 0.37
    headin = feat[:, :, py:py + H, px:px + W]
    if sparse:
        out = O.sparse_head_forward(work, headin, mask)[:, 0]
        selmask = mask
    else:
        out = O.head_forward(work, headin)[:, 0]
        selmask = torch.ones(B, H, W, dtype=torch.bool)
    scale = F.relu(out)
    pd = scale * building[:, 0]
    pc = (pd * (admin == census.view(-1, 1, 1))).sum((1, 2))
    loss = (pc * g_pc).sum() + (pd * g_pd).sum() + (scale * g_sm).sum() + g_const * scale[selmask].sum()
    loss.backward()
    ht = [sd[n].cuda() for n in names]
    grads, g_feat = ops.head_bwd(feat.detach().cuda(), py, px, H, W, ht, building.cuda(),
                                 mask=mask.to(torch.uint8).cuda() if sparse else None,
                                 admin_mask=admin.cuda(), census_idx=census.cuda(), g_popcount=g_pc.cuda(),
                                 g_popdense=g_pd.cuda(), g_scale_map=g_sm.cuda(),
                                 g_scale_const=torch.tensor([g_const], device="cuda"))
    for n, gr in zip(names, grads):
        ref = work[n].grad
        tol = 3e-6 * max(1.0, ref.abs().max().item())
        torch.testing.assert_close(gr.cpu(), ref, rtol=2e-5, atol=tol, msg=lambda m, n=n: f"{n}: {m}")
    torch.testing.assert_close(g_feat.cpu(), feat.grad, rtol=2e-5, atol=1e-6)
    # second output channel of head.6 gets exact zeros; run-to-run determinism
    assert torch.all(grads[6][1] == 0) and grads[7][1].item() == 0.0
    grads2, g_feat2 = ops.head_bwd(feat.detach().cuda(), py, px, H, W, ht, building.cuda(),
                                   mask=mask.to(torch.uint8).cuda() if sparse else None,
                                   admin_mask=admin.cuda(), census_idx=census.cuda(), g_popcount=g_pc.cuda(),
                                   g_popdense=g_pd.cuda(), g_scale_map=g_sm.cuda(),
                                   g_scale_const=torch.tensor([g_const], device="cuda"))
    assert all(torch.equal(a, b) for a, b in zip(grads, grads2)) and torch.equal(g_feat, g_feat2)
    # deferred reduction (PC_HEAD_BWD_DEFER_REDUCE + kind 3 of pc_wgrad_reduce_batch: the trainer's path): same sums up to the
    # association order of the partial list; = and += forms; a skipped tensor (NULL) stays untouched
    for acc in (False, True):
        tgt = [torch.full_like(t, 0.25) for t in ht]
        tgt[3] = None
        hp, g_feat3 = ops.head_bwd(feat.detach().cuda(), py, px, H, W, ht, building.cuda(),
                                   mask=mask.to(torch.uint8).cuda() if sparse else None,
                                   admin_mask=admin.cuda(), census_idx=census.cuda(), g_popcount=g_pc.cuda(),
                                   g_popdense=g_pd.cuda(), g_scale_map=g_sm.cuda(),
                                   g_scale_const=torch.tensor([g_const], device="cuda"), grads=tgt, accumulate=acc, defer_reduce=True)
        assert isinstance(hp, ops.HeadPartials) and torch.equal(g_feat3, g_feat)
        assert all(torch.all(t == 0.25) for t in tgt if t is not None)          # nothing written yet
        wb = ops.WgradBatch(torch.device("cuda"))
        wb.head_reduce(hp)
        wb.finish()
        for i, (a, b) in enumerate(zip(tgt, grads)):
            if a is None:
                continue
            want = b + 0.25 if acc else b
            if i in (6, 7) and acc:
                want = want.clone()                     # (+= leaves the structurally-zero second row alone)
                want.view(-1)[want.numel() // 2:] = 0.25
            torch.testing.assert_close(a, want, rtol=1e-5, atol=3e-6 * max(1.0, b.abs().max().item()), msg=lambda m, i=i: f"tensor {i}: {m}")
        if not acc:
            assert torch.all(tgt[6][1] == 0) and tgt[7][1].item() == 0.0


@pytest.mark.parametrize("case", ["regions", "empty_selection", "no_occupancy"])
@pytest.mark.parametrize("C_", [16, 2])
def test_building_score_mask_equals_the_two_separate_ops(case, C_):
    """pc_building_score_mask (one launch) == pc_outconv_sigmoid_crop + pc_sparsity_mask, bit for bit: building score, mask
    and {nsel, nregion}; incl. the batch-wide empty selection (popcorn.py:374-375 fallback) and repeated calls (the
    library-owned accumulator is re-zeroed per call)."""
    from popcorn_amd import ops
    B, H, W, pad = 3, 37, 52, 14
    g = torch.Generator().manual_seed(5)
    feat = torch.randn(B, C_, H + 2 * pad, W + 2 * pad, generator=g).cuda()
    w = torch.randn(C_, generator=g).cuda()
    bias = torch.randn(1, generator=g).cuda()
    admin = torch.randint(1, 4, (B, H, W), generator=g).float().cuda()
    census = torch.tensor([1, 2, 3], dtype=torch.int64).cuda()
    if case == "empty_selection":
        census = torch.tensor([7, 8, 9], dtype=torch.int64).cuda()          # no pixel belongs to the regions asked for
    rowsel = (torch.rand(H, generator=g) < 0.5).to(torch.uint8).cuda()
    colsel = (torch.rand(W, generator=g) < 0.5).to(torch.uint8).cuda()
    occ = case != "no_occupancy"
    rb = ops.outconv_sigmoid_crop(feat, w, bias, H, W, pad, pad)
    rm, rc = ops.sparsity_mask(rb, admin, census, rowsel, colsel, occ)
    for _ in range(3):
        b2, m2, c2 = ops.building_score_mask(feat, w, bias, H, W, pad, pad, admin, census, rowsel, colsel, occ)
        assert torch.equal(b2, rb) and torch.equal(m2, rm) and c2.tolist() == rc.tolist()
    if case == "empty_selection":
        assert rc.tolist() == [0, 0] and int(rm.sum()) == 0
    else:
        assert rc[0].item() == int(rm.sum()) > 0


@pytest.mark.parametrize("losses,lams", [(("l1_loss",), (1.0,)), (("log_l1_loss",), (1.0,)), (("mse_loss",), (1.0,)),
                                         (("log_mse_loss",), (1.0,)), (("l1_loss", "log_mse_loss", "mse_loss"), (0.5, 2.0, 0.01))])
def test_loss_fwd_bwd_kernel_all_losses_vs_oracle(losses, lams):
    """pc_loss_fwd_bwd for every loss the reference offers (utils/losses.py:49-56) and a weighted mix, against the oracle's
    get_loss + autograd: loss value, regulariser, d(lam_weak * loss)/d popcount and the constant regulariser gradient
    (utils/losses.py:74-76), incl. a rank-style global batch (inv_B of twice the local batch)."""
    from oracle import popcorn_oracle as O
    from popcorn_amd import ops
    from popcorn_amd.train import LOSS_INDEX
    g = np.load(os.path.join(G, "g6_loss_metrics.npz"))
    pred, y, scale = (torch.from_numpy(g[k]) for k in ("pred", "y", "scale"))
    lam_weak, sreg = 100.0, 0.01
    p = pred.clone().requires_grad_(True)
    sc = scale.clone().requires_grad_(True)
    ref, _ = O.get_loss({"popcount": p}, {"y": y}, scale=sc, loss=list(losses), lam=list(lams), scale_regularization=sreg, tag="weak")
    (ref * lam_weak).backward()
    lam4 = [0.0] * 4
    for lo, la in zip(losses, lams):
        lam4[LOSS_INDEX[lo]] += la
    B = pred.numel()
    for world in (1, 2):
        stats = torch.tensor([float(scale.numel()), float(scale.double().sum())], dtype=torch.float64, device="cuda")
        loss_out = torch.zeros(2, device="cuda")
        g_pc = torch.zeros(B, device="cuda")
        g_sc = torch.zeros(1, device="cuda")
        ops.loss_fwd_bwd(pred.cuda(), y.cuda(), stats, lam4, sreg, lam_weak, 1.0 / (world * B), loss_out, g_pc, g_sc)
        torch.cuda.synchronize()
        reg = sreg * scale.abs().mean().item()
        assert abs(loss_out[1].item() - reg) <= 1e-6 * max(1.0, reg)
        assert abs((loss_out[0].item() - reg) * world + reg - ref.item()) <= 2e-6 * max(1.0, abs(ref.item()))
        torch.testing.assert_close(g_pc.cpu() * world, p.grad, rtol=2e-6, atol=1e-7)
        assert abs(g_sc.item() - sc.grad[0].item()) <= 1e-6 * abs(sc.grad[0].item())
