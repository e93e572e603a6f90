"""bf16 mixed precision (BASELINE config 4; PC_PREC_BF16, rounding points in include/popcorn_hip.h).

The reference has no such mode, so there is no reference fixture: the HIP path is compared with the oracle's bf16
restatement (``with O.bf16_mode()``: explicit ``.to(bfloat16)`` casts at the same rounding points, stock torch ops in
between), and both are placed against the fp32 oracle.  Two fp32 evaluations of the same bf16 definition differ whenever an
fp32 sum lands within rounding of a bf16 tie (a 2^-8 = 0.4 % step on that element), so the bar cannot be 1e-4:

    TOLERANCES (measured on MI355X with tools/bf16_errors.py, two seeds; the test bound is ~2.5x the measurement)
      quantity                     HIP-bf16 vs oracle-bf16      oracle-bf16 vs oracle-fp32 (what rounding itself costs)
      popdensemap, max rel         7.6e-3   -> bound 2e-2        3.3e-2 .. 3.8e-2
      popcount, max rel            7.5e-5   -> bound 5e-4        1.0e-2
      loss, rel                    1e-5     -> bound 1e-4        5.5e-3
      gradients, worst tensor      4.7e-3   -> bound 1.5e-2      5.2e-2 .. 5.6e-2
      gradients, median tensor     9e-4     -> bound 3e-3        1.0e-2
    and every HIP-vs-oracle distance must stay below HALF the rounding band of the same quantity.
Index paths (mask, Nsel) stay exact; activations and activation gradients are bf16 CONTAINERS (torch.bfloat16 tensors, half
the HBM bytes of the fp32 mode); master weights, weight gradients and Adam stay fp32."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import popcorn_oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")

TOL_MAP, TOL_COUNT, TOL_LOSS, TOL_GRAD_WORST, TOL_GRAD_MEDIAN = 2e-2, 5e-4, 1e-4, 1.5e-2, 3e-3


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def _model(seed=1600):
    from popcorn_amd.model import POPCORN
    torch.manual_seed(seed)
    m = POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
    return m, {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}


def _sample():
    g = np.load(os.path.join(G, "g5_train.npz"))
    return {k: torch.from_numpy(g[k]) for k in ("input", "admin_mask", "census_idx", "y")}


def _is_bf16(t):
    """bf16 container, or fp32 values that are all bf16-representable"""
    if t.dtype == torch.bfloat16:
        return True
    return bool(((t.contiguous().view(torch.int32) & 0xFFFF) == 0).all())


def _dev(x):
    """activation / gradient operand of a bf16-mode op: a channels-last bf16 tensor on the GPU"""
    return x.cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("padding", [True, False])
@pytest.mark.parametrize("sparse", [True, False])
def test_bf16_forward_vs_bf16_oracle(padding, sparse):
    m, sd = _model()
    s = _sample()
    with torch.no_grad():
        torch.manual_seed(1600)
        o32 = O.popcorn_forward(sd, {k: v.clone() for k, v in s.items()}, padding=padding, sparse=sparse)
        with O.bf16_mode():
            torch.manual_seed(1600)
            in16 = {k: v.clone() for k, v in s.items()}
            o16 = O.popcorn_forward(sd, in16, padding=padding, sparse=sparse)
        m.set_precision("bf16")
        torch.manual_seed(1600)
        inp = {k: v.cuda() for k, v in s.items()}
        h16 = m(inp, padding=padding, sparse=sparse)
    assert h16["scale"].numel() == o16["scale"].numel()                              # Nsel: index path, exact
    for key, tol in (("popdensemap", TOL_MAP), ("popcount", TOL_COUNT)):
        e, band = rel(h16[key].cpu(), o16[key]), rel(o16[key], o32[key])
        assert e < tol and e < 0.5 * band, (key, e, band)
        assert band > 10 * 1e-4, "bf16 rounding must be visible against the fp32 result, else the mode is not active"
    assert rel(inp["building_counts"].cpu(), in16["building_counts"]) < TOL_MAP      # frozen extractor, same rounding points


def test_bf16_stored_activations_are_bf16_values_and_fp32_mode_is_untouched():
    from popcorn_amd import _lib as L
    m, sd = _model()
    x = _sample()["input"].cuda()
    eng = m.engines()[0]
    with torch.no_grad(), L.precision("bf16"):
        feats, saved = eng.forward(x, 14, 14, 128, 128, save=True)
    for s in ("sar_stream", "optical_stream"):
        for k in ("a1", "a2", "b1", "b2", "c1", "c2", "u2", "e1", "e2", "u1", "f1", "pa2", "pb2"):
            assert saved[s][k].dtype == torch.bfloat16, (s, k)
    assert feats.dtype == torch.bfloat16 and saved["X"].dtype == torch.float32
    assert L.lib().pc_get_precision() == L.PC_PREC_FP32                               # the context restored the mode
    with torch.no_grad():
        feats32, _ = eng.forward(x, 14, 14, 128, 128, save=False)
    assert feats32.dtype == torch.float32 and not _is_bf16(feats32)
    ref = O.dualstream_features(sd, "unetmodel", O.reorder_channels(O.add_padding(x.cpu(), True)[0]))
    assert rel(feats32.cpu(), ref) < 1e-4                                             # fp32 path unchanged by the bf16 build


def test_bf16_train_step_vs_bf16_oracle_and_fp32_master_weights():
    from popcorn_amd.train import FusedTrainStep
    m, sd = _model()
    s = _sample()
    torch.manual_seed(3)
    l32, out32, g32, _ = O.train_step_grads(sd, dict(s))
    with O.bf16_mode():
        torch.manual_seed(3)
        l16, out16, g16, _ = O.train_step_grads(sd, dict(s))
    m.set_precision("bf16")
    tr = FusedTrainStep(m, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01)
    p0 = tr.flat_p.clone()
    torch.manual_seed(3)
    loss = tr.step({k: v.cuda() for k, v in s.items()})
    torch.cuda.synchronize()
    assert abs(loss[0].item() - l16.item()) < TOL_LOSS * abs(l16.item())
    assert abs(l16.item() - l32.item()) > 10 * TOL_LOSS * abs(l32.item())
    e = rel(tr.last["popcount"].cpu(), out16["popcount"])
    assert e < TOL_COUNT and e < 0.5 * rel(out16["popcount"], out32["popcount"])
    assert int(tr.stats[0].item()) == out16["scale"].numel()                         # Nsel: index path, exact
    assert set(g16) == set(tr.grads)
    eh = {n: rel(tr.grads[n].cpu(), g16[n]) for n in g16}
    eo = {n: rel(g16[n], g32[n]) for n in g16}
    assert max(eh.values()) < TOL_GRAD_WORST and max(eh.values()) < 0.5 * max(eo.values()), max(eh, key=eh.get)
    assert float(np.median(list(eh.values()))) < TOL_GRAD_MEDIAN
    # weight gradients are fp32 sums (not bf16-rounded), master weights and Adam moments are fp32
    assert not _is_bf16(tr.flat_g) and not _is_bf16(tr.flat_p) and not _is_bf16(tr.m)
    assert not torch.equal(tr.flat_p, p0)
    # the update equals torch-style Adam on the HIP gradients (fp32 master copy; same check as the fp32 path)
    grads = {n: tr.grads[n].cpu() for n in tr.names}
    _, clipped = O.clip_grad_norm(grads, 0.01)
    new = O.adam_step(sd, clipped, {}, lr=1e-4, weight_decay=1e-5)
    table = dict(m.named_parameters())
    for n in tr.names:
        torch.testing.assert_close(table[n].detach().cpu(), new[n], rtol=0, atol=2e-7, msg=lambda t, n=n: f"{n}: {t}")


def test_bf16_training_lowers_the_loss_with_graph_replay():
    from popcorn_amd.train import FusedTrainStep
    m, _ = _model()
    m.set_precision("bf16")
    s = {k: v.cuda() for k, v in _sample().items()}
    runs = []
    for use_graph in (False, True):
        mm, _ = _model()
        mm.set_precision("bf16")
        tr = FusedTrainStep(mm, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01, use_graph=use_graph)
        losses = []
        for it in range(5):
            torch.manual_seed(50)
            losses.append(tr.step(dict(s))[0].item())
        torch.cuda.synchronize()
        runs.append((losses, tr.flat_p.clone()))
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])        # graph replay == eager, bit for bit
    assert all(np.isfinite(runs[0][0])) and runs[0][0][-1] < runs[0][0][0]


# ---- the data-parallel half of config 4 -------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_rank(rank, world, port, q):
    import torch.distributed as dist
    from popcorn_amd.distributed import FlatReducer
    from popcorn_amd.train import FusedTrainStep
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    m, _ = _model()
    m.set_precision("bf16")
    tr = FusedTrainStep(m, lr=1e-3, weight_decay=5e-7, gradient_clip=0.01, reducer=FlatReducer(), use_graph=True)   # che recipe wd
    g = np.load(os.path.join(G, "g5_train.npz"))
    full = {k: torch.from_numpy(g[k]) for k in ("input", "admin_mask", "census_idx", "y")}
    full = {k: torch.cat([v, v.flip(0)[:1]]) for k, v in full.items()}              # 4 tiles
    idx = list(range(rank, 4, world))
    s = {k: v[idx].cuda() for k, v in full.items()}
    for step in range(3):
        torch.manual_seed(100 + step)
        tr.step(s)
    torch.cuda.synchronize()
    if rank == 0:
        q.put(tr.flat_p.cpu().numpy().tolist())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_bf16_two_rank_data_parallel_equals_single_process():
    """che recipe (README.md:182: -wd 5e-7) in bf16 mode on two ranks (gloo group on the one test GPU; RCCL needs a GPU
    per rank): gradients are summed in fp32 by ONE all-reduce of the flat buffer, so three steps on the two halves of a
    batch give the single-process parameters up to the order of the fp32 sums."""
    from tests.test_gpu_dp import _get
    outs = []
    for world in (1, 2):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_dp_rank, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        outs.append(torch.tensor(_get(q, procs)))
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    p1, p2 = outs
    assert (p1 - p2).abs().max().item() <= 2e-5 * p1.abs().max().item()


# ---- op level: the bf16-MFMA conv kernels (v_mfma_f32_16x16x32_bf16) --------------------------------------------------
def _bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def _mk(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _bn(c, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1, torch.randn(c, generator=g) * 0.2,
            torch.rand(c, generator=g) + 0.3)


def _close_bf16(out, ref_fp32):
    """out must equal round_bf16(ref) up to the ties an fp32 summation order can flip: every element within one bf16 step
    (2^-7 relative, or a tiny absolute floor around zero), almost all of them exactly equal."""
    want = _bf(ref_fp32)
    assert out.dtype == torch.bfloat16
    o = out.float().cpu()
    err = (o - want).abs()
    assert bool((err <= 2.0 ** -7 * want.abs() + 1e-6).all()), err.max().item()
    assert float((err == 0).float().mean()) > 0.98


@pytest.mark.parametrize("cin,cout", [(8, 8), (16, 8), (32, 8), (8, 16), (16, 16)])      # (2 / 4 -> 8: the reflect loader, below)
@pytest.mark.parametrize("shape", [(2, 64, 64), (1, 37, 53), (3, 16, 32)])
def test_bf16_conv_fwd_op(cin, cout, shape):
    from popcorn_amd import ops, _lib as L
    import torch.nn.functional as F
    B, H, W = shape
    x = _bf(_mk(B, cin, H, W, seed=1))                     # activations arrive rounded from their producer
    w = _mk(cout, cin, 3, 3, seed=2, scale=0.2)
    b = _mk(cout, seed=3, scale=0.1)
    gamma, beta, mean, var = _bn(cout, 4)
    y = F.conv2d(x.double(), _bf(w).double(), b.double(), padding=1)
    ref = F.relu(F.batch_norm(y, mean.double(), var.double(), gamma.double(), beta.double(), training=False, eps=1e-5)).float()
    with L.precision("bf16"):
        out = ops.conv3x3_bn_relu(_dev(x), w.cuda(), b.cuda(), gamma.cuda(), beta.cuda(), mean.cuda(), var.cuda())
        with pytest.raises(RuntimeError):               # an fp32 container in bf16 mode is refused, not converted
            ops.conv3x3_bn_relu(x.cuda(), w.cuda(), b.cuda(), gamma.cuda(), beta.cuda(), mean.cuda(), var.cuda())
    _close_bf16(out, ref)


def test_bf16_conv_asymmetric_taps_and_loaders():
    """Single non-zero taps (catch any row / column / channel-slot mix-up of the K = (row, channel) packing), the fused
    pool and concat loaders, the reflect loader with un-rounded input, and the dgrad epilogues, all in bf16 mode."""
    from popcorn_amd import ops, _lib as L
    import torch.nn.functional as F
    x = _bf(_mk(1, 16, 20, 40, seed=5))
    with L.precision("bf16"):
        for (co, ci, dy, dx) in [(1, 6, 0, 2), (7, 0, 2, 0), (3, 11, 1, 0), (0, 15, 2, 2), (5, 8, 0, 0)]:
            w = torch.zeros(8, 16, 3, 3)
            w[co, ci, dy, dx] = 1.0
            ref = F.conv2d(x, w, None, padding=1)
            out = ops.conv3x3_bn_relu(_dev(x), w.cuda(), None, relu=False)
            assert torch.equal(out.float().cpu(), ref), (co, ci, dy, dx)
        # pool loader
        xs = _bf(_mk(2, 8, 64, 64, seed=6))
        w = _mk(16, 8, 3, 3, seed=7, scale=0.2)
        b = _mk(16, seed=8, scale=0.1)
        ref = F.relu(F.conv2d(F.max_pool2d(xs, 2).double(), _bf(w).double(), b.double(), padding=1)).float()
        out = ops.conv3x3_bn_relu(_dev(xs), w.cuda(), b.cuda(), a_mode=L.PC_SRC_POOL2)
        _close_bf16(out, ref)
        # concat loader (aligned: staged path) and with an offset up tensor (generic path)
        for (hs, ws, hu, wu) in [(32, 64, 32, 64), (23, 35, 22, 34)]:
            skip, upt = _bf(_mk(2, 16, hs, ws, seed=10)), _bf(_mk(2, 16, hu, wu, seed=11))
            w = _mk(8, 32, 3, 3, seed=12, scale=0.1)
            b = _mk(8, seed=13, scale=0.1)
            dy, dx = hs - hu, ws - wu
            upp = F.pad(upt, (dx // 2, dx - dx // 2, dy // 2, dy - dy // 2))
            ref = F.relu(F.conv2d(torch.cat([skip, upp], 1).double(), _bf(w).double(), b.double(), padding=1)).float()
            out = ops.conv3x3_bn_relu(_dev(skip), w.cuda(), b.cuda(), b=_dev(upt), b_offset=(dy // 2, dx // 2))
            _close_bf16(out, ref)
        # reflect loader: un-rounded 6-channel input, channel gather
        X = _mk(2, 6, 100, 100, seed=14)
        w = _mk(8, 4, 3, 3, seed=15, scale=0.3)
        b = _mk(8, seed=16, scale=0.1)
        chmap = (2, 1, 0, 3)
        xp = F.pad(X[:, list(chmap)], (14, 14, 14, 14), mode="reflect")
        ref = F.relu(F.conv2d(_bf(xp).double(), _bf(w).double(), b.double(), padding=1)).float()
        out = ops.conv3x3_raw(X.cuda(), w.cuda(), L.bn(b.cuda()), a_mode=L.PC_SRC_REFLECT, a_pad=(14, 14), chmap=chmap,
                              out_hw=(128, 128), a_channels=4)
        _close_bf16(out, ref)
        # data gradient with the ReLU / BN epilogue of the producing layer
        g = _bf(_mk(2, 8, 32, 64, seed=17))
        w = _mk(8, 16, 3, 3, seed=18, scale=0.2)
        act = _bf(F.relu(_mk(2, 8, 32, 64, seed=19)))
        gamma, beta, mean, var = _bn(8, 20)
        scale = gamma / torch.sqrt(var + 1e-5)
        full = F.conv_transpose2d(g.double(), _bf(w).double(), padding=1)[:, 8:16]
        ref = (full * (act > 0) * scale.view(1, 8, 1, 1).double()).float()
        out = L.empty_act(2, 8, 32, 64, "cuda")
        ops.conv3x3_dgrad(_dev(g), w.cuda(), 8, 8, out, act=_dev(act), act_bn=L.bn(None, gamma.cuda(), beta.cuda(), mean.cuda(), var.cuda()))
        _close_bf16(out, ref)


@pytest.mark.parametrize("cin,cout", [(8, 8), (16, 8), (32, 8), (8, 16), (16, 16)])
@pytest.mark.parametrize("shape", [(2, 64, 64), (3, 16, 32), (1, 40, 52), (2, 128, 128), (1, 37, 53)])
def test_bf16_conv_wgrad_op(cin, cout, shape):
    """Weight / bias gradient in bf16 mode (v_mfma_f32_16x16x32_bf16 with the reduction over a whole strip row on K, the
    horizontal tap on the gradient operand): bf16 operands, fp32 sums -- so against torch on the same rounded operands it
    is an fp32-accuracy check (<= 2e-5 of the largest entry), and the result is NOT bf16-rounded."""
    from popcorn_amd import ops, _lib as L
    import torch.nn.functional as F
    B, H, W = shape
    x = _bf(_mk(B, cin, H, W, seed=40))
    w = _mk(cout, cin, 3, 3, seed=41, scale=0.2).double().requires_grad_(True)
    bias = torch.zeros(cout, dtype=torch.double, requires_grad=True)
    g = _bf(_mk(B, cout, H, W, seed=42))
    F.conv2d(x.double(), w, bias, padding=1).backward(g.double())
    with L.precision("bf16"):
        dw, db = ops.conv3x3_wgrad(_dev(x), _dev(g), cout)
        dw2, db2 = ops.conv3x3_wgrad(_dev(x), _dev(g), cout)
    sw, sb = w.grad.abs().max().item(), bias.grad.abs().max().item()
    assert (dw.cpu().double() - w.grad).abs().max().item() <= 2e-5 * sw
    assert (db.cpu().double() - bias.grad).abs().max().item() <= 2e-5 * max(sb, 1.0)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)                            # deterministic
    assert dw.dtype == torch.float32 and not _is_bf16(dw)


@pytest.mark.parametrize("cin", [2, 4])
def test_bf16_first_layer_wgrad_reflect_loader(cin):
    """First layers: planar fp32 model input through the reflect loader (rounded to bf16 when staged), channels-last bf16
    gradient, 2 / 4 input channels with a channel gather."""
    from popcorn_amd import ops, _lib as L
    import torch.nn.functional as F
    X = _mk(2, 6, 100, 100, seed=60)
    chmap = (4, 5, 0, 0) if cin == 2 else (2, 1, 0, 3)
    xp = _bf(F.pad(X[:, list(chmap[:cin])], (14, 14, 14, 14), mode="reflect"))
    w = _mk(8, cin, 3, 3, seed=61, scale=0.3).double().requires_grad_(True)
    bias = torch.zeros(8, dtype=torch.double, requires_grad=True)
    g = _bf(_mk(2, 8, 128, 128, seed=62))
    F.conv2d(xp.double(), w, bias, padding=1).backward(g.double())
    with L.precision("bf16"):
        dw, db = ops.conv3x3_wgrad(X.cuda(), _dev(g), 8, a_mode=L.PC_SRC_REFLECT, a_pad=(14, 14), chmap=chmap, a_channels=cin)
    assert (dw.cpu().double() - w.grad).abs().max().item() <= 2e-5 * w.grad.abs().max().item()
    assert (db.cpu().double() - bias.grad).abs().max().item() <= 2e-5 * bias.grad.abs().max().item()


def test_bf16_conv_wgrad_pool_and_concat_loaders():
    from popcorn_amd import ops, _lib as L
    import torch.nn.functional as F
    with L.precision("bf16"):
        xs = _bf(_mk(2, 8, 64, 128, seed=43))
        w = _mk(16, 8, 3, 3, seed=44, scale=0.2).double().requires_grad_(True)
        y = F.conv2d(F.max_pool2d(xs, 2).double(), w, None, padding=1)
        g = _bf(_mk(*y.shape, seed=45))
        y.backward(g.double())
        dw, _ = ops.conv3x3_wgrad(_dev(xs), _dev(g), 16, a_mode=L.PC_SRC_POOL2)
        assert (dw.cpu().double() - w.grad).abs().max().item() <= 2e-5 * w.grad.abs().max().item()
        skip, upt = _bf(_mk(2, 16, 32, 64, seed=46)), _bf(_mk(2, 16, 32, 64, seed=47))
        w = _mk(8, 32, 3, 3, seed=48, scale=0.1).double().requires_grad_(True)
        y = F.conv2d(torch.cat([skip, upt], 1).double(), w, None, padding=1)
        g = _bf(_mk(*y.shape, seed=49))
        y.backward(g.double())
        dw, _ = ops.conv3x3_wgrad(_dev(skip), _dev(g), 8, b=_dev(upt), b_offset=(0, 0))
        assert (dw.cpu().double() - w.grad).abs().max().item() <= 2e-5 * w.grad.abs().max().item()
        # an up-sampled half that is smaller than the skip tensor (Up's zero F.pad): placement offset, odd sizes
        skip, upt = _bf(_mk(2, 16, 23, 35, seed=50)), _bf(_mk(2, 16, 22, 34, seed=51))
        w = _mk(8, 32, 3, 3, seed=52, scale=0.1).double().requires_grad_(True)
        y = F.conv2d(torch.cat([skip, F.pad(upt, (0, 1, 0, 1))], 1).double(), w, None, padding=1)
        g = _bf(_mk(*y.shape, seed=53))
        y.backward(g.double())
        dw, _ = ops.conv3x3_wgrad(_dev(skip), _dev(g), 8, b=_dev(upt), b_offset=(0, 0))
        assert (dw.cpu().double() - w.grad).abs().max().item() <= 2e-5 * w.grad.abs().max().item()


# ---- op level: the channels-last bf16 transposed-conv kernels ----------------------------------------------------------
@pytest.mark.parametrize("C_", [8, 16])
@pytest.mark.parametrize("shape", [(2, 32, 32), (1, 11, 19), (3, 16, 48)])
def test_bf16_convt_fwd_dgrad_wgrad_op(C_, shape):
    from popcorn_amd import ops, _lib as L
    import torch.nn.functional as F
    B, H, W = shape
    x = _bf(_mk(B, C_, H, W, seed=70))
    w = _mk(C_, C_, 2, 2, seed=71, scale=0.3)
    bias = _mk(C_, seed=72, scale=0.1)
    xd = x.double().requires_grad_(True)
    wd = _bf(w).double().requires_grad_(True)
    bd = bias.double().requires_grad_(True)
    y = F.conv_transpose2d(xd, wd, bd, stride=2)
    g = _bf(_mk(*y.shape, seed=73))
    y.backward(g.double())
    act = _bf(F.relu(_mk(B, C_, H, W, seed=74)))
    gamma, beta, mean, var = _bn(C_, 75)
    scale = gamma / torch.sqrt(var + 1e-5)
    with L.precision("bf16"):
        out = ops.convt2x2(_dev(x), w.cuda(), bias.cuda())
        _close_bf16(out, y.detach().float())
        gx = L.empty_act(B, C_, H, W, "cuda")
        ops.convt2x2_dgrad(_dev(g), w.cuda(), gx)
        _close_bf16(gx, xd.grad.float())
        gx2 = L.empty_act(B, C_, H, W, "cuda")
        ops.convt2x2_dgrad(_dev(g), w.cuda(), gx2, act=_dev(act), act_bn=L.bn(None, gamma.cuda(), beta.cuda(), mean.cuda(), var.cuda()))
        _close_bf16(gx2, (xd.grad * (act > 0) * scale.view(1, C_, 1, 1).double()).float())
        dw, db = ops.convt2x2_wgrad(_dev(x), _dev(g))
        dw2, db2 = ops.convt2x2_wgrad(_dev(x), _dev(g))
    assert (dw.cpu().double() - wd.grad).abs().max().item() <= 2e-5 * wd.grad.abs().max().item()
    assert (db.cpu().double() - bd.grad).abs().max().item() <= 2e-5 * bd.grad.abs().max().item()
    assert torch.equal(dw, dw2) and torch.equal(db, db2)


# ---- op level: the bf16 head kernels (channels-last feature / gradient maps) -------------------------------------------
@pytest.mark.parametrize("sparse", [True, False, "few"])
@pytest.mark.parametrize("shape", [(3, 100, 100, 128, 128, 14, 14), (1, 37, 29, 64, 64, 13, 17), (40, 100, 100, 128, 128, 14, 14)])
def test_bf16_head_fwd_bwd_vs_bf16_oracle_autograd(sparse, shape):
    """Head forward and backward in bf16 mode against torch autograd through the oracle head with the bf16 rounding points;
    all four upstream-gradient routes.  The B = 40 case (25,000 pixel groups) runs the cooperative backward kernel through
    several rounds of its workgroup exchange (more groups than 256 workgroups x 8 waves)."""
    from popcorn_amd import ops, _lib as L
    import torch.nn.functional as F
    B, H, W, Hp, Wp, py, px = shape
    sd = O.load_golden_state(G)
    names = [f"head.{i}.{n}" for i in (0, 2, 4, 6) for n in ("weight", "bias")]
    work = dict(sd)
    for n in names:
        work[n] = sd[n].clone().requires_grad_(True)
    feat = _bf(_mk(B, 16, Hp, Wp, seed=21)).requires_grad_(True)             # the feature map arrives rounded from its producers
    gen = torch.Generator().manual_seed(22)
    building = torch.rand(B, 1, H, W, generator=gen)
    admin = (torch.rand(B, H, W, generator=gen) < 0.6).float() * 5.0
    census = torch.full((B,), 5, dtype=torch.int64)
    mask = (torch.rand(B, H, W, generator=gen) < 0.5) & (admin == 5.0)
    if sparse == "few":          # a handful of selected pixels: most 16-pixel groups (and whole exchange rounds) contribute nothing
        mask = mask & (torch.rand(B, H, W, generator=gen) < 0.004)
    g_pc = torch.randn(B, generator=gen)
    g_pd = torch.randn(B, H, W, generator=gen) * 0.1
    g_sm = torch.randn(B, H, W, generator=gen) * 0.1
    g_const = 0.37
    headin = feat[:, :, py:py + H, px:px + W]
    with O.bf16_mode():
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
    kw = dict(mask=mask.to(torch.uint8).cuda() if sparse else None, admin_mask=admin.cuda(), census_idx=census.cuda())
    with L.precision("bf16"):
        s_map, pd_gpu, pc_gpu = ops.head_fwd(_dev(feat.detach()), py, px, H, W, ht, building.cuda(), **kw)
        grads, g_feat = ops.head_bwd(_dev(feat.detach()), py, px, H, W, ht, building.cuda(), g_popcount=g_pc.cuda(),
                                     g_popdense=g_pd.cuda(), g_scale_map=g_sm.cuda(),
                                     g_scale_const=torch.tensor([g_const], device="cuda"), **kw)
        grads2, g_feat2 = ops.head_bwd(_dev(feat.detach()), py, px, H, W, ht, building.cuda(), g_popcount=g_pc.cuda(),
                                       g_popdense=g_pd.cuda(), g_scale_map=g_sm.cuda(),
                                       g_scale_const=torch.tensor([g_const], device="cuda"), **kw)
    # forward: fp32 evaluation of the same bf16 definition -- differences only where an fp32 sum sits on a bf16 tie
    assert rel(s_map.cpu(), scale.detach()) < TOL_MAP and rel(pc_gpu.cpu(), pc.detach()) < 5 * TOL_COUNT
    assert g_feat.dtype == torch.bfloat16 and g_feat.is_contiguous(memory_format=torch.channels_last)
    worst = {n: rel(gr.cpu(), work[n].grad) for n, gr in zip(names, grads)}
    assert max(worst.values()) < TOL_GRAD_WORST, worst
    gf = g_feat.float().cpu()
    # the gradient map is rounded per element: a bf16 step (2^-7) on its largest entries, and a rare ReLU-mask flip of a hidden
    # unit at a pixel, bound the maximum; almost all elements agree to two bf16 steps
    assert rel(gf, feat.grad) < 2 * TOL_MAP
    off = (gf - feat.grad).abs() > 2.0 ** -6 * feat.grad.abs() + 1e-4 * feat.grad.abs().max()
    assert off.float().mean().item() < 1e-3
    assert torch.equal(gf[:, :, :py], torch.zeros_like(gf[:, :, :py])) and torch.equal(gf[:, :, :, :px], torch.zeros_like(gf[:, :, :, :px]))
    assert torch.equal(gf[:, :, py + H:], torch.zeros_like(gf[:, :, py + H:])) and torch.equal(gf[:, :, :, px + W:], torch.zeros_like(gf[:, :, :, px + W:]))
    assert torch.all(grads[6][1] == 0) and grads[7][1].item() == 0.0
    assert all(torch.equal(a, b) for a, b in zip(grads, grads2)) and torch.equal(g_feat, g_feat2)       # deterministic


# ---- other tile geometries than 100 x 100 through the whole bf16 train step ---------------------------------------------
@pytest.mark.parametrize("geom", [(2, 64, 48, "full"), (3, 37, 53, "disc"), (1, 131, 77, "disc"), (2, 32, 32, "full")])
def test_bf16_train_step_other_geometries(geom):
    """Ragged / small tiles take the partial-strip paths of every channels-last kernel (bounds predicates, fallback pooling
    loader when the pooled second output does not qualify, placement offsets of the up-sampled halves).  Loss and census
    counts against the bf16 oracle at the bounds of this file; every gradient tensor within TOL_GRAD_WORST or, where a few
    hundred selected pixels make single ReLU-tie flips visible, well inside the band bf16 rounding itself spans."""
    from popcorn_amd.train import FusedTrainStep
    from popcorn_amd.data.synthetic import make_raw_batch
    B, H, W, region = geom
    m, sd = _model()
    batch = make_raw_batch(B, H, W, seed=7, region=region)
    x = O.select_normalize(batch["raw"])
    s = {"input": x, "admin_mask": batch["admin_mask"], "census_idx": batch["census_idx"], "y": batch["y"]}
    torch.manual_seed(3)
    l32, out32, g32, _ = O.train_step_grads(sd, dict(s))
    with O.bf16_mode():
        torch.manual_seed(3)
        l16, out16, g16, _ = O.train_step_grads(sd, dict(s))
    m.set_precision("bf16")
    tr = FusedTrainStep(m, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01)
    torch.manual_seed(3)
    loss = tr.step({k: v.cuda() for k, v in s.items()})
    torch.cuda.synchronize()
    assert abs(loss[0].item() - l16.item()) < 2 * TOL_LOSS * max(1.0, abs(l16.item()))
    assert rel(tr.last["popcount"].cpu(), out16["popcount"]) < 2 * TOL_COUNT
    assert int(tr.stats[0].item()) == out16["scale"].numel()
    for n in g16:
        e, band = rel(tr.grads[n].cpu(), g16[n]), rel(g16[n], g32[n])
        assert e < max(TOL_GRAD_WORST, 0.75 * band), (n, e, band)


def test_bf16_che_recipe_through_the_clis(tmp_path):
    """BASELINE config 4's recipe flags (README.md:182: -wd 5e-7 --biasinit 0.2267) through the run_train / run_eval
    counterparts with --precision bf16: the fused bf16 step trains (finite, logged, checkpoint in the reference's format with
    fp32 parameters), validation + the in-training target test run in bf16, and run_eval stitches a raster from the checkpoint."""
    import json as _json
    from popcorn_amd import cli
    from popcorn_amd.cli import Trainer, train_parser
    base = ("-S2 -NIR -S1 -occmodel -senbuilds -pret -wd 5e-7 --biasinit 0.2267 --synthetic_regions 8 -wb 4 "
            f"--save_dir {tmp_path} -lt 1 -wv -val 1 --fixed_hw 64 64 --precision bf16 -e 2").split()
    t = Trainer(train_parser().parse_args(base))
    assert t.model.precision == "bf16"
    t.train()
    recs = [_json.loads(l) for l in open(os.path.join(t.exp, "train_log.jsonl"))]
    losses = [r["loss"] for r in recs if "loss" in r and "iter" in r]
    assert len(losses) >= 2 and all(np.isfinite(losses))
    assert any(any(k.endswith("/val") for k in r) for r in recs) and any(any(k.endswith("/targettest") for k in r) for r in recs)
    ck = os.path.join(t.exp, "last_model.pth")
    d = torch.load(ck, weights_only=False)
    assert all(v.dtype in (torch.float32, torch.int64) for v in d["model"].values())          # master weights stay fp32
    res = cli.run_eval(("-S2 -NIR -S1 -occmodel -senbuilds -pret --biasinit 0.2267 --raster_hw 300 420 --patchsize 256 --overlap 32 "
                        f"--seed 1600 --precision bf16 --save_dir {tmp_path} -r {ck}").split())
    assert np.isfinite(res["Population_MainCensus_synthetic_fine/r2"]) and np.isfinite(res["Population_AdjCensus_synthetic_fine/l1_loss"])


def test_bf16_sliding_window_inference_tracks_fp32():
    """BASELINE config 5's path in bf16 mode: the stitched population map of a small raster (overlapping windows, two ensemble
    members) stays inside the bf16 rounding band of the fp32 map; window bookkeeping (count map = finite everywhere) unchanged."""
    from popcorn_amd.eval import evaluate_raster
    from popcorn_amd.model import POPCORN
    torch.manual_seed(1600)
    models = [POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda().eval()
              for _ in range(2)]
    g = torch.Generator().manual_seed(5)
    raster = torch.randn(1, 6, 600, 700, generator=g).cuda()
    outs = {}
    for prec in ("fp32", "bf16"):
        for m in models:
            m.set_precision(prec)
        mean, std, smean, sstd = evaluate_raster(models, raster, patchsize=256, overlap=32)
        outs[prec] = (mean.cpu(), smean.cpu())
    for a, b in zip(outs["bf16"], outs["fp32"]):
        assert torch.isfinite(a).all()
        assert rel(a, b) < 5e-2                                                     # per-pixel bf16 rounding of a deep chain
        assert ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item() < 3e-2


@pytest.mark.parametrize("shape", [(2, 64, 64), (1, 37, 53), (3, 128, 128)])
@pytest.mark.parametrize("case", ["plain_masked", "concat_up_half", "concat_skip_half_accumulate", "g8_x16_up_half_of_32", "g16_x16_masked"])
def test_bf16_conv_backward_fused_op(shape, case):
    """pc_conv3x3_bwd_group: data gradient + weight / bias gradient of a conv layer (8 / 16 output channels, an 8- or 16-channel
    column block of its input) in one launch, against torch on the same rounded operands; incl. column blocks of concat layers
    with a placement offset of the up-sampled half and an accumulating data gradient."""
    from popcorn_amd import ops, _lib as L
    import torch.nn.functional as F
    B, H, W = shape
    gc, xc, cin_total, c0 = {"plain_masked": (8, 8, 8, 0), "concat_up_half": (8, 8, 16, 8), "concat_skip_half_accumulate": (8, 8, 16, 0),
                             "g8_x16_up_half_of_32": (8, 16, 32, 16), "g16_x16_masked": (16, 16, 16, 0)}[case]
    up_half = "up_half" in case
    hx, wx = (H - 1, W - 1) if up_half else (H, W)          # the up-sampled half is smaller: zero F.pad, offset (0, 0)
    x = _bf(F.relu(_mk(B, xc, hx, wx, seed=80)))
    xfull = F.pad(x, (0, W - wx, 0, H - hx))
    w = _mk(gc, cin_total, 3, 3, seed=81, scale=0.2)
    g = _bf(_mk(B, gc, H, W, seed=82))
    gamma, beta, mean, var = _bn(xc, 83)
    scale = gamma / torch.sqrt(var + 1e-5)
    wd = _bf(w).double().requires_grad_(True)
    xd = torch.zeros(B, cin_total, H, W, dtype=torch.double)
    xd[:, c0:c0 + xc] = xfull.double()
    xd.requires_grad_(True)
    bias = torch.zeros(gc, dtype=torch.double, requires_grad=True)
    F.conv2d(xd, wd, bias, padding=1).backward(g.double())
    gx = xd.grad[:, c0:c0 + xc]
    masked = not up_half
    ref = (gx * (xfull > 0) * scale.view(1, xc, 1, 1).double()) if masked else gx
    prev = _bf(_mk(B, xc, H, W, seed=84))
    acc = case == "concat_skip_half_accumulate"
    with L.precision("bf16"):
        out = _dev(prev) if acc else L.empty_act(B, xc, H, W, "cuda")
        dw = torch.full((gc, cin_total, 3, 3), 7.0, device="cuda")
        db = torch.empty(gc, device="cuda")
        wb = ops.WgradBatch(torch.device("cuda"))
        wb.conv3x3_bwd_group([{"g": _dev(g), "x": _dev(x), "w": w.cuda(), "out": out, "dw": dw, "db": db,
                               "x_bn": L.bn(None, gamma.cuda(), beta.cuda(), mean.cuda(), var.cuda()) if masked else None}],
                             cin_total, c0, accumulate=acc)
        wb.finish()
    _close_bf16(out, (ref + (prev.double() if acc else 0)).float())
    gw = wd.grad[:, c0:c0 + xc]
    assert (dw[:, c0:c0 + xc].cpu().double() - gw).abs().max().item() <= 2e-5 * gw.abs().max().item()
    other = [c for c in range(cin_total) if not c0 <= c < c0 + xc]
    assert bool((dw[:, other] == 7.0).all())                                    # the other column blocks are not touched
    assert (db.cpu().double() - bias.grad).abs().max().item() <= 2e-5 * bias.grad.abs().max().item()


@pytest.mark.parametrize("shape", [(2, 64, 64), (1, 37, 53), (3, 50, 50)])
@pytest.mark.parametrize("xc", [8, 16])
def test_bf16_conv_backward_fused_pool_op(shape, xc):
    """pc_conv3x3_bwd_group with pool_act (the first conv of a Down block, networks.py:289 MaxPool2d(2) -> double_conv): x is the
    saved pooled map, the data gradient is scattered (+=) to the first arg-max of every 2x2 window of the full-resolution
    gradient with the producer's ReLU / BN factor; against torch autograd through max_pool2d on the same rounded operands
    (ties between window elements are made impossible by construction: distinct bf16 values per window)."""
    from popcorn_amd import ops, _lib as L
    import torch.nn.functional as F
    B, H2, W2 = shape                                        # full resolution (odd sizes: the last row / column is not pooled)
    H, W = H2 // 2, W2 // 2
    gc = 16
    act = _bf(F.relu(_mk(B, xc, H2, W2, seed=90)))
    # distinct positive values inside every window where any is positive, so that torch's and the kernel's arg-max agree
    bump = torch.tensor([[0.0, 1.0], [2.0, 3.0]]).repeat((H2 + 1) // 2, (W2 + 1) // 2)[:H2, :W2] / 64
    act = _bf(torch.where(act > 0, act + bump, act))
    for _ in range(3):                                       # re-round until the bumps survive bf16 rounding as distinct values
        win = F.unfold(act[:, :, :2 * H, :2 * W].reshape(B * xc, 1, 2 * H, 2 * W), 2, stride=2)
        srt = win.sort(dim=1).values
        if not bool(((srt[:, 1:] == srt[:, :-1]) & (srt[:, 1:] > 0)).any()):
            break
        act = _bf(torch.where(act > 0, act * 1.5 + bump, act))
    pooled = F.max_pool2d(act, 2)
    w = _mk(gc, xc, 3, 3, seed=91, scale=0.2)
    g = _bf(_mk(B, gc, H, W, seed=92))
    gamma, beta, mean, var = _bn(xc, 93)
    scale = gamma / torch.sqrt(var + 1e-5)
    ad = act.double().requires_grad_(True)
    wd = _bf(w).double().requires_grad_(True)
    bias = torch.zeros(gc, dtype=torch.double, requires_grad=True)
    F.conv2d(F.max_pool2d(ad, 2), wd, bias, padding=1).backward(g.double())
    ref = ad.grad * (act > 0) * scale.view(1, xc, 1, 1).double()
    prev = _bf(_mk(B, xc, H2, W2, seed=94))
    with L.precision("bf16"):
        out = _dev(prev)
        dw = torch.empty(gc, xc, 3, 3, device="cuda")
        db = torch.empty(gc, device="cuda")
        wb = ops.WgradBatch(torch.device("cuda"))
        wb.conv3x3_bwd_group([{"g": _dev(g), "x": _dev(pooled), "w": w.cuda(), "out": out, "dw": dw, "db": db, "pool_act": _dev(act),
                               "x_bn": L.bn(None, gamma.cuda(), beta.cuda(), mean.cuda(), var.cuda())}], xc, 0)
        wb.finish()
    _close_bf16(out, (ref + prev.double()).float())
    assert (dw.cpu().double() - wd.grad).abs().max().item() <= 2e-5 * wd.grad.abs().max().item()
    assert (db.cpu().double() - bias.grad).abs().max().item() <= 2e-5 * bias.grad.abs().max().item()


@pytest.mark.parametrize("C_", [8, 16])
@pytest.mark.parametrize("shape", [(2, 32, 32), (3, 9, 21), (1, 40, 16)])
def test_bf16_convt_backward_fused_op(C_, shape):
    """pc_convt2x2_bwd_group in bf16 mode: data gradient (masked by x's producer) and weight / bias gradient of a transposed conv
    from one pass over x and g, against torch on the same rounded operands (any geometry: ragged rows included)."""
    from popcorn_amd import ops, _lib as L
    import torch.nn.functional as F
    B, H, W = shape
    x = _bf(F.relu(_mk(B, C_, H, W, seed=71)))
    w = _mk(C_, C_, 2, 2, seed=72, scale=0.3)
    g = _bf(_mk(B, C_, 2 * H, 2 * W, seed=73))
    gamma, beta, mean, var = _bn(C_, 74)
    scale = gamma / torch.sqrt(var + 1e-5)
    xd = x.double().requires_grad_(True)
    wd = _bf(w).double().requires_grad_(True)
    bias = torch.zeros(C_, dtype=torch.double, requires_grad=True)
    F.conv_transpose2d(xd, wd, bias, stride=2).backward(g.double())
    ref = xd.grad * (x > 0) * scale.view(1, C_, 1, 1).double()
    with L.precision("bf16"):
        out = L.empty_act(B, C_, H, W, "cuda")
        dw, db = torch.empty(C_, C_, 2, 2, device="cuda"), torch.empty(C_, device="cuda")
        wb = ops.WgradBatch(torch.device("cuda"))
        wb.convt2x2_bwd_group([{"x": _dev(x), "g": _dev(g), "w": w.cuda(), "out": out, "dw": dw, "db": db,
                                "x_bn": L.bn(None, gamma.cuda(), beta.cuda(), mean.cuda(), var.cuda())}])
        wb.finish()
    _close_bf16(out, ref.float())
    assert (dw.cpu().double() - wd.grad).abs().max().item() <= 2e-5 * wd.grad.abs().max().item()
    assert (db.cpu().double() - bias.grad).abs().max().item() <= 2e-5 * bias.grad.abs().max().item()


def test_bf16_training_tracks_fp32_training_over_200_steps():
    """200 fused steps from the same seed on the same teacher-labelled census batches (tests/bf16_quality.py): both runs learn
    (loss / 3 or better, R^2 from < 0 to > 0.9) and the bf16 run follows the fp32 run.  Single epochs of EITHER run scatter around
    their trend (8 batches per epoch, clip 0.01 under Adam: an epoch 50-100 % above its neighbours every ten or so, fp32 and bf16
    alike, at different places), so the comparison is on MEDIANS over the last 10 epochs and inside that scatter: loss within 35 %
    (recorded: fp32 0.0200, bf16 0.0242; the fp32 run with another summation order lands 0.0196 - 0.0237, and 0.0261 with round 5's
    parallel popcount sums -- a quarter of the first epoch's 0.1025 is 0.0256, inside that scatter, hence a third), R^2 within 0.03
    (0.963 vs 0.960; table in DESIGN_HISTORY.md section 7).  What it rules out is a bf16 run that stalls, diverges or converges elsewhere."""
    from tests.bf16_quality import run
    r = run(steps=200)
    l32, l16 = r["loss_median_last_10_epochs"]["fp32"], r["loss_median_last_10_epochs"]["bf16"]
    assert 3 * l32 < r["loss_first_epoch"]["fp32"] and 3 * l16 < r["loss_first_epoch"]["bf16"], r       # both learn
    assert abs(l16 - l32) <= 0.35 * l32, (l16, l32)
    q32, q16 = r["r2_median_last_10_epochs"]["fp32"], r["r2_median_last_10_epochs"]["bf16"]
    assert q32 > 0.9 and q16 > 0.9 and abs(q16 - q32) <= 0.03, r
    assert r["relative_param_distance"] < 0.5, r          # the two runs end closer to each other than either moved from the start


# ---- the gradient-truncation regimes, the autograd node on an odd padded geometry and the unfused backward, in bf16 mode --------
def _bf16_grad_gate(got, g16, g32, where):
    """The gate of test_bf16_train_step_other_geometries: every tensor within TOL_GRAD_WORST of the bf16 oracle or well inside the
    band bf16 rounding itself spans (bf16 oracle vs fp32 oracle), and the median over tensors under TOL_GRAD_MEDIAN."""
    assert set(got) == set(g16), (where, set(got) ^ set(g16))
    errs = []
    for n in g16:
        e, band = rel(got[n].cpu(), g16[n]), rel(g16[n], g32[n])
        errs.append(e)
        assert e < max(TOL_GRAD_WORST, 0.75 * band), (where, n, e, band)
    assert float(np.median(errs)) < TOL_GRAD_MEDIAN, (where, float(np.median(errs)))


@pytest.mark.parametrize("flags", [dict(encoder_no_grad=True), dict(encoder_no_grad=True, unet_no_grad=True)])
def test_bf16_train_step_truncation_regimes(flags):
    """limit1 / limit2 (run_train.py:191-198) in bf16 mode: with the encoder frozen the first convs of the Up blocks take the
    separate weight-gradient + data-gradient launches instead of the fused backward, and the transposed convs their
    weight-gradient-only form (engine.py: backward); with the whole U-Net frozen only the head is differentiated."""
    from popcorn_amd.train import FusedTrainStep
    m, sd = _model()
    s = _sample()
    torch.manual_seed(3)
    _, _, g32, _ = O.train_step_grads(sd, dict(s), **flags)
    with O.bf16_mode():
        torch.manual_seed(3)
        l16, out16, g16, _ = O.train_step_grads(sd, dict(s), **flags)
    m.set_precision("bf16")
    tr = FusedTrainStep(m, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01)
    p0 = tr.flat_p.clone()
    torch.manual_seed(3)
    loss = tr.step({k: v.cuda() for k, v in s.items()}, **flags)
    torch.cuda.synchronize()
    assert abs(loss[0].item() - l16.item()) < TOL_LOSS * abs(l16.item())
    assert rel(tr.last["popcount"].cpu(), out16["popcount"]) < TOL_COUNT
    _bf16_grad_gate({n: tr.grads[n] for n in g16}, g16, g32, str(flags))
    # frozen groups: gradient zero, parameters untouched by the step (torch.optim.Adam skips .grad is None)
    params = dict(m.named_parameters())
    frozen = [n for n in tr.names if n not in g16]
    assert frozen
    for n in frozen:
        assert float(tr.grads[n].abs().max()) == 0.0, n
        assert torch.equal(params[n].detach().cpu(), sd[n]), n
    assert not torch.equal(tr.flat_p, p0)


def test_bf16_autograd_node_padding_true_on_an_odd_geometry():
    """model.forward(padding=True) + loss.backward() through _PopcornFn in bf16 mode on a tile whose padded extent is not a
    multiple of 4 (90 x 86 -> 118 x 114: 59 x 57 at the second level, odd pooling windows, ragged strips of the fused backward and
    the max-pool scatter kernels)."""
    from popcorn_amd.utils.losses import get_loss
    from popcorn_amd.data.synthetic import make_raw_batch
    m, sd = _model()
    batch = make_raw_batch(2, 90, 86, seed=11, region="disc")
    x = O.select_normalize(batch["raw"])
    s = {"input": x, "admin_mask": batch["admin_mask"], "census_idx": batch["census_idx"], "y": batch["y"]}

    def oracle_grads():
        work = dict(sd)
        names = O.trainable_names(sd)
        for n in names:
            work[n] = sd[n].detach().clone().requires_grad_(True)
        torch.manual_seed(9)
        out = O.popcorn_forward(work, dict(s), padding=True, sparse=True)
        l, _ = O.get_loss(out, s, scale=out["scale"], loss=("log_l1_loss",), lam=(1.0,), scale_regularization=0.01, tag="weak")
        (l * 100.0).backward()
        return l.detach(), out, {n: work[n].grad for n in names if work[n].grad is not None}

    l32, _, g32 = oracle_grads()
    with O.bf16_mode():
        l16, out16, g16 = oracle_grads()
    m.set_precision("bf16")
    m.train()
    m.zero_grad()
    sc = {k: v.cuda() for k, v in s.items()}
    torch.manual_seed(9)
    o = m(sc, train=True, padding=True, sparse=True)
    loss, _ = get_loss(o, sc, scale=o["scale"], loss=["log_l1_loss"], lam=[1.0], scale_regularization=0.01, tag="weak")
    (loss * 100.0).backward()
    assert abs(loss.item() - l16.item()) < 2 * TOL_LOSS * max(1.0, abs(l16.item()))
    assert rel(o["popcount"].detach().cpu(), out16["popcount"].detach()) < 2 * TOL_COUNT
    got = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}
    _bf16_grad_gate(got, g16, g32, "padding=True 90x86")
    m.zero_grad()


def test_bf16_train_step_with_the_unfused_backward(monkeypatch):
    """POPCORN_FUSED_CONV_BWD=0 (the A/B switch of engine.py) keeps the separate data-gradient / weight-gradient launches
    reachable: same gate as the fused step, and both forms agree with each other to bf16 rounding of the gradients."""
    from popcorn_amd import engine as E
    from popcorn_amd.train import FusedTrainStep
    s = _sample()
    m0, sd = _model()
    torch.manual_seed(3)
    _, _, g32, _ = O.train_step_grads(sd, dict(s))
    with O.bf16_mode():
        torch.manual_seed(3)
        l16, _, g16, _ = O.train_step_grads(sd, dict(s))
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(E, "FUSED_CONV_BWD", fused)
        m, _ = _model()
        m.set_precision("bf16")
        tr = FusedTrainStep(m, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01)
        torch.manual_seed(3)
        loss = tr.step({k: v.cuda() for k, v in s.items()})
        torch.cuda.synchronize()
        assert abs(loss[0].item() - l16.item()) < TOL_LOSS * abs(l16.item())
        _bf16_grad_gate({n: tr.grads[n] for n in g16}, g16, g32, f"fused={fused}")
        res[fused] = {n: tr.grads[n].clone() for n in g16}
    assert max(rel(res[True][n], res[False][n]) for n in g16) < TOL_GRAD_WORST


# ---- round 4: shared channels-last input of the first layers (pc_ingest_cl8, weight / gradient channel windows) -----------------
@pytest.mark.parametrize("shape,pads", [((3, 100, 100), (14, 14, 14, 14)), ((2, 37, 53), (5, 6, 2, 9)), ((1, 64, 48), (0, 0, 8, 8))])
@pytest.mark.parametrize("norm", [True, False])
def test_bf16_ingest_cl8_vs_torch(shape, pads, norm):
    """Band select + normalise + reflect pad + round, as one channels-last bf16 tensor with 8-channel slots: BIT-exact against
    torch (same fp32 division as pc_select_normalize, same round-to-nearest-even), zero in the unused channel slots."""
    import torch.nn.functional as F
    from popcorn_amd import ops
    from popcorn_amd.data import stats
    B, H, W = shape
    top, bottom, left, right = pads
    g = torch.Generator().manual_seed(7)
    raw = torch.cat([torch.randint(0, 10000, (B, 13, H, W), generator=g).float(), torch.randn(B, 2, H, W, generator=g) * 4 - 12], 1)
    order = [4, 5, 2, 1, 0, 3]                                  # model channels in stream order: [VV, VH | B, G, R, NIR]
    band = [stats.BAND6[c] for c in order]
    mean = [stats.MEAN6[c] for c in order] if norm else None
    std = [stats.STD6[c] for c in order] if norm else None
    out = ops.ingest_cl8(raw.cuda(), band, mean, std, top, bottom, left, right)
    assert out.dtype == torch.bfloat16 and tuple(out.shape) == (B, 8, H + top + bottom, W + left + right) and out.stride(1) == 1
    x = raw[:, band]
    if norm:
        x = (x - torch.tensor(mean).view(1, 6, 1, 1)) / torch.tensor(std).view(1, 6, 1, 1)
    ref = F.pad(x, (left, right, top, bottom), mode="reflect").to(torch.bfloat16)
    assert torch.equal(out[:, :6].cpu(), ref)
    assert not out[:, 6:].any()


@pytest.mark.parametrize("window", [(0, 2), (2, 4), (0, 8), (5, 3)])
@pytest.mark.parametrize("shape", [(2, 128, 128), (1, 37, 53)])
def test_bf16_conv_fwd_and_wgrad_with_a_channel_window_of_a_shared_input(window, shape):
    """pc_conv_fwd_desc.w_ci0 / w_cin and pc_wgrad_reduce_desc.src_cin / src_ci0: a conv whose weight covers a channel window of an
    8-channel channels-last input == the conv over the sliced input; its weight gradient == the window of the full gradient."""
    import torch.nn.functional as F
    from popcorn_amd import ops, _lib as L
    B, H, W = shape
    ci0, cin = window
    x8 = _bf(_mk(B, 8, H, W, seed=70))
    w = _mk(8, cin, 3, 3, seed=71, scale=0.3)
    b = _mk(8, seed=72, scale=0.1)
    gamma, beta, mean, var = _bn(8, 73)
    wd = _bf(w).double().requires_grad_(True)
    bias = b.double().clone().requires_grad_(True)
    y = F.conv2d(x8[:, ci0:ci0 + cin].double(), wd, bias, padding=1)
    ref = F.relu(F.batch_norm(y, mean.double(), var.double(), gamma.double(), beta.double(), training=False, eps=1e-5)).float()
    gout = _bf(_mk(B, 8, H, W, seed=74))
    y.backward(gout.double())
    with L.precision("bf16"):
        out = L.empty_act(B, 8, H, W, "cuda")
        bnd = L.bn(b.cuda(), gamma.cuda(), beta.cuda(), mean.cuda(), var.cuda(), 1e-5)
        ops.conv3x3_fwd_group([{"a": _dev(x8), "w": w.cuda(), "bn": bnd, "out": out, "w_window": (ci0, cin)}])
        _close_bf16(out, ref)
        dw = torch.full((8, cin, 3, 3), float("nan"), device="cuda")
        db = torch.full((8,), float("nan"), device="cuda")
        wb = ops.WgradBatch(torch.device("cuda"))
        wb.conv3x3_group([{"a": _dev(x8), "g": _dev(gout), "dw": dw, "db": db, "src_window": (ci0, cin)}], 8)
        wb.finish()
    # (the weight gradient contracts the UNROUNDED-weight-independent operands x and g: compare with autograd of the sliced conv)
    sw = wd.grad.abs().max().item()
    assert (dw.cpu().double() - wd.grad).abs().max().item() <= 2e-5 * sw
    assert (db.cpu().double() - bias.grad).abs().max().item() <= 2e-5 * max(bias.grad.abs().max().item(), 1.0)
    with L.precision("fp32"):
        with pytest.raises(RuntimeError):                      # the window is a bf16-mode feature: refused, not ignored, in fp32 mode
            ops.conv3x3_fwd_group([{"a": x8.cuda(), "w": w.cuda(), "bn": bnd, "out": torch.empty(B, 8, H, W, device="cuda"),
                                    "w_window": (ci0, cin)}])


def test_bf16_whole_level_forward_kernel_is_bit_identical_to_the_three_launches(monkeypatch):
    """level2_cl.hip (down2's DoubleConv + up2's ConvTranspose2d on the 32 x 32 level, one launch, whole tile in LDS) against the
    layer-by-layer launches: same MFMA instruction order per accumulator and the same rounding points, so c1, c2, u2, everything
    downstream and the frozen extractor's partial logits are BIT-identical; and the op refuses what it cannot take."""
    from popcorn_amd import engine as E, ops, _lib as L
    from popcorn_amd.model import POPCORN
    torch.manual_seed(1600)
    m = POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
    m.set_precision("bf16")
    eng_u, eng_b = m.engines()
    X = torch.randn(3, 6, 100, 100, generator=torch.Generator().manual_seed(2)).cuda()
    outs, launches = {}, {}
    orig = ops.level2_fwd_group
    with L.precision("bf16"):
        for flag in (True, False):
            monkeypatch.setattr(E, "FUSED_LEVEL2", flag)
            n = [0]

            def counted(problems, _n=n):
                _n[0] += 1
                return orig(problems)
            monkeypatch.setattr(ops, "level2_fwd_group", counted)
            (f_b, f_u), (_, saved) = E.forward_multi([eng_b, eng_u], X, 14, 14, 128, 128, [False, True], logit_only=[True, False])
            torch.cuda.synchronize()
            launches[flag] = n[0]
            outs[flag] = (f_b.clone(), f_u.clone(), {s: {k: saved[s][k].clone() for k in ("c1", "c2", "u2", "e1", "e2")}
                                                     for s in ("sar_stream", "optical_stream")})
        assert launches == {True: 1, False: 0}
        assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
        for s in ("sar_stream", "optical_stream"):
            for k in ("c1", "c2", "u2", "e1", "e2"):
                a, b = outs[True][2][s][k], outs[False][2][s][k]
                assert a.dtype == torch.bfloat16 and a.stride(1) == 1 and torch.equal(a, b), (s, k)
        assert not ops.level2_fwd_ok(torch.zeros(2, 16, 32, 32, device="cuda"), None)                         # fp32 container
        assert not ops.level2_fwd_ok(L.empty_act(2, 16, 30, 32, "cuda"), None)
        assert ops.level2_fwd_ok(L.empty_act(2, 16, 32, 32, "cuda"), L.empty_act(2, 16, 64, 64, "cuda"))


@pytest.mark.parametrize("B,nprob", [(3, 2), (1, 1), (5, 4)])
def test_bf16_whole_level_backward_kernel_against_the_two_launches(B, nprob):
    """level2_bwd_cl_kernel (both conv layers of the 32 x 32 level: data-gradient chain, pooling scatter, both weight / bias gradients
    in one launch, the intermediate gradient never leaves LDS) against conv3x3_bwd_cl_kernel<16, 16> + <16, 16, pool> on the same
    operands: the accumulated full-resolution gradient is BIT-identical (same MFMA order per accumulator, same rounding point of the
    intermediate); the weight gradients are sums over other pixel groups and agree to fp32 summation noise."""
    from popcorn_amd import ops, _lib as L
    import torch.nn.functional as F
    probs_f, probs_l, keep = [], [], []
    with L.precision("bf16"):
        for i in range(nprob):
            act = _bf(F.relu(_mk(B, 16, 64, 64, seed=300 + 10 * i)))
            x = F.max_pool2d(act, 2)
            c1 = _bf(F.relu(_mk(B, 16, 32, 32, seed=301 + 10 * i)))
            g2 = _bf(_mk(B, 16, 32, 32, seed=302 + 10 * i))
            w1 = _mk(16, 16, 3, 3, seed=303 + 10 * i, scale=0.2).cuda()
            w2 = _mk(16, 16, 3, 3, seed=304 + 10 * i, scale=0.2).cuda()
            bn1 = [t.cuda() for t in _bn(16, 305 + 10 * i)]
            bna = [t.cuda() for t in _bn(16, 306 + 10 * i)]
            prev = _bf(_mk(B, 16, 64, 64, seed=307 + 10 * i))
            keep.append((bn1, bna))
            d = {"g2": _dev(g2), "c1": _dev(c1), "x": _dev(x), "act": _dev(act), "w1": w1, "w2": w2,
                 "bn1": L.bn(None, *bn1), "act_bn": L.bn(None, *bna)}
            mk = lambda: {k: torch.full(sh, float("nan"), device="cuda") for k, sh in
                          (("dw1", (16, 16, 3, 3)), ("db1", (16,)), ("dw2", (16, 16, 3, 3)), ("db2", (16,)))}
            probs_f.append(dict(d, out=_dev(prev), **mk()))
            probs_l.append(dict(d, out=_dev(prev), g1=L.empty_act(B, 16, 32, 32, "cuda"), **mk()))
            assert ops.level2_bwd_ok(d["g2"], d["c1"], d["x"], d["act"], probs_f[-1]["out"])
        wb = ops.WgradBatch(torch.device("cuda"))
        wb.level2_bwd_group(probs_f)
        wb.finish()
        wb = ops.WgradBatch(torch.device("cuda"))
        wb.conv3x3_bwd_group([{"g": p["g2"], "x": p["c1"], "w": p["w2"], "out": p["g1"], "dw": p["dw2"], "db": p["db2"], "x_bn": p["bn1"]}
                              for p in probs_l], 16, 0)
        wb.conv3x3_bwd_group([{"g": p["g1"], "x": p["x"], "w": p["w1"], "out": p["out"], "dw": p["dw1"], "db": p["db1"],
                               "pool_act": p["act"], "x_bn": p["act_bn"]} for p in probs_l], 16, 0)
        wb.finish()
        torch.cuda.synchronize()
        assert not ops.level2_bwd_ok(torch.zeros(2, 16, 32, 32, device="cuda"), probs_f[0]["c1"], probs_f[0]["x"], probs_f[0]["act"],
                                     probs_f[0]["out"])                                                          # fp32 container
    for pf, pl in zip(probs_f, probs_l):
        assert torch.equal(pf["out"], pl["out"])
        for k in ("dw1", "db1", "dw2", "db2"):
            a, b = pf[k].double(), pl[k].double()
            assert torch.isfinite(a).all() and (a - b).abs().max().item() <= 2e-6 * b.abs().max().item(), k


def test_bf16_train_step_with_the_whole_level_backward_kernel_equals_the_layerwise_step(monkeypatch):
    """A bf16 train step with level2_bwd_cl_kernel in it against the same step with the two layer launches (POPCORN_FUSED_LEVEL2_BWD):
    the data gradients are bit-identical, so every gradient outside the level's two conv layers is too; the level's own four weight /
    bias gradients are sums over other pixel groups (fp32 summation noise) -- and the kernel really ran."""
    from popcorn_amd import engine as E, ops
    from popcorn_amd.data import stats
    from popcorn_amd.data.synthetic import make_raw_batch
    from popcorn_amd.model import POPCORN
    from popcorn_amd.train import FusedTrainStep
    batch = make_raw_batch(3, 100, 100, seed=19, device="cuda", region="disc")
    x = ops.select_normalize(batch["raw"], stats.BAND6, stats.MEAN6, stats.STD6)
    grads, calls = {}, {}
    orig = ops.WgradBatch.level2_bwd_group
    for flag in (True, False):
        monkeypatch.setattr(E, "FUSED_LEVEL2_BWD", flag)
        n = [0]

        def counted(self, problems, _n=n):
            _n[0] += 1
            return orig(self, problems)
        monkeypatch.setattr(ops.WgradBatch, "level2_bwd_group", counted)
        torch.manual_seed(1600)
        m = POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
        m.set_precision("bf16")
        tr = FusedTrainStep(m, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01)
        torch.manual_seed(4)
        tr.step({"input": x, "admin_mask": batch["admin_mask"], "census_idx": batch["census_idx"], "y": batch["y"]})
        torch.cuda.synchronize()
        grads[flag] = {k: v.clone() for k, v in tr.grads.items()}
        calls[flag] = n[0]
    assert calls == {True: 1, False: 0}
    level = [k for k in grads[True] if ".down2." in k and ".conv." in k]
    assert len(level) >= 8                                           # two streams x two layers x (weight, bias)
    for k in grads[True]:
        a, b = grads[True][k], grads[False][k]
        if k in level and (k.endswith("conv.0.weight") or k.endswith("conv.0.bias") or k.endswith("conv.3.weight") or k.endswith("conv.3.bias")):
            assert (a - b).abs().max().item() <= 2e-6 * max(b.abs().max().item(), 1e-12), k
        else:
            assert torch.equal(a, b), k


@pytest.mark.parametrize("use_graph", [False, True])
def test_bf16_fused_step_from_raw_tiles_equals_step_from_normalised_input(use_graph):
    """bf16 mode: the step fed the RAW 15-band tile (first launch = pc_ingest_cl8: select + normalise + pad + round into the shared
    channels-last input) against the step fed the normalised input (the same ingest without the normalisation): identical
    values after rounding, so losses and parameters agree bit for bit."""
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
        m.set_precision("bf16")
        tr = FusedTrainStep(m, lr=1e-3, weight_decay=1e-5, gradient_clip=0.01, use_graph=use_graph)
        losses = []
        for step in range(3):
            torch.manual_seed(50 + step)
            losses.append(tr.step({key: data, "admin_mask": batch["admin_mask"], "census_idx": batch["census_idx"], "y": batch["y"]}).tolist())
        torch.cuda.synchronize()
        runs.append((losses, tr.flat_p.clone()))
    assert runs[0][0] == runs[1][0]
    assert torch.equal(runs[0][1], runs[1][1])


@pytest.mark.parametrize("ic", [2, 4])
def test_bf16_single_modality_train_step_vs_bf16_oracle(ic):
    """S1-only / S2-only variants (popcorn.py:48-54,136-145) in bf16 mode: one stream runs, the other 8 feature channels are a ZERO
    channels-last tensor (round 4: that allocation used torch.zeros with a memory_format argument and raised -- the combination had never
    been run).  Loss, popcount and the 32 gradients against the bf16 oracle, same gates as the dual-stream model."""
    from popcorn_amd.model import POPCORN
    from popcorn_amd.train import FusedTrainStep
    g = np.load(os.path.join(G, "g8_single_modality.npz"))
    torch.manual_seed(1600)
    m = POPCORN(input_channels=ic, feature_extractor="DDA", occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    s = {k: torch.from_numpy(g[f"ic{ic}/{k}"]) for k in ("input", "admin_mask", "census_idx", "y")}
    torch.manual_seed(5)
    _, _, g32, _ = O.train_step_grads(sd, dict(s))
    with O.bf16_mode():
        torch.manual_seed(5)
        l16, out16, g16, _ = O.train_step_grads(sd, dict(s))
    m.set_precision("bf16")
    tr = FusedTrainStep(m, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01)
    torch.manual_seed(5)
    loss = tr.step({k: v.cuda() for k, v in s.items()})
    torch.cuda.synchronize()
    assert abs(loss[0].item() - l16.item()) < TOL_LOSS * abs(l16.item())
    assert rel(tr.last["popcount"].cpu(), out16["popcount"]) < TOL_COUNT
    assert len(g16) == 32
    _bf16_grad_gate({n: tr.grads[n] for n in g16}, g16, g32, f"ic{ic}")


@pytest.mark.parametrize("hw", [(100, 100), (72, 40), (37, 53)])
def test_bf16_transposed_conv_in_the_conv_epilogue_is_bit_identical(monkeypatch, hw):
    """EPI_UPT: up1's ConvTranspose2d runs in the epilogue of the conv that produces its input (the lane's rounded output is the MFMA
    operand): u1, e2, f1, the features and the frozen extractor's logits are BIT-identical to the two-launch form, at exact-2x and ragged
    geometries, and the 8-channel transposed-conv launch is gone."""
    from popcorn_amd import engine as E, ops, _lib as L
    from popcorn_amd.model import POPCORN
    from popcorn_amd.model.popcorn import pad_geometry
    torch.manual_seed(1600)
    m = POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
    m.set_precision("bf16")
    eng_u, eng_b = m.engines()
    H, W = hw
    X = torch.randn(2, 6, H, W, generator=torch.Generator().manual_seed(3)).cuda()
    pt, pb, pl, pr = pad_geometry(H, W, True)
    Hp, Wp = H + pt + pb, W + pl + pr
    outs, n8 = {}, {}
    orig = ops.convt2x2_group
    with L.precision("bf16"):
        for flag in (True, False):
            monkeypatch.setattr(E, "FUSED_UPT", flag)
            cnt = [0]

            def counted(problems, _c=cnt):
                _c[0] += sum(1 for pr_ in problems if pr_["x"].shape[1] == 8)
                return orig(problems)
            monkeypatch.setattr(ops, "convt2x2_group", counted)
            (f_b, f_u), (_, saved) = E.forward_multi([eng_b, eng_u], X, pt, pl, Hp, Wp, [False, True], logit_only=[True, False])
            torch.cuda.synchronize()
            n8[flag] = cnt[0]
            outs[flag] = (f_b.clone(), f_u.clone(), {s: {k: saved[s][k].clone() for k in ("u1", "e2", "f1")} for s in ("sar_stream", "optical_stream")})
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
    for s in ("sar_stream", "optical_stream"):
        for k in ("u1", "e2", "f1"):
            assert torch.equal(outs[True][2][s][k], outs[False][2][s][k]), (s, k)
    exact2x = (Hp // 2) * 2 == Hp and (Wp // 2) * 2 == Wp
    assert n8[False] == 4 and n8[True] == (0 if exact2x else 4), (n8, exact2x)


def test_bf16_training_step_with_and_without_the_fused_transposed_conv_epilogue_is_bit_identical(monkeypatch):
    """``FUSED_UPT`` (up1's transposed conv in the epilogue of up2's second conv) in TRAINING: loss, every gradient and the parameters after
    the step are bit-identical to the separate launch (the switch was only covered in forward passes)."""
    from popcorn_amd import engine as E, ops
    from popcorn_amd.data import stats
    from popcorn_amd.data.synthetic import make_raw_batch
    from popcorn_amd.model import POPCORN
    from popcorn_amd.train import FusedTrainStep
    batch = make_raw_batch(3, 100, 100, seed=19, device="cuda", region="disc")
    x = ops.select_normalize(batch["raw"], stats.BAND6, stats.MEAN6, stats.STD6)
    res = {}
    for flag in (True, False):
        monkeypatch.setattr(E, "FUSED_UPT", flag)
        torch.manual_seed(1600)
        m = POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
        m.set_precision("bf16")
        tr = FusedTrainStep(m, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01)
        torch.manual_seed(4)
        loss = tr.step({"input": x, "admin_mask": batch["admin_mask"], "census_idx": batch["census_idx"], "y": batch["y"]})
        torch.cuda.synchronize()
        res[flag] = (loss.clone(), tr.flat_g.clone(), tr.flat_p.clone())
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)
