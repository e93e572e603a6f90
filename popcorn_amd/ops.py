"""Op-level Python wrappers over the C ABI (one call = one kernel launch on the current stream).

Used by the kernel parity tests and by the nn.Module glue; the training hot path goes through the fused
composite entry points instead (popcorn_amd/engine.py).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


def conv3x3_raw(a, w, bnd, relu=True, b=None, a_mode=L.PC_SRC_DIRECT, a_pad=(0, 0), chmap=(0, 1, 2, 3), out=None,
                out_hw=None, b_offset=(0, 0), a_channels=None):
    """Conv2d(3x3, pad 1)(cat[a, b]) -> BN(eval) -> ReLU with a prebuilt pc_bn descriptor.  networks.py:259-266,318.

    a_mode POOL2: a is at twice the resolution (MaxPool2d(2) fused).  a_mode REFLECT: a is reflect-padded by
    a_pad=(top,left) up to out_hw and its channels gathered through chmap."""
    L.require_device(a, w)
    B = a.shape[0]
    Ca = a.shape[1] if a_channels is None else a_channels
    Cb = 0 if b is None else b.shape[1]
    Cout = w.shape[0]
    if a_mode == L.PC_SRC_POOL2:
        H, W = a.shape[2] // 2, a.shape[3] // 2
    elif a_mode == L.PC_SRC_REFLECT:
        H, W = out_hw
    else:
        H, W = a.shape[2], a.shape[3]
    sa = L.src(a, C_=Ca, mode=a_mode, oy=a_pad[0], ox=a_pad[1], chmap=chmap)
    sb = L.src(b, oy=b_offset[0], ox=b_offset[1]) if b is not None else None
    if out is None:
        out = L.empty_act(B, Cout, H, W, a.device)
    d = L.dst(out)
    code = L.lib().pc_conv3x3_bn_relu_fwd(C.byref(sa), C.byref(sb) if sb is not None else None, L.ptr(w), C.byref(bnd),
                                          int(relu), C.byref(d), B, H, W, Ca + Cb, Cout, L.stream_ptr())
    L.check(code, "pc_conv3x3_bn_relu_fwd")
    return out


def conv3x3_fwd_group(problems, relu=True, a_mode=L.PC_SRC_DIRECT, a_pad=(0, 0), out_hw=None, a_channels=None,
                      b_offset=(0, 0)):
    """Grouped form of conv3x3_raw: ``problems`` = list (<= 4) of dicts {a, w, bn, out, b (optional), chmap (optional),
    pool_out (optional: MaxPool2d(2) of out, see pool_out_like), dot_w + dot_out (optional, 8 -> 8 layers: write
    sum_co dot_w[co] * out[co] as a one-channel map instead of out)}
    with identical geometry; one launch (blockIdx.y = problem)."""
    n = len(problems)
    assert 1 <= n <= L.PC_MAX_GROUP
    a0, w0 = problems[0]["a"], problems[0]["w"]
    L.require_device(a0, w0)
    B = a0.shape[0]
    Ca = a0.shape[1] if a_channels is None else a_channels
    Cb = 0 if problems[0].get("b") is None else problems[0]["b"].shape[1]
    Cout = w0.shape[0]
    if a_mode == L.PC_SRC_POOL2:
        H, W = a0.shape[2] // 2, a0.shape[3] // 2
    elif a_mode == L.PC_SRC_REFLECT:
        H, W = out_hw
    else:
        H, W = a0.shape[2], a0.shape[3]
    keep = []
    descs = (L.PcConvFwdDesc * n)()
    for i, pr in enumerate(problems):
        sa = L.src(pr["a"], C_=Ca, mode=a_mode, oy=a_pad[0], ox=a_pad[1], chmap=pr.get("chmap", (0, 1, 2, 3)))
        sb = L.src(pr["b"], oy=b_offset[0], ox=b_offset[1]) if pr.get("b") is not None else None
        d = L.dst(pr["out"]) if pr.get("out") is not None else None
        keep += [sa, sb, d]
        descs[i].a = C.pointer(sa)
        descs[i].b = C.pointer(sb) if sb is not None else None
        descs[i].w = pr["w"].data_ptr()
        descs[i].bn = C.pointer(pr["bn"])
        descs[i].out = C.pointer(d) if d is not None else None
        if pr.get("dot_w") is not None:          # 1x1-conv partial sum over the 8 output channels instead of the map
            dd = L.dst(pr["dot_out"])
            keep += [dd, pr["dot_w"]]
            descs[i].dot_w = pr["dot_w"].data_ptr()
            descs[i].dot_out = C.pointer(dd)
        if pr.get("pool_out") is not None:
            dp = L.dst(pr["pool_out"])
            keep.append(dp)
            descs[i].pool_out = C.pointer(dp)
        if pr.get("upt_out") is not None:        # bf16 mode, 8 -> 8: the transposed conv that follows, in this launch's epilogue
            du = L.dst(pr["upt_out"])
            keep.append(du)
            descs[i].upt_w, descs[i].upt_out = pr["upt_w"].data_ptr(), C.pointer(du)
            descs[i].upt_b = pr["upt_b"].data_ptr() if pr.get("upt_b") is not None else None
        if pr.get("w_window") is not None:       # bf16 mode: w covers input channels [ci0, ci0 + cin) of the shared 8-channel input
            descs[i].w_ci0, descs[i].w_cin = int(pr["w_window"][0]), int(pr["w_window"][1])
            assert pr["w"].shape[1] == descs[i].w_cin and Ca + Cb == 8
    L.check(L.lib().pc_conv3x3_bn_relu_fwd_group(n, descs, int(relu), B, H, W, Ca + Cb, Cout, L.stream_ptr()),
            "pc_conv3x3_bn_relu_fwd_group")


def pool_out_like(out):
    """Buffer for the 2x2-max-pooled second output of conv3x3_fwd_group (problem key "pool_out"), or None when the
    geometry of ``out`` (B, C, H, W) does not qualify (pc_conv3x3_pool_out_ok)."""
    B, Cc, H, W = out.shape
    d = L.dst(out)
    if not L.lib().pc_conv3x3_pool_out_ok(C.byref(d), H, W):
        return None
    return L.empty_act(B, Cc, H // 2, W // 2, out.device)


def conv3x3_dgrad_group(problems, c0, cn, pool=False, accumulate=False):
    """Grouped conv3x3_dgrad: problems = list of dicts {g, w, out, act (optional), act_bn (optional), c0_add (optional: this
    problem's input-channel block starts at c0 + c0_add -- both column blocks of a concat layer in one launch)}."""
    n = len(problems)
    assert 1 <= n <= L.PC_MAX_GROUP
    g0, w0 = problems[0]["g"], problems[0]["w"]
    L.require_device(g0, w0)
    B, Cg, H, W = g0.shape
    keep = []
    descs = (L.PcConvDgradDesc * n)()
    for i, pr in enumerate(problems):
        sg, d = L.src(pr["g"]), L.dst(pr["out"])
        sa = L.src(pr["act"]) if pr.get("act") is not None else None
        keep += [sg, d, sa]
        descs[i].g = C.pointer(sg)
        ca = int(pr.get("c0_add", 0))
        assert 0 <= c0 + ca and c0 + ca + cn <= pr["w"].shape[1]
        descs[i].w = pr["w"].data_ptr() + 4 * 9 * ca
        descs[i].act = C.pointer(sa) if sa is not None else None
        descs[i].act_bn = C.pointer(pr["act_bn"]) if pr.get("act_bn") is not None else None
        descs[i].out = C.pointer(d)
    L.check(L.lib().pc_conv3x3_dgrad_group(n, descs, w0.shape[1], c0, cn, int(pool), int(accumulate), B, H, W, Cg,
                                           L.stream_ptr()), "pc_conv3x3_dgrad_group")


def conv3x3_bwd_ok(g, x, out, pool_act=None):
    """fp32 mode: does the one-launch backward of a conv layer (WgradBatch.conv3x3_bwd_group) take this gradient / 8-channel input
    block / output (and, for a Down block's first layer, the full-resolution activation the input was pooled from)?"""
    if g.dtype != torch.float32:
        return False
    B, _, H, W = g.shape
    sg, sx, d = L.src(g), L.src(x), L.dst(out)
    spa = L.src(pool_act) if pool_act is not None else None
    return bool(L.lib().pc_conv3x3_bwd_ok(C.byref(sg), C.byref(sx), C.byref(d), C.byref(spa) if spa is not None else None, B, H, W))


def conv3x3_bn_relu(a, w, bias, gamma=None, beta=None, mean=None, var=None, eps=1e-5, relu=True, **kw):
    return conv3x3_raw(a, w, L.bn(bias, gamma, beta, mean, var, eps), relu=relu, **kw)


def conv3x3_dgrad(g, w, c0, cn, out, act=None, act_bn=None, pool=False, accumulate=False):
    """Data gradient of the conv above for forward input channels [c0, c0+cn) -> out (see popcorn_hip.h)."""
    L.require_device(g, w, out)
    B, Cg, H, W = g.shape
    sg = L.src(g)
    d = L.dst(out)
    sa = L.src(act) if act is not None else None
    code = L.lib().pc_conv3x3_dgrad(C.byref(sg), L.ptr(w), w.shape[1], c0, cn,
                                    C.byref(sa) if sa is not None else None,
                                    C.byref(act_bn) if act_bn is not None else None,
                                    int(pool), int(accumulate), C.byref(d), B, H, W, Cg, L.stream_ptr())
    L.check(code, "pc_conv3x3_dgrad")
    return out


_ws_cache = {}
_ws_retired = []       # outgrown workspaces stay allocated: a captured HIP graph may hold their device pointers


def _workspace(nbytes: int, device) -> torch.Tensor:
    key = (str(device), "ws")
    t = _ws_cache.get(key)
    if t is None or t.numel() < nbytes:
        if t is not None:
            _ws_retired.append(t)
        t = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _ws_cache[key] = t
    return t


def conv3x3_wgrad(a, g, cout, dw=None, db=None, b=None, accumulate=False, a_mode=L.PC_SRC_DIRECT, a_pad=(0, 0),
                  chmap=(0, 1, 2, 3), b_offset=(0, 0), a_channels=None):
    """Weight/bias gradient of conv3x3 over x = cat[a, b] with output gradient g (already ReLU/BN-masked)."""
    L.require_device(a, g)
    B, Cg, H, W = g.shape
    Ca = a.shape[1] if a_channels is None else a_channels
    Cb = 0 if b is None else b.shape[1]
    cin = Ca + Cb
    sa = L.src(a, C_=Ca, mode=a_mode, oy=a_pad[0], ox=a_pad[1], chmap=chmap)
    sb = L.src(b, oy=b_offset[0], ox=b_offset[1]) if b is not None else None
    sg = L.src(g)
    if dw is None:
        dw = torch.empty(cout, cin, 3, 3, device=g.device, dtype=torch.float32)
    if db is None:
        db = torch.empty(cout, device=g.device, dtype=torch.float32)
    ws = _workspace(L.lib().pc_conv3x3_wgrad_ws_bytes(cin, cout), g.device)
    code = L.lib().pc_conv3x3_wgrad(C.byref(sa), C.byref(sb) if sb is not None else None, C.byref(sg), L.ptr(dw),
                                    L.ptr(db), int(accumulate), L.ptr(ws), B, H, W, cin, cout, L.stream_ptr())
    L.check(code, "pc_conv3x3_wgrad")
    return dw, db


def convt2x2(x, w, bias, out=None):
    """ConvTranspose2d(C, C, 2, stride=2).  networks.py:302,306."""
    L.require_device(x, w)
    B, Cc, H, W = x.shape
    if out is None:
        out = L.empty_act(B, Cc, 2 * H, 2 * W, x.device)
    sx, d = L.src(x), L.dst(out)
    L.check(L.lib().pc_convt2x2_fwd(C.byref(sx), L.ptr(w), L.ptr(bias), C.byref(d), B, H, W, Cc, L.stream_ptr()),
            "pc_convt2x2_fwd")
    return out


def convt2x2_group(problems):
    """Grouped ConvTranspose2d: problems = list of {x, w, bias, out}; one launch."""
    n = len(problems)
    x0 = problems[0]["x"]
    B, Cc, H, W = x0.shape
    keep, descs = [], (L.PcConvtFwdDesc * n)()
    for i, pr in enumerate(problems):
        sx, d = L.src(pr["x"]), L.dst(pr["out"])
        keep += [sx, d]
        descs[i].x, descs[i].w, descs[i].bias, descs[i].out = C.pointer(sx), pr["w"].data_ptr(), pr["bias"].data_ptr(), C.pointer(d)
    L.check(L.lib().pc_convt2x2_fwd_group(n, descs, B, H, W, Cc, L.stream_ptr()), "pc_convt2x2_fwd_group")


def level2_fwd_ok(x, u2):
    """Does the one-launch 32 x 32 level (pc_level2_fwd_group) take these tensors?  x: (B,16,32,32) pooled map, u2: (B,16,64,64)."""
    adt = L.act_dtype()                    # fp32 mode: planar fp32; bf16 mode: channels-last bf16 (level2_cl.hip)
    planar = adt == torch.float32
    if x is None or x.dim() != 4 or tuple(x.shape[1:]) != (16, 32, 32) or x.dtype != adt or x.stride(3 if planar else 1) != 1:
        return False
    if u2 is not None and (tuple(u2.shape[1:]) != (16, 64, 64) or u2.dtype != adt or u2.stride(3 if planar else 1) != 1):
        return False
    sx = L.src(x)
    du = L.dst(u2) if u2 is not None else None
    return bool(L.lib().pc_level2_fwd_ok(C.byref(sx), C.byref(du) if du is not None else None))


def conv3x3_up_fwd_ok(skip, z, out):
    """Does pc_conv3x3_up_fwd_group (first conv of an Up block computed from the LOW-resolution map, no up-sampled tensor) take
    these tensors?  skip (B,Cs,H,W), z (B,C,H/2,W/2), out (B,8,H,W)."""
    if skip is None or z is None or skip.dtype != torch.float32 or z.dtype != torch.float32 or out.dtype != torch.float32:
        return False
    if skip.stride(3) != 1 or z.stride(3) != 1 or out.stride(3) != 1:
        return False
    B, Cs, H, W = skip.shape
    ss, sz, do = L.src(skip), L.src(z), L.dst(out)
    return bool(L.lib().pc_conv3x3_up_fwd_ok(C.byref(ss), C.byref(sz), C.byref(do), H, W, Cs, z.shape[1]))


def conv3x3_up_compose(problems):
    """Composed operand images for a list (<= 8) of Up-block convolutions {w, wt, bt} (any mix of the 8 + 8 and 16 + 16 channel
    shapes) in ONE launch.  Returns one workspace tensor per problem (pass it as ``ws`` to conv3x3_up_fwd_group)."""
    n = len(problems)
    assert 1 <= n <= 2 * L.PC_MAX_GROUP
    dev = problems[0]["w"].device
    descs = (L.PcConvUpFwdDesc * n)()
    cs, cz, slots = (C.c_int * n)(), (C.c_int * n)(), []
    for i, pr in enumerate(problems):
        Cz = pr["wt"].shape[0]
        cs[i], cz[i] = pr["w"].shape[1] - Cz, Cz
        slots.append(torch.empty(int(L.lib().pc_conv3x3_up_ws_bytes(Cz)), dtype=torch.uint8, device=dev))
        descs[i].w, descs[i].wt = pr["w"].data_ptr(), pr["wt"].data_ptr()
        descs[i].bt = pr["bt"].data_ptr() if pr.get("bt") is not None else None
        descs[i].ws = slots[i].data_ptr()
    L.check(L.lib().pc_conv3x3_up_compose_group(n, descs, cs, cz, L.stream_ptr()), "pc_conv3x3_up_compose_group")
    return slots


def conv3x3_up_fwd_group(problems, relu=True):
    """conv3x3(cat[skip, ConvTranspose2d(z)]) + BN + ReLU without materialising the up-sampled map (networks.py:302-318):
    problems = list (<= 4) of dicts {skip, z, w ([8][Cs + C][3][3]), wt ([C][C][2][2]), bt ([C]), bn, out, ws (optional: the
    workspace conv3x3_up_compose filled for this problem)}.  Returns the workspaces (the backward pass reads them again)."""
    n = len(problems)
    assert 1 <= n <= L.PC_MAX_GROUP
    skip0, z0 = problems[0]["skip"], problems[0]["z"]
    L.require_device(skip0, z0)
    B, Cs, H, W = skip0.shape
    Cz = z0.shape[1]
    nbytes = int(L.lib().pc_conv3x3_up_ws_bytes(Cz))
    pre = all(pr.get("ws") is not None for pr in problems)
    keep, slots = [], []
    descs = (L.PcConvUpFwdDesc * n)()
    for i, pr in enumerate(problems):
        # composed operand images of this call (forward stages, bias table, the backward's data-gradient image): a fresh tensor
        # per problem -- the backward pass of a saved network reads it again (under graph capture it lives in the graph's pool)
        slots.append(pr["ws"] if pre else torch.empty(nbytes, dtype=torch.uint8, device=skip0.device))
        ss, sz, do = L.src(pr["skip"]), L.src(pr["z"]), L.dst(pr["out"])
        keep += [ss, sz, do]
        descs[i].skip, descs[i].z, descs[i].out = C.pointer(ss), C.pointer(sz), C.pointer(do)
        descs[i].w, descs[i].wt = pr["w"].data_ptr(), pr["wt"].data_ptr()
        descs[i].bt = pr["bt"].data_ptr() if pr.get("bt") is not None else None
        descs[i].bn = C.pointer(pr["bn"])
        descs[i].ws = slots[i].data_ptr()
    L.check(L.lib().pc_conv3x3_up_fwd_group(n, descs, int(relu) | (2 if pre else 0), B, H, W, Cs, Cz, L.stream_ptr()),
            "pc_conv3x3_up_fwd_group")
    return slots


_up_bwd_ws = {}


def conv3x3_up_bwd_ok(g, z, gz):
    if g is None or z is None or g.dtype != torch.float32 or z.dtype != torch.float32 or g.stride(3) != 1 or z.stride(3) != 1:
        return False
    if gz is not None and (gz.dtype != torch.float32 or gz.stride(3) != 1):
        return False
    B, Cg, H, W = g.shape
    sg, sz = L.src(g), L.src(z)
    dz = L.dst(gz) if gz is not None else None
    return bool(L.lib().pc_conv3x3_up_bwd_ok(C.byref(sg), C.byref(sz), C.byref(dz) if dz is not None else None, H, W, z.shape[1], z.shape[1]))


def conv3x3_up_bwd_group(problems, accumulate=False):
    """Backward of the up-sampled half of an Up block's first conv from the low-resolution map (pc_conv3x3_up_bwd_group):
    problems = list of {g (B,8,H,W), z (B,C,H/2,W/2), z_bn, gz (out, optional), w, wt, bt, fwd_ws (slot returned by
    conv3x3_up_fwd_group), dw (full conv weight gradient [8][Cs + C][3][3]), dwt, dbt}."""
    n = len(problems)
    g0, z0 = problems[0]["g"], problems[0]["z"]
    L.require_device(g0, z0)
    B, Cg, H, W = g0.shape
    Cz = z0.shape[1]
    nbytes = int(L.lib().pc_conv3x3_up_bwd_ws_bytes(B, H, Cz))
    key = (str(g0.device), Cz, nbytes)
    slots = _up_bwd_ws.setdefault(key, [])
    keep = []
    descs = (L.PcConvUpBwdDesc * n)()
    for i, pr in enumerate(problems):
        if len(slots) <= i:
            slots.append(torch.empty(nbytes, dtype=torch.uint8, device=g0.device))
        sg, sz = L.src(pr["g"]), L.src(pr["z"])
        dz = L.dst(pr["gz"]) if pr.get("gz") is not None else None
        keep += [sg, sz, dz]
        descs[i].g, descs[i].z = C.pointer(sg), C.pointer(sz)
        descs[i].z_bn = C.pointer(pr["z_bn"])
        descs[i].gz = C.pointer(dz) if dz is not None else None
        descs[i].w, descs[i].wt = pr["w"].data_ptr(), pr["wt"].data_ptr()
        descs[i].bt = pr["bt"].data_ptr() if pr.get("bt") is not None else None
        descs[i].fwd_ws = pr["fwd_ws"].data_ptr()
        descs[i].ws = slots[i].data_ptr()
        descs[i].dw, descs[i].dwt = pr["dw"].data_ptr(), pr["dwt"].data_ptr()
        descs[i].dbt = pr["dbt"].data_ptr() if pr.get("dbt") is not None else None
    L.check(L.lib().pc_conv3x3_up_bwd_group(n, descs, int(accumulate), B, H, W, Cz, Cz, L.stream_ptr()), "pc_conv3x3_up_bwd_group")


def level2_bwd_ok(g2, c1, x, act, out):
    """Does the one-launch backward of the 32 x 32 level (pc_level2_bwd_group) take these tensors?"""
    adt = L.act_dtype()                    # fp32 mode: planar fp32 (level2.hip); bf16 mode: channels-last bf16 (level2_cl.hip)
    unit = 3 if adt == torch.float32 else 1
    for t, shp in ((g2, (16, 32, 32)), (c1, (16, 32, 32)), (x, (16, 32, 32)), (act, (16, 64, 64)), (out, (16, 64, 64))):
        if t is None or t.dim() != 4 or tuple(t.shape[1:]) != shp or t.dtype != adt or t.stride(unit) != 1:
            return False
    sg, sc, sx, sa, do = L.src(g2), L.src(c1), L.src(x), L.src(act), L.dst(out)
    return bool(L.lib().pc_level2_bwd_ok(C.byref(sg), C.byref(sc), C.byref(sx), C.byref(sa), C.byref(do)))


def level2_fwd_group(problems):
    """DoubleConv(16,16) + ConvTranspose2d(16,16,2,2) of the 32 x 32 level in ONE launch (networks.py:253-271,302): problems =
    list (<= 4) of dicts {x (pooled input), w1, bn1, w2, bn2, wt, bt, u2 (out), c1 / c2 (optional outs: saved activations)}."""
    n = len(problems)
    assert 1 <= n <= L.PC_MAX_GROUP
    L.require_device(problems[0]["x"])
    B = problems[0]["x"].shape[0]
    keep = []
    descs = (L.PcLevel2FwdDesc * n)()
    for i, pr in enumerate(problems):
        sx = L.src(pr["x"])
        du = L.dst(pr["u2"]) if pr.get("u2") is not None else None
        d1 = L.dst(pr["c1"]) if pr.get("c1") is not None else None
        d2 = L.dst(pr["c2"]) if pr.get("c2") is not None else None
        keep += [sx, du, d1, d2]
        descs[i].x = C.pointer(sx)
        descs[i].w1, descs[i].w2, descs[i].wt = pr["w1"].data_ptr(), pr["w2"].data_ptr(), pr["wt"].data_ptr()
        descs[i].bt = pr["bt"].data_ptr() if pr.get("bt") is not None else None
        descs[i].bn1, descs[i].bn2 = C.pointer(pr["bn1"]), C.pointer(pr["bn2"])
        descs[i].c1 = C.pointer(d1) if d1 is not None else None
        descs[i].c2 = C.pointer(d2) if d2 is not None else None
        descs[i].u2 = C.pointer(du) if du is not None else None
    L.check(L.lib().pc_level2_fwd_group(n, descs, B, L.stream_ptr()), "pc_level2_fwd_group")


def convt2x2_dgrad_group(problems):
    """Grouped convT data gradient: problems = list of {g, w, out, act, act_bn}."""
    n = len(problems)
    B, Cc, H, W = problems[0]["out"].shape
    keep, descs = [], (L.PcConvtDgradDesc * n)()
    for i, pr in enumerate(problems):
        sg, d = L.src(pr["g"]), L.dst(pr["out"])
        sa = L.src(pr["act"]) if pr.get("act") is not None else None
        keep += [sg, d, sa]
        descs[i].g, descs[i].w, descs[i].out = C.pointer(sg), pr["w"].data_ptr(), C.pointer(d)
        descs[i].act = C.pointer(sa) if sa is not None else None
        descs[i].act_bn = C.pointer(pr["act_bn"]) if pr.get("act_bn") is not None else None
    L.check(L.lib().pc_convt2x2_dgrad_group(n, descs, B, H, W, Cc, L.stream_ptr()), "pc_convt2x2_dgrad_group")


def convt2x2_dgrad(g, w, out, act=None, act_bn=None):
    L.require_device(g, w, out)
    B, Cc, H, W = out.shape
    sg, d = L.src(g), L.dst(out)
    sa = L.src(act) if act is not None else None
    L.check(L.lib().pc_convt2x2_dgrad(C.byref(sg), L.ptr(w), C.byref(sa) if sa is not None else None,
                                      C.byref(act_bn) if act_bn is not None else None, C.byref(d), B, H, W, Cc,
                                      L.stream_ptr()), "pc_convt2x2_dgrad")
    return out


def convt2x2_wgrad(x, g, dw=None, db=None, accumulate=False):
    L.require_device(x, g)
    B, Cc, H, W = x.shape
    if dw is None:
        dw = torch.empty(Cc, Cc, 2, 2, device=x.device, dtype=torch.float32)
    if db is None:
        db = torch.empty(Cc, device=x.device, dtype=torch.float32)
    ws = _workspace(L.lib().pc_convt2x2_wgrad_ws_bytes(Cc), x.device)
    sx, sg = L.src(x), L.src(g)
    L.check(L.lib().pc_convt2x2_wgrad(C.byref(sx), C.byref(sg), L.ptr(dw), L.ptr(db), int(accumulate), L.ptr(ws),
                                      B, H, W, Cc, L.stream_ptr()), "pc_convt2x2_wgrad")
    return dw, db


def outconv_sigmoid_crop(feat, w, bias, H, W, py, px, out=None):
    """fusion_out_conv (1x1, 16->1) + sigmoid + crop.  popcorn.py:301,317-320."""
    L.require_device(feat, w, bias)
    B = feat.shape[0]
    if out is None:
        out = torch.empty(B, 1, H, W, device=feat.device, dtype=torch.float32)
    sf, d = L.src(feat), L.dst(out)
    L.check(L.lib().pc_outconv_sigmoid_crop(C.byref(sf), L.ptr(w), L.ptr(bias), C.byref(d), B, H, W, py, px,
                                            L.stream_ptr()), "pc_outconv_sigmoid_crop")
    return out


def sparsity_mask(building, admin_mask, census_idx, rowsel, colsel, occupancymodel=True):
    """popcorn.py:361-377.  rowsel/colsel: uint8 device vectors.  Returns (mask uint8 (B,H,W), counts int32[2])."""
    L.require_device(building, admin_mask, census_idx, rowsel, colsel)
    B, H, W = admin_mask.shape
    mask = torch.empty(B, H, W, dtype=torch.uint8, device=building.device)
    counts = torch.empty(2, dtype=torch.int32, device=building.device)
    L.check(L.lib().pc_sparsity_mask(L.ptr(building), L.ptr(admin_mask), L.ptr(census_idx), L.ptr(rowsel), L.ptr(colsel),
                                     int(occupancymodel), L.ptr(mask), L.ptr(counts), B, H, W, L.stream_ptr()),
            "pc_sparsity_mask")
    return mask, counts


def sparsity_mask_unet(building, admin_mask, census_idx, rowsel, colsel, threshold=0.001):
    """get_sparsity_mask(sparse_unet=True), popcorn.py:336-359.  Returns (mask uint8 (B,H,W), ratio float32 (B,))."""
    L.require_device(building, admin_mask, census_idx, rowsel, colsel)
    B, H, W = admin_mask.shape
    mask = torch.empty(B, H, W, dtype=torch.uint8, device=building.device)
    ratio = torch.empty(B, dtype=torch.float32, device=building.device)
    L.check(L.lib().pc_sparsity_mask_unet(L.ptr(building), L.ptr(admin_mask), L.ptr(census_idx), L.ptr(rowsel), L.ptr(colsel),
                                          C.c_float(threshold), L.ptr(mask), L.ptr(ratio), B, H, W, L.stream_ptr()),
            "pc_sparsity_mask_unet")
    return mask, ratio


def building_score_mask(feat, w, bias, H, W, py, px, admin_mask, census_idx, rowsel, colsel, occupancymodel=True):
    """outconv_sigmoid_crop + sparsity_mask in one launch.  Returns (building (B,1,H,W), mask uint8 (B,H,W), counts)."""
    L.require_device(feat, w, bias, admin_mask, census_idx, rowsel, colsel)
    B = feat.shape[0]
    dev = feat.device
    building = torch.empty(B, 1, H, W, device=dev, dtype=torch.float32)
    mask = torch.empty(B, H, W, dtype=torch.uint8, device=dev)
    counts = torch.empty(2, dtype=torch.int32, device=dev)
    sf, d = L.src(feat), L.dst(building)
    L.check(L.lib().pc_building_score_mask(C.byref(sf), L.ptr(w), L.ptr(bias), C.byref(d), L.ptr(admin_mask),
                                           L.ptr(census_idx), L.ptr(rowsel), L.ptr(colsel), int(occupancymodel),
                                           L.ptr(mask), L.ptr(counts), B, H, W, py, px, L.stream_ptr()),
            "pc_building_score_mask")
    return building, mask, counts


def _hw_array(head_tensors):
    arr = (C.c_void_p * 8)(*[t.data_ptr() for t in head_tensors])
    return arr


PC_HEAD_FWD_PACK_BOTH, PC_HEAD_BWD_PACKED, PC_HEAD_FWD_DEFER_REDUCE, PC_HEAD_BWD_DEFER_REDUCE = 1, 1, 2, 2


def head_fwd(feat, py, px, H, W, head_tensors, building, mask=None, admin_mask=None, census_idx=None,
             want_scale=True, stats=None, nsel_counts=None, pack_both=False, defer_reduce=False):
    """Sparse/dense head + occupancy product + census reduction.  popcorn.py:161-190.
    head_tensors = [w0,b0,w2,b2,w4,b4,w6,b6].  Returns (scale_map, popdensemap, popcount)."""
    L.require_device(feat, building, *head_tensors)
    B = feat.shape[0]
    dev = feat.device
    scale_map = torch.empty(B, H, W, device=dev, dtype=torch.float32) if want_scale else None
    popdense = torch.empty(B, H, W, device=dev, dtype=torch.float32)
    popcount = torch.empty(B, device=dev, dtype=torch.float32)
    ws = _workspace(L.lib().pc_head_ws_bytes(B, H, W), dev)
    sf = L.src(feat)
    hw = _hw_array(head_tensors)
    L.check(L.lib().pc_head_fwd(C.byref(sf), py, px, hw, L.ptr(mask), L.ptr(building), L.ptr(admin_mask),
                                L.ptr(census_idx), L.ptr(scale_map), L.ptr(popdense), L.ptr(popcount), L.ptr(stats),
                                L.ptr(nsel_counts), L.ptr(ws), B, H, W,
                                (PC_HEAD_FWD_PACK_BOTH if pack_both else 0) | (PC_HEAD_FWD_DEFER_REDUCE if defer_reduce else 0), L.stream_ptr()),
            "pc_head_fwd")
    return scale_map, popdense, popcount


def compact_masked(src, mask):
    """src[mask] in row-major order (popcorn.py:173).  Returns (buffer of src.numel() floats, device int32 count)."""
    L.require_device(src, mask)
    n = src.numel()
    out = torch.empty(n, device=src.device, dtype=torch.float32)
    cnt = torch.empty(1, device=src.device, dtype=torch.int32)
    ws = _workspace(L.lib().pc_compact_ws_bytes(n), src.device)
    L.check(L.lib().pc_compact_masked(L.ptr(src), L.ptr(mask), L.ptr(out), L.ptr(cnt), L.ptr(ws), C.c_int64(n),
                                      L.stream_ptr()), "pc_compact_masked")
    return out, cnt


def scatter_masked(src, mask):
    """out[mask] = src (row-major), zeros elsewhere: the autograd of compact_masked / of the reference's ``scale[mask]``."""
    L.require_device(src, mask)
    n = mask.numel()
    out = torch.empty(mask.shape, device=mask.device, dtype=torch.float32)
    ws = _workspace(L.lib().pc_compact_ws_bytes(n), mask.device)
    L.check(L.lib().pc_scatter_masked(L.ptr(src), L.ptr(mask), L.ptr(out), L.ptr(ws), C.c_int64(n), L.stream_ptr()),
            "pc_scatter_masked")
    return out


def reflect_pad(x, top, bottom, left, right):
    """F.pad(x, (left, right, top, bottom), mode='reflect') for a contiguous (..., H, W) fp32 tensor."""
    L.require_device(x)
    x = x.contiguous()
    H, W = x.shape[-2:]
    out = torch.empty(*x.shape[:-2], H + top + bottom, W + left + right, device=x.device, dtype=torch.float32)
    planes = x.numel() // (H * W)
    L.check(L.lib().pc_reflect_pad(L.ptr(x), L.ptr(out), C.c_int64(planes), H, W, top, bottom, left, right, L.stream_ptr()),
            "pc_reflect_pad")
    return out


def reflect_pad_select(x, sel, top, bottom, left, right):
    """out[b][j] = F.pad(x[b][sel[j]], reflect): the padded, channel-gathered copy of a (B, C, H, W) fp32 input."""
    L.require_device(x)
    x = x.contiguous()
    B, Cin, H, W = x.shape
    out = torch.empty(B, len(sel), H + top + bottom, W + left + right, device=x.device, dtype=torch.float32)
    arr = (C.c_int * len(sel))(*[int(v) for v in sel])
    L.check(L.lib().pc_reflect_pad_select(L.ptr(x), L.ptr(out), B, Cin, len(sel), arr, H, W, top, bottom, left, right, L.stream_ptr()),
            "pc_reflect_pad_select")
    return out


def select_normalize_pad(raw, band, mean, std, top, bottom, left, right, out=None):
    """Band selection + (x - mean) / std + reflect padding in one pass: out[b][j] = pad((raw[b][band[j]] - mean[j]) / std[j])."""
    L.require_device(raw)
    assert raw.is_contiguous() and raw.dtype == torch.float32
    B, Craw, H, W = raw.shape
    n = len(band)
    if out is None:
        out = torch.empty(B, n, H + top + bottom, W + left + right, device=raw.device, dtype=torch.float32)
    L.check(L.lib().pc_select_normalize_pad(L.ptr(raw), L.ptr(out), B, Craw, n, (C.c_int * n)(*[int(v) for v in band]),
                                            (C.c_float * n)(*[float(v) for v in mean]), (C.c_float * n)(*[float(v) for v in std]),
                                            H, W, top, bottom, left, right, L.stream_ptr()), "pc_select_normalize_pad")
    return out


def ingest_cl8(raw, band, mean, std, top, bottom, left, right, out=None):
    """PC_PREC_BF16 ingest (pc_ingest_cl8): band selection + (x - mean) / std (mean / std None: already normalised) + reflect padding
    of a planar fp32 (B, Craw, H, W) tile into ONE channels-last bf16 tensor (B, 8, Hp, Wp): channel j = the j-th selected band,
    channels >= len(band) zero."""
    L.require_device(raw)
    assert raw.is_contiguous() and raw.dtype == torch.float32
    B, Craw, H, W = raw.shape
    n = len(band)
    if out is None:
        out = torch.empty(B, 8, H + top + bottom, W + left + right, device=raw.device, dtype=torch.bfloat16, memory_format=torch.channels_last)
    assert out.stride(1) == 1 and out.stride(3) == 8
    fa = lambda v: None if v is None else (C.c_float * n)(*[float(t) for t in v])  # noqa: E731
    L.check(L.lib().pc_ingest_cl8(L.ptr(raw), L.ptr(out), B, Craw, n, (C.c_int * n)(*[int(v) for v in band]), fa(mean), fa(std),
                                  H, W, top, bottom, left, right, L.stream_ptr()), "pc_ingest_cl8")
    return out


def ingest_split(s2_u16, s1, band, mean, std, top, bottom, left, right, cl8=None):
    """pc_ingest_split: the one-pass ingest (band select + normalise + reflect padding + stream order) from the two tensors a loader
    ships -- s2_u16 (B, C2, H, W) uint16 reflectance digital numbers, s1 (B, C1, H, W) fp32 -- into the padded model input of the
    current arithmetic mode (cl8 None): planar fp32 (B, n, Hp, Wp) or the channels-last bf16 slot tensor (B, 8, Hp, Wp).  band indexes
    [s2 | s1]."""
    L.require_device(s2_u16, s1)
    assert s2_u16.dtype == torch.uint16 and s1.dtype == torch.float32 and s2_u16.is_contiguous() and s1.is_contiguous()
    B, C2, H, W = s2_u16.shape
    C1 = s1.shape[1]
    assert tuple(s1.shape) == (B, C1, H, W)
    n = len(band)
    if cl8 is None:
        cl8 = L.act_dtype() == torch.bfloat16
    Hp, Wp = H + top + bottom, W + left + right
    if cl8:
        out = torch.empty(B, 8, Hp, Wp, device=s1.device, dtype=torch.bfloat16, memory_format=torch.channels_last)
    else:
        out = torch.empty(B, n, Hp, Wp, device=s1.device, dtype=torch.float32)
    fa = lambda v: (C.c_float * n)(*[float(t) for t in v])  # noqa: E731
    L.check(L.lib().pc_ingest_split(L.ptr(s2_u16), C2, L.ptr(s1), C1, L.ptr(out), int(bool(cl8)), B, n, (C.c_int * n)(*[int(v) for v in band]),
                                    fa(mean), fa(std), H, W, top, bottom, left, right, L.stream_ptr()), "pc_ingest_split")
    return out


def head_bwd(feat, py, px, H, W, head_tensors, building, mask=None, admin_mask=None, census_idx=None,
             g_popcount=None, g_popdense=None, g_scale_map=None, g_scale_const=None, grads=None, accumulate=False,
             g_feat=None, feat_bn=None, packed=False, defer_reduce=False):
    """Backward of head_fwd.  Returns (list of 8 head grads, g_feat (B,16,Hp,Wp)).  pack_both (head_fwd) / packed (here): a training
    step's forward call assembles the backward's weight image too (same weights, same workspace): one launch less.  defer_reduce:
    the weight gradients stay per-workgroup partials; returns (HeadPartials, g_feat) instead -- hand the first to the
    ``WgradBatch`` of the same backward pass (``head_reduce``), whose batched reduction writes ``grads``: one launch less."""
    L.require_device(feat, building, *head_tensors)
    B, _, Hp, Wp = feat.shape
    dev = feat.device
    if grads is None:
        grads = [torch.empty_like(t) for t in head_tensors]
    if g_feat is None:
        g_feat = L.empty_act(B, 16, Hp, Wp, dev)
    ws = _workspace(L.lib().pc_head_ws_bytes(B, H, W), dev)
    sf, d = L.src(feat), L.dst(g_feat)
    hw = _hw_array(head_tensors)
    dhw = (C.c_void_p * 8)(*[0 if t is None else t.data_ptr() for t in grads])
    L.check(L.lib().pc_head_bwd(C.byref(sf), py, px, hw, L.ptr(mask), L.ptr(building), L.ptr(admin_mask),
                                L.ptr(census_idx), L.ptr(g_popcount), L.ptr(g_popdense), L.ptr(g_scale_map),
                                L.ptr(g_scale_const), dhw, int(accumulate), C.byref(d),
                                C.byref(feat_bn[0]) if feat_bn else None, C.byref(feat_bn[1]) if feat_bn else None,
                                Hp, Wp, L.ptr(ws), B, H, W,
                                (PC_HEAD_BWD_PACKED if packed else 0) | (PC_HEAD_BWD_DEFER_REDUCE if defer_reduce else 0),
                                L.stream_ptr()), "pc_head_bwd")
    if defer_reduce:
        pp, nwg = C.c_void_p(0), C.c_int(0)
        L.check(L.lib().pc_head_bwd_partials(L.ptr(ws), B, H, W, C.byref(pp), C.byref(nwg)), "pc_head_bwd_partials")
        return HeadPartials(pp.value, nwg.value, list(grads), bool(accumulate)), g_feat
    return grads, g_feat


class HeadPartials:
    """The unreduced weight-gradient partials of a ``head_bwd(defer_reduce=True)`` call: (device pointer, workgroups, the 8 gradient
    tensors they are to be summed into -- None = skip --, accumulate)."""

    def __init__(self, ptr_, nwg, grads, accumulate):
        self.ptr, self.nwg, self.grads, self.accumulate = ptr_, nwg, grads, accumulate


def select_normalize(raw, band6, mean6, std6, out=None):
    """Band selection + (x - mean) / std.  PopulationDataset.py:566-568 + utils/utils.py:105-127."""
    L.require_device(raw)
    B, Craw, H, W = raw.shape
    assert raw.is_contiguous() and raw.dtype == torch.float32
    if out is None:
        out = torch.empty(B, 6, H, W, device=raw.device, dtype=torch.float32)
    L.check(L.lib().pc_select_normalize(L.ptr(raw), Craw, (C.c_int * 6)(*band6), (C.c_float * 6)(*mean6),
                                        (C.c_float * 6)(*std6), L.ptr(out), B, H, W, L.stream_ptr()),
            "pc_select_normalize")
    return out


def augment_raw(s2, s1, admin_mask, params, out=None, admin_out=None):
    """The trainer's augmentations (run_train.py:386-402) with parameters drawn on the host (utils/transform.py: draw_fused_params), applied
    while the raw 6-channel tile [S2 | S1] is assembled -- ONE launch (pc_augment_raw).  s2 (B, 4, H, W) digital numbers, s1 (B, 2, H, W),
    admin_mask (B, H, W), fp32 device tensors; returns (raw (B, 6, Ho, Wo), admin (B, Ho, Wo)); params None = no augmentation."""
    L.require_device(s2, s1, admin_mask)
    B, c2, H, W = s2.shape
    if c2 != 4 or tuple(s1.shape) != (B, 2, H, W) or tuple(admin_mask.shape) != (B, H, W):
        raise ValueError(f"augment_raw: S2 (B, 4, H, W), S1 (B, 2, H, W), admin_mask (B, H, W); got {tuple(s2.shape)}, {tuple(s1.shape)}, "
                         f"{tuple(admin_mask.shape)}")
    for t in (s2, s1, admin_mask):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError("augment_raw: contiguous fp32 tensors")
    pr = params or {}
    k = int(pr.get("rot", 0)) & 3
    Ho, Wo = (W, H) if k & 1 else (H, W)
    if out is None:
        out = torch.empty(B, 6, Ho, Wo, device=s2.device, dtype=torch.float32)
    if admin_out is None:
        admin_out = torch.empty(B, Ho, Wo, device=s2.device, dtype=torch.float32)
    beta, gamma = pr.get("beta"), pr.get("gamma")
    L.check(L.lib().pc_augment_raw(L.ptr(s2), L.ptr(s1), L.ptr(admin_mask), L.ptr(out), L.ptr(admin_out), B, H, W,
                                   int(bool(pr.get("vflip"))), int(bool(pr.get("hflip"))), k, int(beta is not None),
                                   C.c_float(beta if beta is not None else 1.0), int(gamma is not None),
                                   C.c_float(gamma if gamma is not None else 1.0), L.stream_ptr()), "pc_augment_raw")
    return out, admin_out


def loss_fwd_bwd(popcount, y, stats, lam4, scale_regularization, lam_weak, inv_B, loss_out, g_popcount, g_scale_const):
    L.require_device(popcount, y)
    L.check(L.lib().pc_loss_fwd_bwd(L.ptr(popcount), L.ptr(y), L.ptr(stats), (C.c_float * 4)(*lam4),
                                    C.c_float(scale_regularization), C.c_float(lam_weak), C.c_float(inv_B),
                                    popcount.numel(), L.ptr(loss_out), L.ptr(g_popcount), L.ptr(g_scale_const),
                                    L.stream_ptr()), "pc_loss_fwd_bwd")


def head_popcount_loss(B, H, W, nsel_counts, y, lam4, scale_regularization, lam_weak, inv_B, popcount, stats, loss_out, g_popcount,
                       g_scale_const):
    """Finishes a ``head_fwd(..., defer_reduce=True)`` call -- popcount[B] and stats {Nsel, sum scale} from the per-chunk partials in the
    head workspace -- and computes the loss forward + backward (``loss_fwd_bwd``) in the same single-block launch."""
    L.require_device(y, popcount)
    ws = _workspace(L.lib().pc_head_ws_bytes(B, H, W), y.device)
    L.check(L.lib().pc_head_popcount_loss(L.ptr(ws), B, H, W, L.ptr(nsel_counts), L.ptr(y), (C.c_float * 4)(*lam4),
                                          C.c_float(scale_regularization), C.c_float(lam_weak), C.c_float(inv_B), L.ptr(popcount),
                                          L.ptr(stats), L.ptr(loss_out), L.ptr(g_popcount), L.ptr(g_scale_const), L.stream_ptr()),
            "pc_head_popcount_loss")


def grad_norm(flat, norm_out):
    L.require_device(flat)
    L.check(L.lib().pc_grad_norm(L.ptr(flat), flat.numel(), L.ptr(norm_out), L.stream_ptr()), "pc_grad_norm")


def adam_groups(segments, active):
    """pc_adam_groups from [(end_offset, group id), ...] (consecutive segments of the flat buffer) and the set of group
    ids that are updated in this call; the others are skipped like parameters whose .grad is None in torch.optim.Adam."""
    g = L.PcAdamGroups()
    assert 1 <= len(segments) <= L.PC_ADAM_MAX_SEG
    g.nseg = len(segments)
    for i, (end, grp) in enumerate(segments):
        assert 0 <= grp < L.PC_ADAM_GROUPS
        g.seg_end[i], g.seg_group[i] = int(end), int(grp)
    g.active_mask = sum(1 << int(a) for a in set(active))
    return g


def _check_steps(step, groups):
    need = L.PC_ADAM_GROUPS if groups is not None else 1
    assert step.dtype == torch.int32 and step.numel() >= need, "step: one int32 counter per Adam group"


def adam_clip_step(p, g, m, v, n_decay, hyper, weight_decay, beta1, beta2, eps, max_norm, norm, step, groups=None):
    L.require_device(p, g, m, v)
    _check_steps(step, groups)
    L.check(L.lib().pc_adam_clip_step(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), p.numel(), n_decay, L.ptr(hyper),
                                      C.c_float(weight_decay), C.c_float(beta1), C.c_float(beta2), C.c_float(eps),
                                      C.c_float(max_norm), L.ptr(norm), L.ptr(step),
                                      C.byref(groups) if groups is not None else None, L.stream_ptr()),
            "pc_adam_clip_step")


def adam_clip_step_fused(p, g, m, v, n_decay, hyper, weight_decay, beta1, beta2, eps, max_norm, norm_out, step, groups=None):
    """grad-norm + clip + Adam in one launch (norm_out receives the total norm).  groups: ops.adam_groups(...) or None."""
    L.require_device(p, g, m, v, hyper, step)
    _check_steps(step, groups)
    L.check(L.lib().pc_adam_clip_step_fused(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), p.numel(), n_decay, L.ptr(hyper),
                                            C.c_float(weight_decay), C.c_float(beta1), C.c_float(beta2), C.c_float(eps),
                                            C.c_float(max_norm), L.ptr(norm_out), L.ptr(step),
                                            C.byref(groups) if groups is not None else None, L.stream_ptr()),
            "pc_adam_clip_step_fused")


class WgradBatch:
    """Collects the first-stage (per-workgroup partial) weight-gradient launches of a backward pass and finishes them
    with ONE batched reduction launch.  Each layer gets its own slice of a persistent workspace."""

    _ws = {}

    def __init__(self, device, accumulate=False):
        self.device = device
        self.accumulate = accumulate
        self.entries = []
        self.raw_entries = []           # (partials ptr, total ptr, nwg, floats): raw sums (kind 2)
        self.chains = []                # composed Up blocks: chain-rule launches that follow the reduction
        self.head = None                # HeadPartials of the pass's head_bwd(defer_reduce=True): finished by the same launch
        self.slot_bytes = int(max(L.lib().pc_conv3x3_wgrad_ws_bytes(32, 8), L.lib().pc_convt2x2_wgrad_ws_bytes(16)))
        self.slot = 0

    def _slice(self, nbytes=0):
        """One workspace slot (or as many consecutive slots as ``nbytes`` needs)."""
        k = max(1, -(-int(nbytes) // self.slot_bytes))
        key = str(self.device)
        need = (self.slot + k) * self.slot_bytes
        buf = WgradBatch._ws.get(key)
        if buf is None or buf.numel() < need:
            if buf is not None:
                _ws_retired.append(buf)       # earlier entries of this batch / captured graphs still point into it
            nbuf = torch.empty(max(need, 32 * self.slot_bytes), dtype=torch.uint8, device=self.device)
            WgradBatch._ws[key] = buf = nbuf
        ptr_ = buf.data_ptr() + self.slot * self.slot_bytes
        self.slot += k
        return ptr_

    def conv3x3(self, a, g, cout, dw, db, b=None, a_mode=L.PC_SRC_DIRECT, a_pad=(0, 0), chmap=(0, 1, 2, 3),
                b_offset=(0, 0), a_channels=None):
        B, Cg, H, W = g.shape
        Ca = a.shape[1] if a_channels is None else a_channels
        Cb = 0 if b is None else b.shape[1]
        sa = L.src(a, C_=Ca, mode=a_mode, oy=a_pad[0], ox=a_pad[1], chmap=chmap)
        sb = L.src(b, oy=b_offset[0], ox=b_offset[1]) if b is not None else None
        sg = L.src(g)
        ws = self._slice()
        nwg = C.c_int(0)
        L.check(L.lib().pc_conv3x3_wgrad_partial(C.byref(sa), C.byref(sb) if sb is not None else None, C.byref(sg),
                                                 C.c_void_p(ws), B, H, W, Ca + Cb, cout, C.byref(nwg), L.stream_ptr()),
                "pc_conv3x3_wgrad_partial")
        self.entries.append((ws, dw, db, nwg.value, Ca + Cb, cout, 0))

    def conv3x3_group(self, problems, cout, a_mode=L.PC_SRC_DIRECT, a_pad=(0, 0), a_channels=None, cin_total=None):
        """problems: list of dicts {a, g, dw, db, b (opt), b_offset (opt), chmap (opt)} of identical geometry -> one launch.
        cin_total: dw is the gradient of a wider weight [cout][cin_total][3][3] whose FIRST Ca (+ Cb) input channels these are"""
        n = len(problems)
        g0, a0 = problems[0]["g"], problems[0]["a"]
        B, Cg, H, W = g0.shape
        Ca = a0.shape[1] if a_channels is None else a_channels
        Cb = 0 if problems[0].get("b") is None else problems[0]["b"].shape[1]
        keep, descs, slots = [], (L.PcConvWgradDesc * n)(), []
        for i, pr in enumerate(problems):
            sa = L.src(pr["a"], C_=Ca, mode=a_mode, oy=a_pad[0], ox=a_pad[1], chmap=pr.get("chmap", (0, 1, 2, 3)))
            bo = pr.get("b_offset", (0, 0))
            sb = L.src(pr["b"], oy=bo[0], ox=bo[1]) if pr.get("b") is not None else None
            sg = L.src(pr["g"])
            ws = self._slice()
            keep += [sa, sb, sg]
            slots.append(ws)
            descs[i].a, descs[i].g, descs[i].ws = C.pointer(sa), C.pointer(sg), ws
            descs[i].b = C.pointer(sb) if sb is not None else None
        nwg = C.c_int(0)
        L.check(L.lib().pc_conv3x3_wgrad_partial_group(n, descs, B, H, W, Ca + Cb, cout, C.byref(nwg), L.stream_ptr()),
                "pc_conv3x3_wgrad_partial_group")
        for ws, pr in zip(slots, problems):
            if pr.get("src_window") is not None:
                # the partials cover all Ca input channels; dw ([cout][cin][3][3]) receives the channels [ci0, ci0 + cin) only
                ci0, cin = pr["src_window"]
                assert pr["dw"].shape[1] == cin and ci0 + cin <= Ca + Cb and cin_total is None
                self.entries.append((ws, pr["dw"], pr["db"], nwg.value, cin, cout, 0, 0, 0, Ca + Cb, ci0))
            elif cin_total is None:
                self.entries.append((ws, pr["dw"], pr["db"], nwg.value, Ca + Cb, cout, 0))
            else:
                self.entries.append((ws, pr["dw"], pr["db"], nwg.value, Ca + Cb, cout, 0, cin_total * 9, 0))

    def conv3x3_bwd_group(self, problems, cin_total, c0, accumulate=False):
        """bf16 mode: data gradient + weight-gradient partials of a conv layer (g: 8 / 16 channels, x: an 8- or 16-channel column
        block of the layer's input) in ONE launch.  problems: list of dicts {g, x, w, out, dw (the full [Cg][cin_total][3][3]
        gradient), db (or None), x_bn (opt: ReLU / BN factor of x's producer), x_offset (opt), c0_add (opt: this problem's column
        block starts at c0 + c0_add -- the blocks of a concat layer in one launch)}; the partials cover the columns
        [c0 + c0_add, c0 + c0_add + x.shape[1]) of dw."""
        n = len(problems)
        g0 = problems[0]["g"]
        B, Cg, H, W = g0.shape
        keep, descs, slots = [], (L.PcConvBwdDesc * n)(), []
        for i, pr in enumerate(problems):
            xo = pr.get("x_offset", (0, 0))
            sg, sx, d = L.src(pr["g"]), L.src(pr["x"], oy=xo[0], ox=xo[1]), L.dst(pr["out"])
            ws = self._slice()
            keep += [sg, sx, d]
            slots.append(ws)
            descs[i].g, descs[i].x, descs[i].w, descs[i].out, descs[i].ws = C.pointer(sg), C.pointer(sx), pr["w"].data_ptr(), C.pointer(d), ws
            descs[i].x_bn = C.pointer(pr["x_bn"]) if pr.get("x_bn") is not None else None
            descs[i].c0_add = int(pr.get("c0_add", 0))
            if pr.get("pool_act") is not None:        # Down blocks: x is the pooled copy of pool_act; out is at twice the resolution (+=)
                spa = L.src(pr["pool_act"])
                keep.append(spa)
                descs[i].pool_act = C.pointer(spa)
        nwg = C.c_int(0)
        L.check(L.lib().pc_conv3x3_bwd_group(n, descs, cin_total, c0, int(accumulate), B, H, W, C.byref(nwg), L.stream_ptr()),
                "pc_conv3x3_bwd_group")
        for ws, pr in zip(slots, problems):
            self.entries.append((ws, pr["dw"], pr.get("db"), nwg.value, pr["x"].shape[1], Cg, 0, cin_total * 9,
                                 (c0 + int(pr.get("c0_add", 0))) * 9))

    def up_bwd_group(self, problems):
        """Composed Up block backward, deferred form: the pass now (data gradient written, partials queued), the reduction inside
        this batch's one reduce launch, the chain rule right after it (``finish``).  problems: as ops.conv3x3_up_bwd_group."""
        n = len(problems)
        g0, z0 = problems[0]["g"], problems[0]["z"]
        B, Cg, H, W = g0.shape
        Cz = z0.shape[1]
        nbytes = int(L.lib().pc_conv3x3_up_bwd_ws_bytes(B, H, Cz))
        descs = (L.PcConvUpBwdDesc * n)()
        keep, slots = [], []
        for i, pr in enumerate(problems):
            sg, sz = L.src(pr["g"]), L.src(pr["z"])
            dz = L.dst(pr["gz"]) if pr.get("gz") is not None else None
            ws = self._slice(nbytes)
            keep += [sg, sz, dz]
            slots.append(ws)
            descs[i].g, descs[i].z = C.pointer(sg), C.pointer(sz)
            descs[i].z_bn = C.pointer(pr["z_bn"])
            descs[i].gz = C.pointer(dz) if dz is not None else None
            descs[i].w, descs[i].wt = pr["w"].data_ptr(), pr["wt"].data_ptr()
            descs[i].bt = pr["bt"].data_ptr() if pr.get("bt") is not None else None
            descs[i].fwd_ws = pr["fwd_ws"].data_ptr()
            descs[i].ws = ws
            descs[i].dw, descs[i].dwt = pr["dw"].data_ptr(), pr["dwt"].data_ptr()
            descs[i].dbt = pr["dbt"].data_ptr() if pr.get("dbt") is not None else None
        nwg, part = C.c_int(0), C.c_int(0)
        L.check(L.lib().pc_conv3x3_up_bwd_partial_group(n, descs, B, H, W, Cz, Cz, C.byref(nwg), C.byref(part), L.stream_ptr()),
                "pc_conv3x3_up_bwd_partial_group")
        for ws in slots:
            self.raw_entries.append((ws, ws + 4 * nwg.value * part.value, nwg.value, part.value))
        self.chains.append((descs, keep, n, nwg.value, Cz, list(problems)))

    def level2_bwd_group(self, problems):
        """Backward of down2's DoubleConv (the 32 x 32 level) in ONE launch: problems = list of {g2 (dL/d conv2 output, masked),
        c1, x (pooled input), w1, w2, bn1 (BN of conv1: relu'(c1) factor), act (b2, full resolution), act_bn, out (G_b2, +=),
        dw1, db1, dw2, db2}.  Queues the two layers' weight / bias gradient partials; the data gradients never leave the kernel."""
        n = len(problems)
        B = problems[0]["g2"].shape[0]
        nbytes = int(L.lib().pc_level2_bwd_ws_bytes(B))
        descs = (L.PcLevel2BwdDesc * n)()
        keep, slices = [], []
        for i, pr in enumerate(problems):
            sg, sc, sx, sa, do = L.src(pr["g2"]), L.src(pr["c1"]), L.src(pr["x"]), L.src(pr["act"]), L.dst(pr["out"])
            ws1, ws2 = self._slice(nbytes), self._slice(nbytes)
            keep += [sg, sc, sx, sa, do]
            slices.append((ws1, ws2))
            descs[i].g2, descs[i].c1, descs[i].x, descs[i].act, descs[i].out = C.pointer(sg), C.pointer(sc), C.pointer(sx), C.pointer(sa), C.pointer(do)
            descs[i].w1, descs[i].w2 = pr["w1"].data_ptr(), pr["w2"].data_ptr()
            descs[i].bn1, descs[i].act_bn = C.pointer(pr["bn1"]), C.pointer(pr["act_bn"])
            descs[i].ws1, descs[i].ws2 = ws1, ws2
        nwg = C.c_int(0)
        L.check(L.lib().pc_level2_bwd_group(n, descs, B, C.byref(nwg), L.stream_ptr()), "pc_level2_bwd_group")
        for (ws1, ws2), pr in zip(slices, problems):
            self.entries.append((ws1, pr["dw1"], pr["db1"], nwg.value, 16, 16, 0))
            self.entries.append((ws2, pr["dw2"], pr["db2"], nwg.value, 16, 16, 0))

    def convt2x2(self, x, g, dw, db):
        B, Cc, H, W = x.shape
        sx, sg = L.src(x), L.src(g)
        ws = self._slice()
        nwg = C.c_int(0)
        L.check(L.lib().pc_convt2x2_wgrad_partial(C.byref(sx), C.byref(sg), C.c_void_p(ws), B, H, W, Cc, C.byref(nwg),
                                                  L.stream_ptr()), "pc_convt2x2_wgrad_partial")
        self.entries.append((ws, dw, db, nwg.value, Cc, Cc, 1))

    def convt2x2_group(self, problems):
        """problems: list (<= 4) of dicts {x, g, dw, db} with identical geometry: one launch."""
        n = len(problems)
        x0 = problems[0]["x"]
        B, Cc, H, W = x0.shape
        descs = (L.PcConvtWgradDesc * n)()
        keep, slices = [], []
        for i, pr in enumerate(problems):
            sx, sg = L.src(pr["x"]), L.src(pr["g"])
            ws = self._slice()
            keep += [sx, sg]
            slices.append(ws)
            descs[i].x, descs[i].g, descs[i].ws = C.pointer(sx), C.pointer(sg), ws
        nwg = C.c_int(0)
        L.check(L.lib().pc_convt2x2_wgrad_partial_group(n, descs, B, H, W, Cc, C.byref(nwg), L.stream_ptr()),
                "pc_convt2x2_wgrad_partial_group")
        for ws, pr in zip(slices, problems):
            self.entries.append((ws, pr["dw"], pr["db"], nwg.value, Cc, Cc, 1))

    def convt2x2_bwd_group(self, problems):
        """Whole backward of a transposed conv in one launch: problems = list of {x, g, w, out, dw, db, x_bn (opt: ReLU / BN factor
        of x's producer)}; writes the data gradient into out and queues the weight / bias gradient partials."""
        n = len(problems)
        B, Cc, H, W = problems[0]["x"].shape
        descs = (L.PcConvtBwdDesc * n)()
        keep, slices = [], []
        for i, pr in enumerate(problems):
            sx, sg, d = L.src(pr["x"]), L.src(pr["g"]), L.dst(pr["out"])
            ws = self._slice()
            keep += [sx, sg, d]
            slices.append(ws)
            descs[i].x, descs[i].g, descs[i].w, descs[i].out, descs[i].ws = C.pointer(sx), C.pointer(sg), pr["w"].data_ptr(), C.pointer(d), ws
            descs[i].x_bn = C.pointer(pr["x_bn"]) if pr.get("x_bn") is not None else None
        nwg = C.c_int(0)
        L.check(L.lib().pc_convt2x2_bwd_group(n, descs, B, H, W, Cc, C.byref(nwg), L.stream_ptr()), "pc_convt2x2_bwd_group")
        for ws, pr in zip(slices, problems):
            self.entries.append((ws, pr["dw"], pr["db"], nwg.value, Cc, Cc, 1))

    def head_reduce(self, hp):
        assert self.head is None and isinstance(hp, HeadPartials)
        self.head = hp

    def finish(self):
        n = len(self.entries) + len(self.raw_entries) + (1 if self.head is not None else 0)
        if n == 0:
            return
        d = (L.PcWgradReduceDesc * n)()
        if self.head is not None:       # (kind 3: dw = HOST array of the 8 gradient tensors)
            hp, self.head = self.head, None
            dhw = (C.c_void_p * 8)(*[0 if t is None else t.data_ptr() for t in hp.grads])
            i = n - 1
            d[i].partial, d[i].dw, d[i].db = hp.ptr, C.cast(dhw, C.c_void_p).value, 0
            d[i].nwg, d[i].Cin, d[i].Cout, d[i].kind, d[i].accumulate, d[i].dw_co_stride = hp.nwg, 0, 0, 3, int(hp.accumulate), 0
        for i, ent in enumerate(self.entries):
            ws, dw, db, nwg, cin, cout, kind = ent[:7]
            d[i].partial = ws
            d[i].dw = dw.data_ptr() + (4 * ent[8] if len(ent) > 7 else 0)        # first column of an 8-channel column block
            d[i].dw_co_stride = ent[7] if len(ent) > 7 else 0
            d[i].db = 0 if db is None else db.data_ptr()
            d[i].nwg, d[i].Cin, d[i].Cout, d[i].kind, d[i].accumulate = nwg, cin, cout, kind, int(self.accumulate)
            if len(ent) > 9:
                d[i].src_cin, d[i].src_ci0 = ent[9], ent[10]
        for j, (pp, tot, nwg, part) in enumerate(self.raw_entries):
            i = len(self.entries) + j
            d[i].partial, d[i].dw, d[i].db = pp, tot, 0
            d[i].nwg, d[i].Cin, d[i].Cout, d[i].kind, d[i].accumulate, d[i].dw_co_stride = nwg, part, 0, 2, 0, 0
        L.check(L.lib().pc_wgrad_reduce_batch(n, d, L.stream_ptr()), "pc_wgrad_reduce_batch")
        c8 = [c for c in self.chains if c[4] == 8]
        c16 = [c for c in self.chains if c[4] == 16]
        if len(c8) == 1 and len(c16) == 1 and len(self.chains) == 2:        # the two Up levels of a U-Net backward: one launch
            L.check(L.lib().pc_conv3x3_up_chain_both(c8[0][2], c8[0][0], c8[0][3], c16[0][2], c16[0][0], c16[0][3],
                                                     int(self.accumulate), L.stream_ptr()), "pc_conv3x3_up_chain_both")
        else:
            for descs, keep, cn, nwg, Cz, probs in self.chains:
                L.check(L.lib().pc_conv3x3_up_chain_group(cn, descs, int(self.accumulate), nwg, Cz, Cz, L.stream_ptr()),
                        "pc_conv3x3_up_chain_group")
        self.entries, self.raw_entries, self.chains = [], [], []
        self.slot = 0
