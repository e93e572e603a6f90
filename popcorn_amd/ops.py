"""Op-level Python wrappers over the C ABI (one call = one kernel launch on the current stream).

Used by the kernel parity tests and by the nn.Module glue; the training hot path goes through the fused
composite entry points instead (popcorn_amd/engine.py).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


def conv3x3_bn_relu(a, w, bias, gamma=None, beta=None, mean=None, var=None, eps=1e-5, relu=True, b=None,
                    a_mode=L.PC_SRC_DIRECT, a_pad=(0, 0), chmap=(0, 1, 2, 3), out=None, out_hw=None,
                    b_offset=(0, 0), a_channels=None):
    """Conv2d(3x3, pad 1)(cat[a, b]) -> BN(eval) -> ReLU.  networks.py:259-266,318.

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
        out = torch.empty(B, Cout, H, W, device=a.device, dtype=torch.float32)
    bnd = L.bn(bias, gamma, beta, mean, var, eps)
    d = L.dst(out)
    code = L.lib().pc_conv3x3_bn_relu_fwd(C.byref(sa), C.byref(sb) if sb is not None else None, L.ptr(w), C.byref(bnd),
                                          int(relu), C.byref(d), B, H, W, Ca + Cb, Cout, L.stream_ptr())
    L.check(code, "pc_conv3x3_bn_relu_fwd")
    return out


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
