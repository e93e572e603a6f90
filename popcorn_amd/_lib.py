"""ctypes binding of libpopcorn_hip.so (include/popcorn_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C popcorn_amd/csrc``.  There is NO CPU
fallback: if the library is missing or a tensor is not on a HIP device, the ops raise.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("POPCORN_HIP_LIB") or os.path.join(_HERE, "libpopcorn_hip.so")   # override: A/B builds

PC_SRC_DIRECT, PC_SRC_POOL2, PC_SRC_REFLECT = 0, 1, 2


class PcSrc(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("bstride", C.c_int64), ("cstride", C.c_int64), ("rstride", C.c_int32), ("mode", C.c_int32),
                ("oy", C.c_int32), ("ox", C.c_int32), ("chmap", C.c_int32 * 4), ("dtype", C.c_int32), ("xstride", C.c_int32)]


class PcDst(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("bstride", C.c_int64), ("cstride", C.c_int64), ("rstride", C.c_int32),
                ("dtype", C.c_int32), ("xstride", C.c_int32), ("_pad", C.c_int32)]


class PcBn(C.Structure):
    _fields_ = [("conv_bias", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("mean", C.c_void_p),
                ("var", C.c_void_p), ("eps", C.c_float), ("_pad", C.c_int32)]


class PcConvFwdDesc(C.Structure):
    _fields_ = [("a", C.POINTER(PcSrc)), ("b", C.POINTER(PcSrc)), ("w", C.c_void_p), ("bn", C.POINTER(PcBn)),
                ("out", C.POINTER(PcDst)), ("pool_out", C.POINTER(PcDst)), ("dot_w", C.c_void_p), ("dot_out", C.POINTER(PcDst)),
                ("w_ci0", C.c_int32), ("w_cin", C.c_int32), ("upt_w", C.c_void_p), ("upt_b", C.c_void_p), ("upt_out", C.POINTER(PcDst))]


class PcConvDgradDesc(C.Structure):
    _fields_ = [("g", C.POINTER(PcSrc)), ("w", C.c_void_p), ("act", C.POINTER(PcSrc)), ("act_bn", C.POINTER(PcBn)),
                ("out", C.POINTER(PcDst))]


class PcConvtFwdDesc(C.Structure):
    _fields_ = [("x", C.POINTER(PcSrc)), ("w", C.c_void_p), ("bias", C.c_void_p), ("out", C.POINTER(PcDst))]


class PcConvtDgradDesc(C.Structure):
    _fields_ = [("g", C.POINTER(PcSrc)), ("w", C.c_void_p), ("act", C.POINTER(PcSrc)), ("act_bn", C.POINTER(PcBn)),
                ("out", C.POINTER(PcDst))]


class PcConvWgradDesc(C.Structure):
    _fields_ = [("a", C.POINTER(PcSrc)), ("b", C.POINTER(PcSrc)), ("g", C.POINTER(PcSrc)), ("ws", C.c_void_p)]


class PcConvtWgradDesc(C.Structure):
    _fields_ = [("x", C.POINTER(PcSrc)), ("g", C.POINTER(PcSrc)), ("ws", C.c_void_p)]


class PcConvtBwdDesc(C.Structure):
    _fields_ = [("x", C.POINTER(PcSrc)), ("g", C.POINTER(PcSrc)), ("w", C.c_void_p), ("x_bn", C.POINTER(PcBn)),
                ("out", C.POINTER(PcDst)), ("ws", C.c_void_p)]


class PcWgradReduceDesc(C.Structure):
    _fields_ = [("partial", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p), ("nwg", C.c_int32), ("Cin", C.c_int32),
                ("Cout", C.c_int32), ("kind", C.c_int32), ("accumulate", C.c_int32), ("dw_co_stride", C.c_int32),
                ("src_cin", C.c_int32), ("src_ci0", C.c_int32)]


class PcConvBwdDesc(C.Structure):
    _fields_ = [("g", C.POINTER(PcSrc)), ("x", C.POINTER(PcSrc)), ("w", C.c_void_p), ("x_bn", C.POINTER(PcBn)),
                ("out", C.POINTER(PcDst)), ("ws", C.c_void_p), ("pool_act", C.POINTER(PcSrc)), ("c0_add", C.c_int32),
                ("_pad", C.c_int32)]


class PcLevel2FwdDesc(C.Structure):
    _fields_ = [("x", C.POINTER(PcSrc)), ("w1", C.c_void_p), ("bn1", C.POINTER(PcBn)), ("w2", C.c_void_p), ("bn2", C.POINTER(PcBn)),
                ("wt", C.c_void_p), ("bt", C.c_void_p), ("c1", C.POINTER(PcDst)), ("c2", C.POINTER(PcDst)), ("u2", C.POINTER(PcDst))]


class PcConvUpFwdDesc(C.Structure):
    _fields_ = [("skip", C.POINTER(PcSrc)), ("z", C.POINTER(PcSrc)), ("w", C.c_void_p), ("wt", C.c_void_p), ("bt", C.c_void_p),
                ("bn", C.POINTER(PcBn)), ("out", C.POINTER(PcDst)), ("ws", C.c_void_p)]


class PcConvUpBwdDesc(C.Structure):
    _fields_ = [("g", C.POINTER(PcSrc)), ("z", C.POINTER(PcSrc)), ("z_bn", C.POINTER(PcBn)), ("gz", C.POINTER(PcDst)),
                ("w", C.c_void_p), ("wt", C.c_void_p), ("bt", C.c_void_p), ("fwd_ws", C.c_void_p), ("ws", C.c_void_p),
                ("dw", C.c_void_p), ("dwt", C.c_void_p), ("dbt", C.c_void_p)]


class PcLevel2BwdDesc(C.Structure):
    _fields_ = [("g2", C.POINTER(PcSrc)), ("c1", C.POINTER(PcSrc)), ("x", C.POINTER(PcSrc)), ("w1", C.c_void_p), ("w2", C.c_void_p),
                ("bn1", C.POINTER(PcBn)), ("act", C.POINTER(PcSrc)), ("act_bn", C.POINTER(PcBn)), ("out", C.POINTER(PcDst)),
                ("ws1", C.c_void_p), ("ws2", C.c_void_p)]


PC_ABI_VERSION = 9
PC_MAX_GROUP = 4
PC_ADAM_MAX_SEG, PC_ADAM_GROUPS = 8, 4
PC_EINVAL, PC_ENOGPU, PC_ENOMEM, PC_ENOTSUP = -1, -2, -3, -4


class PcAdamGroups(C.Structure):
    _fields_ = [("nseg", C.c_int32), ("seg_end", C.c_int32 * PC_ADAM_MAX_SEG), ("seg_group", C.c_int32 * PC_ADAM_MAX_SEG),
                ("active_mask", C.c_int32)]


# ---- native step executor (pc_train_step) ------------------------------------------------------------------------------------------
PC_STEP_CONVS = 10
PC_DATA_INPUT, PC_DATA_RAW, PC_DATA_SPLIT = 0, 1, 2
PC_STEP_FWD, PC_STEP_BWD, PC_STEP_UPD = 1, 2, 4
PC_STEP_SEL_MAX = 16384


class PcStepStream(C.Structure):
    _fields_ = [("w", C.c_void_p * PC_STEP_CONVS), ("bn", PcBn * PC_STEP_CONVS), ("wt", C.c_void_p * 2), ("bt", C.c_void_p * 2),
                ("dw", C.c_void_p * PC_STEP_CONVS), ("db", C.c_void_p * PC_STEP_CONVS), ("dwt", C.c_void_p * 2), ("dbt", C.c_void_p * 2),
                ("chan", C.c_int32 * 4), ("cin", C.c_int32), ("feat_c0", C.c_int32)]


class PcStepNet(C.Structure):
    _fields_ = [("s", PcStepStream * 2), ("fusion_w", C.c_void_p), ("fusion_b", C.c_void_p)]


class PcStepPlan(C.Structure):
    _fields_ = [("unet", PcStepNet), ("extractor", PcStepNet), ("head_w", C.c_void_p * 8), ("head_dw", C.c_void_p * 8),
                ("flat_p", C.c_void_p), ("flat_g", C.c_void_p), ("adam_m", C.c_void_p), ("adam_v", C.c_void_p),
                ("n", C.c_int32), ("n_decay", C.c_int32), ("n_head", C.c_int32), ("occupancymodel", C.c_int32),
                ("hyper_dev", C.c_void_p), ("step_dev", C.c_void_p), ("norm_dev", C.c_void_p), ("stats_dev", C.c_void_p),
                ("loss_dev", C.c_void_p), ("g_scale_const_dev", C.c_void_p), ("groups", PcAdamGroups),
                ("weight_decay", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("max_norm", C.c_float),
                ("scale_regularization", C.c_float), ("lam_weak", C.c_float), ("lam4", C.c_float * 4), ("extractor_pad", C.c_int32),
                ("band", C.c_int32 * 8), ("mean", C.c_float * 8), ("stdv", C.c_float * 8)]


class PcStepIo(C.Structure):
    _fields_ = [("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("data_kind", C.c_int32), ("data", C.c_void_p), ("data2", C.c_void_p),
                ("craw", C.c_int32), ("dp", C.c_int32), ("admin_mask", C.c_void_p), ("census_idx", C.c_void_p), ("y", C.c_void_p),
                ("sel", C.c_void_p), ("encoder_no_grad", C.c_int32), ("unet_no_grad", C.c_int32), ("inv_B", C.c_float), ("_pad", C.c_int32),
                ("sel_host", C.c_void_p), ("arena", C.c_void_p), ("arena_bytes", C.c_int64), ("arena_needed", C.c_int64), ("off_popcount", C.c_int64),
                ("off_popdense", C.c_int64), ("off_scale", C.c_int64), ("off_mask", C.c_int64), ("off_building", C.c_int64),
                ("launches", C.c_int32), ("_pad2", C.c_int32)]


class PopcornHipError(RuntimeError):
    pass


_lib = None


def lib():
    """Load the library (once).  Raises PopcornHipError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PopcornHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C popcorn_amd/csrc`).  popcorn_amd has no CPU fallback.")
        cand = C.CDLL(LIB_PATH)
        # the .so is a build artefact (git-ignored): refuse a stale one instead of handing it descriptors of another layout
        ver = cand.pc_abi_version() if hasattr(cand, "pc_abi_version") else -1
        sizes = [cand.pc_sizeof(i) for i in range(8)] if hasattr(cand, "pc_sizeof") else []
        want = [C.sizeof(t) for t in (PcSrc, PcDst, PcBn, PcConvFwdDesc, PcAdamGroups, PcLevel2FwdDesc, PcStepPlan, PcStepIo)]
        if ver != PC_ABI_VERSION or sizes != want:
            raise PopcornHipError(f"{LIB_PATH} is stale: ABI version {ver} (binding: {PC_ABI_VERSION}), struct sizes {sizes} "
                                  f"(binding: {want}); rebuild it with `make -C popcorn_amd/csrc`")
        _lib = cand
        _lib.pc_error_string.restype = C.c_char_p
        _lib.pc_step_create.restype = C.c_void_p
        _lib.pc_step_create.argtypes = [C.POINTER(PcStepPlan)]
        _lib.pc_step_destroy.restype = None
        _lib.pc_step_destroy.argtypes = [C.c_void_p]
        _lib.pc_train_step.argtypes = [C.c_void_p, C.POINTER(PcStepIo), C.c_int, C.c_void_p]
        for name in ("pc_conv3x3_wgrad_ws_bytes", "pc_convt2x2_wgrad_ws_bytes", "pc_head_ws_bytes",
                     "pc_compact_ws_bytes", "pc_unet_ws_bytes", "pc_level2_bwd_ws_bytes", "pc_conv3x3_up_ws_bytes", "pc_conv3x3_up_bwd_ws_bytes"):
            if hasattr(_lib, name):
                getattr(_lib, name).restype = C.c_int64
    return _lib


PC_PREC_FP32, PC_PREC_BF16 = 0, 1
PC_F32_T, PC_BF16_T = 0, 1          # enum pc_dtype
PRECISIONS = {"fp32": PC_PREC_FP32, "bf16": PC_PREC_BF16}


def act_dtype():
    """Container type of activation / activation-gradient tensors in the current arithmetic mode."""
    return torch.bfloat16 if lib().pc_get_precision() == PC_PREC_BF16 else torch.float32


_PAD_ROWS = [False]


class padded_rows:
    """``with padded_rows():`` -- planar fp32 activations whose width is not a multiple of 4 are allocated with their rows padded
    to one (a (B, C, H, W) view of a (B, C, H, W4) buffer; the pad columns are never read as data).  Every row then starts on a
    16-byte boundary and the conv kernels keep their aligned staged loaders and vector stores -- e.g. the building extractor's
    1038- and 519-pixel-wide levels of a 2048 x 2048 inference window (popcorn.py:231-258 pads it by 14), which otherwise take the
    per-element loader (3.4x slower on those launches).  Used by the inference path (eval.py); training tiles are unaffected."""

    def __init__(self, on=True):
        self.on = on

    def __enter__(self):
        self.prev = _PAD_ROWS[0]
        _PAD_ROWS[0] = self.on
        return self

    def __exit__(self, *exc):
        _PAD_ROWS[0] = self.prev
        return False


def empty_act(B, C_, H, W, device, zero=False):
    """Activation / activation-gradient tensor (B, C, H, W) in the layout of the current arithmetic mode: planar fp32, or
    channels-last bf16 (one aligned 16-byte slot per pixel and 8-channel group; include/popcorn_hip.h)."""
    mk = torch.zeros if zero else torch.empty
    if lib().pc_get_precision() == PC_PREC_BF16:
        t = torch.empty(B, C_, H, W, device=device, dtype=torch.bfloat16, memory_format=torch.channels_last)
        return t.zero_() if zero else t          # (torch.zeros takes no memory_format)
    if _PAD_ROWS[0] and W % 4:
        return mk(B, C_, H, (W + 3) // 4 * 4, device=device, dtype=torch.float32)[..., :W]
    return mk(B, C_, H, W, device=device, dtype=torch.float32)


def as_act(t):
    """Copy of a (B, C, H, W) tensor in the container type and layout of the current arithmetic mode (tests, tools)."""
    if lib().pc_get_precision() == PC_PREC_BF16:
        return t.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    return t.float().contiguous()


class precision:
    """``with precision("bf16"):`` -- the arithmetic mode of every call enqueued inside the block (pc_set_precision; the
    mode is read at enqueue time, so a HIP graph captured inside the block keeps it).  Restores the previous mode."""

    def __init__(self, mode):
        self.mode = PRECISIONS[mode] if isinstance(mode, str) else int(mode)

    def __enter__(self):
        self.prev = lib().pc_set_precision(self.mode)
        if self.prev < 0:
            raise PopcornHipError(f"pc_set_precision({self.mode}) failed")
        return self

    def __exit__(self, *a):
        lib().pc_set_precision(self.prev)
        return False


_SYNC_DEBUG = os.environ.get("POPCORN_SYNC_DEBUG") == "1"     # debugging aid: name every enqueued call and drain the device after it


def check(code: int, what: str = ""):
    if _SYNC_DEBUG and code == 0:
        import sys
        print(f"[popcorn] {what} enqueued", file=sys.stderr, flush=True)
        torch.cuda.synchronize()
    if code != 0:
        msg = lib().pc_error_string(int(code)).decode()
        raise PopcornHipError(f"{what}: libpopcorn_hip error {code}: {msg}")


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise PopcornHipError("popcorn_amd ops run on a HIP device only (tensor is on %s); there is no CPU path"
                                  % t.device)


_STREAM = [None]


class stream_scope:
    """``with stream_scope():`` -- resolve torch's current stream ONCE for every launch enqueued inside the block.  An eager train step
    on a small census region is bound by the host (40+ launches, each of which asked torch for the current stream: ~8 us a call, a
    quarter of the step's enqueue time, tools/host_time_eager.py); the stream cannot change inside the blocks that use this (one
    section of a step, on the stream it was entered on).  Re-entrant; the outermost scope decides."""

    def __enter__(self):
        self.owner = _STREAM[0] is None
        if self.owner:
            _STREAM[0] = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        return self

    def __exit__(self, *exc):
        if self.owner:
            _STREAM[0] = None
        return False


def stream_ptr() -> C.c_void_p:
    return _STREAM[0] if _STREAM[0] is not None else C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


_CHMAP0 = (0, 1, 2, 3)


def src(t: torch.Tensor, C_=None, mode=PC_SRC_DIRECT, oy=0, ox=0, chmap=_CHMAP0) -> PcSrc:
    """Descriptor of a (B, C, H, W) fp32 or bf16 tensor, planar (unit x stride) or channels-last (unit channel stride);
    strides in elements."""
    st, sh, dt = t.stride(), t.shape, t.dtype
    assert dt in (torch.float32, torch.bfloat16) and len(sh) == 4 and (st[3] == 1 or st[1] == 1)
    s = PcSrc()
    s.xstride = st[3]
    s.dtype = PC_BF16_T if dt == torch.bfloat16 else PC_F32_T
    s.ptr = t.data_ptr()
    s.C = sh[1] if C_ is None else C_
    s.H, s.W = sh[2], sh[3]
    s.bstride, s.cstride, s.rstride = st[0], st[1], st[2]
    s.mode, s.oy, s.ox = mode, oy, ox
    if chmap != _CHMAP0:
        s.chmap[:] = list(chmap)
    else:
        s.chmap[1], s.chmap[2], s.chmap[3] = 1, 2, 3
    return s


def dst(t: torch.Tensor) -> PcDst:
    st, dt = t.stride(), t.dtype
    assert dt in (torch.float32, torch.bfloat16) and len(st) == 4 and (st[3] == 1 or st[1] == 1)
    d = PcDst()
    d.xstride = st[3]
    d.dtype = PC_BF16_T if dt == torch.bfloat16 else PC_F32_T
    d.ptr = t.data_ptr()
    d.bstride, d.cstride, d.rstride = st[0], st[1], st[2]
    return d


def bn(conv_bias=None, gamma=None, beta=None, mean=None, var=None, eps=1e-5) -> PcBn:
    b = PcBn()
    b.conv_bias = 0 if conv_bias is None else conv_bias.data_ptr()
    b.gamma = 0 if gamma is None else gamma.data_ptr()
    b.beta = 0 if beta is None else beta.data_ptr()
    b.mean = 0 if mean is None else mean.data_ptr()
    b.var = 0 if var is None else var.data_ptr()
    b.eps = eps
    b._keep = (conv_bias, gamma, beta, mean, var)      # the descriptor holds raw pointers: keep the tensors alive with it
    return b
