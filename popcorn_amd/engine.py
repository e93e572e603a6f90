"""HIP execution engine for the dual-stream U-Net (forward, building score, backward).

Orchestrates the kernels of libpopcorn_hip.so for the computation the reference expresses as
``DualStreamUNet.forward`` / ``UNet.forward`` / ``DoubleConv`` / ``Down`` / ``Up``
(model/DDA_model/utils/networks.py:121-151,192-237,253-320) and its autograd backward.  Fusions relative to the
reference's op list:

  * reflect padding + channel reorder (popcorn.py:231-258,130-134) -> loader of the first conv
  * Conv2d + BatchNorm2d(eval) + ReLU -> one kernel (BN folded in the epilogue; BN is frozen, networks.py:184-189)
  * MaxPool2d(2) -> loader of the next conv;  its backward -> epilogue of that conv's data-gradient
  * torch.cat([skip, up]) + Up's zero F.pad -> two-source loader (no concat buffer)
  * ReLU/BN backward -> epilogue of the kernel that produces the gradient (no elementwise passes)
  * the two streams write straight into the 16-channel feature map (no final cat)

All launches go to the current torch stream, so a whole step can be captured into one HIP graph
(``torch.cuda.graph``) and replayed.  There is no non-HIP path.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from . import ops

BN_EPS = 1e-5
STREAMS = (("sar_stream", (4, 5, 0, 0), 2, 0), ("optical_stream", (2, 1, 0, 3), 4, 8))   # name, chmap, Cin, feat ch0

# (tag, conv key, bn key) per stream, relative to the stream prefix
CONVS = {
    "inc1": ("inc.conv.conv.0", "inc.conv.conv.1"),
    "inc2": ("inc.conv.conv.3", "inc.conv.conv.4"),
    "d1a": ("down_seq.down1.mpconv.1.conv.0", "down_seq.down1.mpconv.1.conv.1"),
    "d1b": ("down_seq.down1.mpconv.1.conv.3", "down_seq.down1.mpconv.1.conv.4"),
    "d2a": ("down_seq.down2.mpconv.1.conv.0", "down_seq.down2.mpconv.1.conv.1"),
    "d2b": ("down_seq.down2.mpconv.1.conv.3", "down_seq.down2.mpconv.1.conv.4"),
    "up2a": ("up_seq.up2.conv.conv.0", "up_seq.up2.conv.conv.1"),
    "up2b": ("up_seq.up2.conv.conv.3", "up_seq.up2.conv.conv.4"),
    "up1a": ("up_seq.up1.conv.conv.0", "up_seq.up1.conv.conv.1"),
    "up1b": ("up_seq.up1.conv.conv.3", "up_seq.up1.conv.conv.4"),
}
CONVTS = {"up2t": "up_seq.up2.up", "up1t": "up_seq.up1.up"}
ENCODER = ("inc1", "inc2", "d1a", "d1b", "d2a", "d2b")


def trainable_names(prefix="unetmodel."):
    """The 48 U-Net tensors that receive a gradient in the reference train step (conv / convT weights + biases of
    both streams; BN affine frozen, out-convs unused).  SURVEY.md section 8a row 18."""
    names = []
    for s, _, _, _ in STREAMS:
        for tag in ("inc1", "inc2", "d1a", "d1b", "d2a", "d2b"):
            names += [f"{prefix}{s}.{CONVS[tag][0]}.weight", f"{prefix}{s}.{CONVS[tag][0]}.bias"]
        names += [f"{prefix}{s}.{CONVTS['up2t']}.weight", f"{prefix}{s}.{CONVTS['up2t']}.bias"]
        for tag in ("up2a", "up2b"):
            names += [f"{prefix}{s}.{CONVS[tag][0]}.weight", f"{prefix}{s}.{CONVS[tag][0]}.bias"]
        names += [f"{prefix}{s}.{CONVTS['up1t']}.weight", f"{prefix}{s}.{CONVTS['up1t']}.bias"]
        for tag in ("up1a", "up1b"):
            names += [f"{prefix}{s}.{CONVS[tag][0]}.weight", f"{prefix}{s}.{CONVS[tag][0]}.bias"]
    return names


class _Layer:
    __slots__ = ("w", "b", "bn", "bn_nobias", "wname", "bname", "_keep")

    def __init__(self, T, conv_key, bn_key):
        self.wname, self.bname = conv_key + ".weight", conv_key + ".bias"
        self.w, self.b = T[self.wname], T[self.bname]
        if bn_key is not None:
            g, be, m, v = (T[bn_key + "." + n] for n in ("weight", "bias", "running_mean", "running_var"))
            self.bn = L.bn(self.b, g, be, m, v, BN_EPS)
            self.bn_nobias = L.bn(None, g, be, m, v, BN_EPS)
            self._keep = (g, be, m, v)
        else:
            self.bn = self.bn_nobias = None
            self._keep = ()


class UNetEngine:
    """Executes one DualStreamUNet on the HIP kernels.  ``tensors``: name -> tensor, names relative to the
    DualStreamUNet module (e.g. 'sar_stream.inc.conv.conv.0.weight').  Tensors are referenced, not copied."""

    def __init__(self, tensors):
        self.layers = {}
        for s, _, _, _ in STREAMS:
            for tag, (ck, bk) in CONVS.items():
                self.layers[(s, tag)] = _Layer(tensors, f"{s}.{ck}", f"{s}.{bk}")
            for tag, ck in CONVTS.items():
                self.layers[(s, tag)] = _Layer(tensors, f"{s}.{ck}", None)
        self.fusion_w = tensors.get("fusion_out_conv.conv.weight")
        self.fusion_b = tensors.get("fusion_out_conv.conv.bias")
        dev = self.layers[("sar_stream", "inc1")].w.device
        if dev.type != "cuda":
            raise L.PopcornHipError(f"popcorn_amd engine needs parameters on a HIP device, got {dev}; there is no CPU path")
        self.device = dev

    # ------------------------------------------------------------------------------------------------ forward
    def _conv(self, lay, a, out=None, **kw):
        bn = lay.bn
        return ops.conv3x3_raw(a, lay.w, bn, out=out, **kw)

    def forward(self, X, pad_top, pad_left, Hp, Wp, save=False, feats=None):
        """X: (B,6,H,W) in the dataset's channel order [R,G,B,NIR,VV,VH]; the conv domain is the reflect-padded
        (Hp,Wp) image.  Returns (features (B,16,Hp,Wp), saved activations or None)."""
        L.require_device(X)
        B = X.shape[0]
        if pad_top >= X.shape[2] or pad_left >= X.shape[3] or Hp - X.shape[2] - pad_top >= X.shape[2] \
                or Wp - X.shape[3] - pad_left >= X.shape[3]:
            raise ValueError("reflect padding must be smaller than the input (same restriction as F.pad reflect)")
        if Hp < 4 or Wp < 4:
            raise ValueError("input too small for two 2x2 poolings")
        dev = X.device
        H1, W1, = Hp // 2, Wp // 2
        H2, W2 = H1 // 2, W1 // 2
        if feats is None:
            feats = torch.empty(B, 16, Hp, Wp, device=dev, dtype=torch.float32)
        saved = {} if save else None
        E = lambda c, h, w: torch.empty(B, c, h, w, device=dev, dtype=torch.float32)  # noqa: E731
        for s, chmap, cin, f0 in STREAMS:
            ly = lambda t: self.layers[(s, t)]  # noqa: E731
            a1 = ops.conv3x3_raw(X, ly("inc1").w, ly("inc1").bn, a_mode=L.PC_SRC_REFLECT, a_pad=(pad_top, pad_left),
                                 chmap=chmap, out_hw=(Hp, Wp), a_channels=cin, out=E(8, Hp, Wp))
            a2 = ops.conv3x3_raw(a1, ly("inc2").w, ly("inc2").bn, out=E(8, Hp, Wp))
            b1 = ops.conv3x3_raw(a2, ly("d1a").w, ly("d1a").bn, a_mode=L.PC_SRC_POOL2, out=E(16, H1, W1))
            b2 = ops.conv3x3_raw(b1, ly("d1b").w, ly("d1b").bn, out=E(16, H1, W1))
            c1 = ops.conv3x3_raw(b2, ly("d2a").w, ly("d2a").bn, a_mode=L.PC_SRC_POOL2, out=E(16, H2, W2))
            c2 = ops.conv3x3_raw(c1, ly("d2b").w, ly("d2b").bn, out=E(16, H2, W2))
            u2 = ops.convt2x2(c2, ly("up2t").w, ly("up2t").b)
            o2 = ((H1 - 2 * H2) // 2, (W1 - 2 * W2) // 2)
            e1 = ops.conv3x3_raw(b2, ly("up2a").w, ly("up2a").bn, b=u2, b_offset=o2, out=E(8, H1, W1))
            e2 = ops.conv3x3_raw(e1, ly("up2b").w, ly("up2b").bn, out=E(8, H1, W1))
            u1 = ops.convt2x2(e2, ly("up1t").w, ly("up1t").b)
            o1 = ((Hp - 2 * H1) // 2, (Wp - 2 * W1) // 2)
            f1 = ops.conv3x3_raw(a2, ly("up1a").w, ly("up1a").bn, b=u1, b_offset=o1, out=E(8, Hp, Wp))
            ops.conv3x3_raw(f1, ly("up1b").w, ly("up1b").bn, out=feats[:, f0:f0 + 8])
            if save:
                saved[s] = dict(a1=a1, a2=a2, b1=b1, b2=b2, c1=c1, c2=c2, u2=u2, e1=e1, e2=e2, u1=u1, f1=f1, o1=o1, o2=o2)
        if save:
            saved["X"] = X
            saved["geom"] = (pad_top, pad_left, Hp, Wp)
            saved["feats"] = feats
        return feats, saved

    def building_score(self, X, pad=14):
        """create_building_score (popcorn.py:279-322): reflect-pad 14, frozen U-Net, fusion_out_conv, sigmoid, crop."""
        B, _, H, W = X.shape
        feats, _ = self.forward(X, pad, pad, H + 2 * pad, W + 2 * pad, save=False)
        return ops.outconv_sigmoid_crop(feats, self.fusion_w, self.fusion_b, H, W, pad, pad)

    def feat_bn(self):
        """BN descriptors of the two layers that produce the feature map (for the head-backward epilogue)."""
        return (self.layers[("sar_stream", "up1b")].bn_nobias, self.layers[("optical_stream", "up1b")].bn_nobias)

    # ----------------------------------------------------------------------------------------------- backward
    def backward(self, saved, G, grads, accumulate=False, encoder_no_grad=False, prefix=""):
        """G: (B,16,Hp,Wp) gradient w.r.t. the conv outputs of the two up1b layers (i.e. already multiplied by
        relu-mask * bn-scale -- the head-backward epilogue does that).  Writes dW/db into ``grads[prefix+name]``
        (= or += per ``accumulate``).  encoder_no_grad: networks.py:124-132 semantics."""
        X = saved["X"]
        pad_top, pad_left, Hp, Wp = saved["geom"]
        B = X.shape[0]
        dev = X.device
        H1, W1 = Hp // 2, Wp // 2
        H2, W2 = H1 // 2, W1 // 2
        E = lambda c, h, w: torch.empty(B, c, h, w, device=dev, dtype=torch.float32)  # noqa: E731
        for s, chmap, cin, f0 in STREAMS:
            A = saved[s]
            ly = lambda t: self.layers[(s, t)]  # noqa: E731

            def wg(tag, a, g, **kw):
                lay = ly(tag)
                ops.conv3x3_wgrad(a, g, lay.w.shape[0], dw=grads[prefix + lay.wname],
                                  db=grads[prefix + lay.bname], accumulate=accumulate, **kw)

            def wgt(tag, x, g):
                lay = ly(tag)
                ops.convt2x2_wgrad(x, g, dw=grads[prefix + lay.wname], db=grads[prefix + lay.bname],
                                   accumulate=accumulate)

            G_f2 = G[:, f0:f0 + 8]
            # up1b
            wg("up1b", A["f1"], G_f2)
            G_f1 = ops.conv3x3_dgrad(G_f2, ly("up1b").w, 0, 8, E(8, Hp, Wp), act=A["f1"], act_bn=ly("up1a").bn_nobias)
            # up1a over cat[a2, pad(u1)]
            wg("up1a", A["a2"], G_f1, b=A["u1"], b_offset=A["o1"])
            if not encoder_no_grad:
                G_a2 = ops.conv3x3_dgrad(G_f1, ly("up1a").w, 0, 8, E(8, Hp, Wp), act=A["a2"], act_bn=ly("inc2").bn_nobias)
            g_u1 = ops.conv3x3_dgrad(G_f1, ly("up1a").w, 8, 8, E(8, Hp, Wp))
            oy, ox = A["o1"]
            g_u1v = g_u1[:, :, oy:oy + 2 * H1, ox:ox + 2 * W1]
            wgt("up1t", A["e2"], g_u1v)
            G_e2 = ops.convt2x2_dgrad(g_u1v, ly("up1t").w, E(8, H1, W1), act=A["e2"], act_bn=ly("up2b").bn_nobias)
            # up2b, up2a
            wg("up2b", A["e1"], G_e2)
            G_e1 = ops.conv3x3_dgrad(G_e2, ly("up2b").w, 0, 8, E(8, H1, W1), act=A["e1"], act_bn=ly("up2a").bn_nobias)
            wg("up2a", A["b2"], G_e1, b=A["u2"], b_offset=A["o2"])
            if not encoder_no_grad:
                G_b2 = ops.conv3x3_dgrad(G_e1, ly("up2a").w, 0, 16, E(16, H1, W1), act=A["b2"], act_bn=ly("d1b").bn_nobias)
            g_u2 = ops.conv3x3_dgrad(G_e1, ly("up2a").w, 16, 16, E(16, H1, W1))
            oy, ox = A["o2"]
            g_u2v = g_u2[:, :, oy:oy + 2 * H2, ox:ox + 2 * W2]
            wgt("up2t", A["c2"], g_u2v)
            if encoder_no_grad:
                continue
            G_c2 = ops.convt2x2_dgrad(g_u2v, ly("up2t").w, E(16, H2, W2), act=A["c2"], act_bn=ly("d2b").bn_nobias)
            # encoder
            wg("d2b", A["c1"], G_c2)
            G_c1 = ops.conv3x3_dgrad(G_c2, ly("d2b").w, 0, 16, E(16, H2, W2), act=A["c1"], act_bn=ly("d2a").bn_nobias)
            wg("d2a", A["b2"], G_c1, a_mode=L.PC_SRC_POOL2)
            ops.conv3x3_dgrad(G_c1, ly("d2a").w, 0, 16, G_b2, act=A["b2"], act_bn=ly("d1b").bn_nobias, pool=True,
                              accumulate=True)
            wg("d1b", A["b1"], G_b2)
            G_b1 = ops.conv3x3_dgrad(G_b2, ly("d1b").w, 0, 16, E(16, H1, W1), act=A["b1"], act_bn=ly("d1a").bn_nobias)
            wg("d1a", A["a2"], G_b1, a_mode=L.PC_SRC_POOL2)
            ops.conv3x3_dgrad(G_b1, ly("d1a").w, 0, 8, G_a2, act=A["a2"], act_bn=ly("inc2").bn_nobias, pool=True,
                              accumulate=True)
            wg("inc2", A["a1"], G_a2)
            G_a1 = ops.conv3x3_dgrad(G_a2, ly("inc2").w, 0, 8, E(8, Hp, Wp), act=A["a1"], act_bn=ly("inc1").bn_nobias)
            wg("inc1", X, G_a1, a_mode=L.PC_SRC_REFLECT, a_pad=(pad_top, pad_left), chmap=chmap, a_channels=cin)
