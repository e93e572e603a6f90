"""POPCORN on MI355X: drop-in for the reference's ``model.popcorn.POPCORN`` (model/popcorn.py:13-377).

Same constructor, same ``forward(inputs, train, padding, return_features, encoder_no_grad, unet_no_grad, sparse)``
signature, same output dict, same state-dict keys, real ``nn.Parameter``s with ``.grad`` -- but every op on the path
runs in the hand-written HIP kernels of libpopcorn_hip.so (popcorn_amd/engine.py).  The module refuses to run on CPU
tensors: there is no stock-PyTorch fallback.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _lib as L
from .. import engine as E
from .. import ops
from . import networks

STAGE1_FEATS = 8           # utils/constants.py:169


def _burn_discriminator_rng():
    """The reference constructs (and then discards) an ``ownDiscriminator(8, 2)`` inside every DualStreamUNet
    (networks.py:44,179,49-66).  Its default initialisation consumes the global RNG *before* the head is built, so a
    given ``torch.manual_seed`` only reproduces the reference's head weights if the same draws are made here."""
    for cin, cout in ((8, 64), (64, 128), (128, 256), (256, 512)):
        nn.Conv2d(cin, cout, kernel_size=3, stride=2, padding=1)
    for cin, cout in ((512, 256), (256, 128), (128, 64), (64, 2)):
        nn.ConvTranspose2d(cin, cout, kernel_size=3, stride=2, padding=1, output_padding=1)


def _new_dualstream(device="cpu"):
    net = networks.DualStreamUNet()
    _burn_discriminator_rng()
    from safetensors.torch import load_file
    net.load_state_dict(load_file(networks.CHECKPOINT), strict=False)
    return net.to(device)


def pad_geometry(H, W, force):
    """add_padding (popcorn.py:231-258) as numbers: (top, bottom, left, right)."""
    if force:
        return 14, 14, 14, 14
    pt = pb = pl = pr = 0
    if H % 32 != 0:
        pt = (64 - H % 64) // 2
        pb = (64 - H % 64) - pt
    if W % 32 != 0:
        pl = (64 - W % 64) // 2
        pr = (64 - W % 64) - pl
    return pt, pb, pl, pr


class POPCORN(nn.Module):
    def __init__(self, input_channels, feature_extractor="DDA", occupancymodel=False, pretrained=False, biasinit=0.75,
                 sentinelbuildings=False):
        super().__init__()
        self.occupancymodel = occupancymodel
        self.sentinelbuildings = sentinelbuildings
        self.feature_extractor = feature_extractor
        self.p = 14
        self.p2d = (self.p,) * 4
        self.parent = None
        self.S1, self.S2 = True, True
        if input_channels == 0:
            self.S1, self.S2 = False, False
        elif input_channels == 2:
            self.S1, self.S2 = True, False
        elif input_channels == 4:
            self.S1, self.S2 = False, True
        if not (self.S1 or self.S2):
            raise NotImplementedError("input_channels=0 (no Sentinel modality) has no features to feed the head; the "
                                      "reference fails on it too (torch.cat of an empty list, networks.py:210)")
        self._streams = tuple(n for n, on in (("sar_stream", self.S1), ("optical_stream", self.S2)) if on)

        self.unetmodel = _new_dualstream()                                   # popcorn.py:57
        if not pretrained:                                                   # popcorn.py:59-66
            for m in self.unetmodel.modules():
                if isinstance(m, nn.Conv2d):
                    nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                elif isinstance(m, nn.BatchNorm2d):
                    nn.init.constant_(m.weight, 1)
                    nn.init.constant_(m.bias, 0)
        head_input_dim = self.S1 * STAGE1_FEATS + self.S2 * STAGE1_FEATS
        self.unetmodel.num_params = sum(p.numel() for p in self.unetmodel.parameters() if p.requires_grad)
        h = 64
        self.head = nn.Sequential(                                           # popcorn.py:80-85
            nn.Conv2d(head_input_dim, h, kernel_size=1, padding=0), nn.ReLU(inplace=True),
            nn.Conv2d(h, h, kernel_size=1, padding=0), nn.ReLU(inplace=True),
            nn.Conv2d(h, h, kernel_size=1, padding=0), nn.ReLU(inplace=True),
            nn.Conv2d(h, 2, kernel_size=1, padding=0))
        self.head[-1].bias.data = biasinit * torch.ones(2)                   # popcorn.py:88
        self.num_params = sum(p.numel() for p in self.head.parameters() if p.requires_grad) + self.unetmodel.num_params
        self.building_extractor = _new_dualstream()                          # popcorn.py:96
        self._engines = None
        self._w0_pad = None
        self.precision = "fp32"          # "bf16": PC_PREC_BF16 mixed precision (include/popcorn_hip.h; BASELINE config 4)

    def set_precision(self, mode):
        """Arithmetic mode of every kernel this module enqueues: "fp32" (the reference's arithmetic) or "bf16" (bf16 MFMA
        operands, fp32 accumulation, fp32 master weights; rounding points in include/popcorn_hip.h).  Build-specific: the
        reference constructor has no such switch."""
        if mode not in L.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(L.PRECISIONS)}")
        self.precision = mode
        return self

    # ------------------------------------------------------------------------------------------- engine plumbing
    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._engines = None           # .cuda()/.to() re-create tensors: drop cached device pointers
        return r

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._engines = None
        return r

    def invalidate_cache(self):
        self._engines = None

    @staticmethod
    def _tensor_table(net):
        t = dict(net.named_parameters())
        t.update(dict(net.named_buffers()))
        return t

    def engines(self):
        if self._engines is None:
            # a 2-channel (S1) / 4-channel (S2) input: the stream's channels are picked straight from it
            chmaps = None if (self.S1 and self.S2) else {"sar_stream": (0, 1, 0, 0), "optical_stream": (2, 1, 0, 3)}
            self._engines = (E.UNetEngine(self._tensor_table(self.unetmodel), self._streams, chmaps),
                             E.UNetEngine(self._tensor_table(self.building_extractor), self._streams, chmaps))
        return self._engines

    @property
    def feat_offset(self):
        """first feature channel of the (only) active stream inside the 16-channel feature map"""
        return 0 if self.S1 else 8

    def head_tensors(self):
        """[w0,b0,w2,b2,w4,b4,w6,b6] as the head kernels expect them (w0: 64 x 16).  Single-modality models have a
        64 x 8 first layer (popcorn.py:68-69): it is embedded at the active stream's feature channels of a zero-padded
        64 x 16 image; the other 8 feature channels are identically zero."""
        t = [getattr(self.head[i], n) for i in (0, 2, 4, 6) for n in ("weight", "bias")]
        if self.S1 and self.S2:
            return t
        w0 = t[0]
        if self._w0_pad is None or self._w0_pad.device != w0.device:
            self._w0_pad = torch.zeros(64, 16, 1, 1, device=w0.device, dtype=torch.float32)
        with torch.no_grad():
            self._w0_pad[:, self.feat_offset:self.feat_offset + 8].copy_(w0)
        t[0] = self._w0_pad
        return t

    def head_grad_targets(self, hgrads):
        """Gradient buffers to hand to the head-backward kernel for parameter gradients ``hgrads`` + a fix-up to call
        afterwards (single modality: the kernel fills a 64 x 16 image whose active half is copied into hgrads[0])."""
        if self.S1 and self.S2:
            return hgrads, (lambda: None)
        pad = torch.empty(64, 16, 1, 1, device=hgrads[0].device, dtype=torch.float32)
        k = [pad] + list(hgrads[1:])
        f0 = self.feat_offset
        return k, (lambda: hgrads[0].copy_(pad[:, f0:f0 + 8]))

    def trainable(self):
        """(names, parameters) that receive gradients, in a fixed order: 24 U-Net tensors per active stream + 8 head
        tensors (56 for the S1+S2 model)."""
        names = E.trainable_names("unetmodel.", self._streams) + [f"head.{i}.{n}" for i in (0, 2, 4, 6) for n in ("weight", "bias")]
        table = dict(self.named_parameters())
        return names, [table[n] for n in names]

    # ------------------------------------------------------------------------------------------------- pieces
    def add_padding(self, data, force=True):
        """popcorn.py:231-258: (padded tensor, (px1, px2, py1, py2)) with None for an unpadded dimension.  The model path
        never materialises this (the padding is fused into the first conv's loader); the method exists for callers of the
        reference API."""
        L.require_device(data)
        pt, pb, pl, pr = pad_geometry(data.shape[2], data.shape[3], force)
        out = ops.reflect_pad(data.float(), pt, pb, pl, pr) if (pt or pb or pl or pr) else data
        if force:
            return out, (pt, pb, pl, pr)
        return out, (pt if (pt or pb) else None, pb if (pt or pb) else None, pl if (pl or pr) else None, pr if (pl or pr) else None)

    def revert_padding(self, data, padding):
        """popcorn.py:261-276 (a view: no copy)."""
        px1, px2, py1, py2 = padding
        if px1 is not None or px2 is not None:
            data = data[:, :, px1:data.shape[2] - px2, :]
        if py1 is not None or py2 is not None:
            data = data[:, :, :, py1:data.shape[3] - py2]
        return data

    def create_building_score(self, inputs):
        """popcorn.py:279-322."""
        X = inputs["input"]
        L.require_device(X)
        self.building_extractor.eval()
        self.unetmodel.freeze_bn_layers()
        with torch.no_grad(), L.precision(self.precision):
            return self.engines()[1].building_score(X.contiguous(), self.p)

    def get_sparsity_mask(self, inputs, sparse_unet=False):
        """popcorn.py:361-377 (non-sparse_unet branch).  The two ``multinomial`` draws come from the CPU global
        generator exactly as in the reference (so a seeded run selects the same grid); the mask itself is built on
        the device.  Returns (bool mask (B,H,W), None)."""
        if sparse_unet:
            # popcorn.py:336-359: threshold 0.001, a 250 x 250 grid, per-sample undersampling ratio
            admin = inputs["admin_mask"]
            B, H, W = admin.shape
            xi = torch.ones(H).multinomial(num_samples=min(250, H), replacement=False)
            yi = torch.ones(W).multinomial(num_samples=min(250, W), replacement=False)
            sel = torch.zeros(H + W, dtype=torch.uint8)
            sel[xi] = 1
            sel[H + yi] = 1
            sel = sel.to(admin.device, non_blocking=True)
            mask, ratio = ops.sparsity_mask_unet(inputs["building_counts"].contiguous(), admin.contiguous().float(),
                                                 inputs["census_idx"].contiguous(), sel[:H], sel[H:], 0.001)
            return mask.bool(), ratio
        mask, _ = self._sparsity_mask_u8(inputs)
        return mask.bool(), None

    def _sparsity_mask_u8(self, inputs):
        admin = inputs["admin_mask"]
        B, H, W = admin.shape
        sub = 60
        xi = torch.ones(H).multinomial(num_samples=min(sub, H), replacement=False)
        yi = torch.ones(W).multinomial(num_samples=min(sub, W), replacement=False)
        sel = torch.zeros(H + W, dtype=torch.uint8)
        sel[xi] = 1
        sel[H + yi] = 1
        sel = sel.to(admin.device, non_blocking=True)
        bc = inputs["building_counts"]
        return ops.sparsity_mask(bc.contiguous(), admin.contiguous().float(), inputs["census_idx"].contiguous(),
                                 sel[:H], sel[H:], self.occupancymodel)

    # ------------------------------------------------------------------------------------------------ forward
    def forward(self, inputs, train=False, padding=True, return_features=True, encoder_no_grad=False,
                unet_no_grad=False, sparse=False):
        X = inputs["input"]
        L.require_device(X)
        if X.dim() != 4:
            raise ValueError("Input tensor must have shape (batch_size, channels, height, width)")
        X = X.contiguous().float()
        B, _, H, W = X.shape
        if "building_counts" not in inputs.keys() or self.sentinelbuildings:
            inputs["building_counts"] = self.create_building_score(inputs)
        building = inputs["building_counts"].contiguous()
        mask = None
        if sparse:
            mask, _ = self._sparsity_mask_u8(inputs)
        self.unetmodel.freeze_bn_layers()
        pt, pb, pl, pr = pad_geometry(H, W, padding)
        geom = (pt, pl, H + pt + pb, W + pl + pr)
        admin = inputs["admin_mask"].contiguous().float() if "admin_mask" in inputs.keys() else None
        census = inputs["census_idx"].contiguous() if admin is not None else None
        if not self.occupancymodel:
            # popcorn.py:179-181: popdensemap = relu(out), no building product
            building = torch.ones_like(building)

        names, params = self.trainable()
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        if unet_no_grad:
            self.unetmodel.eval()
        with L.precision(self.precision):
            if need_grad:
                popcount, popdense, scale = _PopcornFn.apply(self, X, building, mask, admin, census, geom, sparse,
                                                             encoder_no_grad, unet_no_grad, *params)
            else:
                popcount, popdense, scale = _forward_nograd(self, X, building, mask, admin, census, geom, sparse)
        aux = {"scale": scale if self.occupancymodel else None}
        return {"popcount": popcount, "popdensemap": popdense, **aux}


def _forward_core(model, X, building, mask, admin, census, geom, sparse, save):
    pt, pl, Hp, Wp = geom
    B, _, H, W = X.shape
    eng = model.engines()[0]
    feats, saved = eng.forward(X, pt, pl, Hp, Wp, save=save)
    ht = model.head_tensors()
    scale_map, popdense, popcount = ops.head_fwd(feats, pt, pl, H, W, ht, building, mask=mask, admin_mask=admin,
                                                 census_idx=census)
    if sparse:
        buf, cnt = ops.compact_masked(scale_map, mask)
        scale = buf[: int(cnt.item())]          # Nsel sizes the returned tensor (the reference's boolean index syncs too)
    else:
        scale = scale_map
    return feats, saved, scale, popdense, popcount


def _forward_nograd(model, X, building, mask, admin, census, geom, sparse):
    with torch.no_grad():
        _, _, scale, popdense, popcount = _forward_core(model, X, building, mask, admin, census, geom, sparse, save=False)
    return popcount, popdense, scale


class _PopcornFn(torch.autograd.Function):
    """One autograd node for unetmodel + head + occupancy product + census reduction."""

    @staticmethod
    def forward(ctx, model, X, building, mask, admin, census, geom, sparse, encoder_no_grad, unet_no_grad, *params):
        feats, saved, scale, popdense, popcount = _forward_core(model, X, building, mask, admin, census, geom, sparse,
                                                                save=not unet_no_grad)
        ctx.model, ctx.saved, ctx.feats = model, saved, feats
        ctx.aux = (X, building, mask, admin, census, geom, sparse, encoder_no_grad, unet_no_grad)
        return popcount, popdense, scale

    @staticmethod
    def backward(ctx, g_popcount, g_popdense, g_scale):
        model = ctx.model
        X, building, mask, admin, census, geom, sparse, encoder_no_grad, unet_no_grad = ctx.aux
        pt, pl, Hp, Wp = geom
        B, _, H, W = X.shape
        names, params = model.trainable()
        eng = model.engines()[0]
        g_scale_map = None
        if g_scale is not None:
            if sparse:
                g_scale_map = ops.scatter_masked(g_scale.contiguous().float(), mask)       # (B, H, W): autograd of scale[mask]
            else:
                g_scale_map = g_scale.contiguous().float()
        grads = {n: torch.empty_like(p) for n, p in zip(names, params)}
        hgrads = [grads[n] for n in names[-8:]]
        khgrads, fix = model.head_grad_targets(hgrads)
        with L.precision(model.precision):
            _, G = ops.head_bwd(ctx.feats, pt, pl, H, W, model.head_tensors(), building, mask=mask, admin_mask=admin,
                                census_idx=census,
                                g_popcount=None if g_popcount is None else g_popcount.contiguous().float(),
                                g_popdense=None if g_popdense is None else g_popdense.contiguous().float(),
                                g_scale_map=g_scale_map, grads=khgrads,
                                feat_bn=None if unet_no_grad else eng.feat_bn())
            fix()
            if not unet_no_grad:
                eng.backward(ctx.saved, G, grads, accumulate=False, encoder_no_grad=encoder_no_grad, prefix="unetmodel.")
        n_unet = len(names) - 8
        if unet_no_grad:
            out = [None] * n_unet + hgrads
        else:
            out = []
            for n in names[:n_unet]:
                is_enc = any(("." + E.CONVS[t][0] + ".") in n for t in E.ENCODER)
                out.append(None if (encoder_no_grad and is_enc) else grads[n])
            out += hgrads
        ctx.saved = None
        return (None,) * 10 + tuple(out)
