"""Parameter containers for the DDA dual-stream U-Net.

Mirrors the *module tree* of the reference (model/DDA_model/utils/networks.py:76-330) so that
``state_dict()`` / ``named_parameters()`` produce exactly the reference's key names and shapes
(SURVEY.md section 8b): published ``last_model.pth`` files and the DDA checkpoint load unchanged, and the
optimizer grouping by parameter name (run_train.py:82-85) keeps working.

These modules hold parameters only.  Their arithmetic lives in ``popcorn_amd.engine`` (hand-written HIP
kernels); calling ``forward`` on them raises -- there is deliberately no stock-PyTorch compute path.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import torch
import torch.nn as nn

TOPOLOGY = (8, 16)                 # utils/constants.py:169-173  (stage1feats, stage2feats)
SAR_IN, OPTICAL_IN = 2, 4          # utils/constants.py:176  (VV,VH / B02,B03,B04,B08)
OUT_CHANNELS = 1
CHECKPOINT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "checkpoints",
                          "fusionda_newAug8_16_checkpoint30_lossweight0.5.safetensors")


class _ParamsOnly(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError("popcorn_amd parameter containers have no stock-PyTorch forward; "
                           "use POPCORN.forward (HIP engine)")


class DoubleConv(_ParamsOnly):
    """conv.{0,3}: Conv2d 3x3; conv.{1,4}: BatchNorm2d; conv.{2,5}: ReLU.  networks.py:253-271."""
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True),
                                  nn.Conv2d(cout, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class InConv(_ParamsOnly):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = DoubleConv(cin, cout)


class Down(_ParamsOnly):
    """mpconv.0: MaxPool2d(2); mpconv.1: DoubleConv.  networks.py:284-295."""
    def __init__(self, cin, cout):
        super().__init__()
        self.mpconv = nn.Sequential(nn.MaxPool2d(2), DoubleConv(cin, cout))


class Up(_ParamsOnly):
    """up: ConvTranspose2d(C/2, C/2, 2, stride 2); conv: DoubleConv(C, out).  networks.py:298-320."""
    def __init__(self, cin, cout):
        super().__init__()
        self.up = nn.ConvTranspose2d(cin // 2, cin // 2, 2, stride=2)
        self.conv = DoubleConv(cin, cout)


class OutConv(_ParamsOnly):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 1)


class UNet(_ParamsOnly):
    """inc / outc / down_seq.down{1,2} / up_seq.up{2,1}; registration order follows networks.py:90-119 so the
    state-dict key ORDER matches too."""
    def __init__(self, n_channels, n_classes=OUT_CHANNELS, topology=TOPOLOGY):
        super().__init__()
        first = topology[0]
        self.inc = InConv(n_channels, first)
        self.outc = OutConv(first, n_classes)
        n = len(topology)
        up_topo = [first]
        down = OrderedDict()
        for idx in range(n):
            cin = topology[idx]
            cout = topology[idx + 1] if idx != n - 1 else topology[idx]
            down[f"down{idx + 1}"] = Down(cin, cout)
            up_topo.append(cout)
        self.down_seq = nn.ModuleDict(down)
        ups = OrderedDict()
        for idx in reversed(range(n)):
            x2 = idx - 1 if idx != 0 else idx
            ups[f"up{idx + 1}"] = Up(up_topo[idx] * 2, up_topo[x2])
        self.up_seq = nn.ModuleDict(ups)


class DualStreamUNet(_ParamsOnly):
    """sar_stream / sar_out_conv / optical_stream / optical_out_conv / fusion_out_conv.  networks.py:156-182.
    The reference also builds a discriminator and immediately drops it (networks.py:44,179); it has no
    parameters in any checkpoint, so it is not created here."""
    def __init__(self):
        super().__init__()
        self.sar_stream = UNet(SAR_IN)
        self.sar_in = SAR_IN
        self.sar_out_conv = OutConv(TOPOLOGY[0], OUT_CHANNELS)
        self.optical_stream = UNet(OPTICAL_IN)
        self.optical_in = OPTICAL_IN
        self.optical_out_conv = OutConv(TOPOLOGY[0], OUT_CHANNELS)
        self.fusion_out_conv = OutConv(2 * TOPOLOGY[0], OUT_CHANNELS)
        self.disc = None
        self.patchsize = 512

    def freeze_bn_layers(self):
        """networks.py:184-189: BN always in eval mode with frozen affine."""
        for layer in self.modules():
            if isinstance(layer, nn.BatchNorm2d):
                layer.eval()
                for p in layer.parameters():
                    p.requires_grad = False


def load_checkpoint(path: str | None = None, device="cpu") -> DualStreamUNet:
    """Counterpart of load_checkpoint(epoch=30, cfg=dda_cfg, device) (networks.py:32-46): a DualStreamUNet
    initialised from the DDA 'fusionda_newAug8_16' checkpoint, shipped as a safetensors data asset (resolved
    relative to the package, not the cwd).  ``path`` may also point at the reference's original .pt file."""
    net = DualStreamUNet()
    path = path or CHECKPOINT
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(path)
    else:
        sd = torch.load(path, map_location="cpu", weights_only=False)["network"]
    net.load_state_dict(sd, strict=False)          # strict=False like the reference (no disc.* keys)
    return net.to(device)
