"""Training-time augmentations with the reference's names, parameters and RNG consumption (utils/transform.py:12-276,
wired up in run_train.py:386-402), restated without torchvision:

* ``RandomBrightness`` / ``RandomGamma`` act on raw Sentinel-2 digital numbers (0..10000) before normalisation.  The
  torchvision functionals they call are absent from this image; their published float-image definitions are
  ``adjust_brightness(x, f) = clamp(f * x, 0, 1)`` (a blend with a black image) and
  ``adjust_gamma(x, g, gain=1) = clamp(gain * x ** g, 0, 1)``.
* the "general" transforms act jointly on ``(input, stacked masks)``.  ``TF.vflip / TF.hflip`` are ``flip`` along
  H / W; ``TF.rotate(x, angle, expand=True)`` for angle in {90, 180, 270} is the exact counter-clockwise quarter-turn
  ``torch.rot90(x, angle // 90, (-2, -1))`` (the ``fill=-1`` of the mask never shows for right angles).

Every draw uses the same generator as the reference -- ``torch.rand(1)`` for the coin flips, Python's ``random`` for the
factors / the angle -- in the same order, so a seeded run consumes both streams identically.  Plain torch ops: they run
on whatever device the tensors live on.
"""
from __future__ import annotations

import random

import torch


class OwnCompose:
    """utils/transform.py:12-22"""

    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


Compose = OwnCompose        # torchvision.transforms.Compose has the same call semantics for these callables


def _split(x):
    return (x, None) if torch.is_tensor(x) else x


class _RandomFlip:
    dim = -2

    def __init__(self, p=0.5, allsame=False):
        self.p = p
        self.allsame = allsame

    def __call__(self, x):
        x, mask = _split(x)
        if self.allsame:                                     # one coin for the whole batch (run_train.py:388-389)
            if torch.rand(1) < self.p:
                x = torch.flip(x, dims=(self.dim,))
                if mask is not None:
                    mask = torch.flip(mask, dims=(self.dim,))
        else:                                                # one coin per sample, in place like the reference
            sel = torch.rand(x.shape[0]) < self.p
            sel = sel.to(x.device)
            x[sel] = torch.flip(x, dims=(self.dim,))[sel]
            if mask is not None:
                mask[sel] = torch.flip(mask, dims=(self.dim,))[sel]
        return x if mask is None else (x, mask)

    def __repr__(self):
        return f"{self.__class__.__name__}(p={self.p}"


class RandomVerticalFlip(_RandomFlip):
    """utils/transform.py:54-95"""
    dim = -2


class RandomHorizontalFlip(_RandomFlip):
    """utils/transform.py:98-139"""
    dim = -1


class RandomRotationTransform:
    """utils/transform.py:142-172: with probability p rotate input and mask by random.choice(angles) (expand=True)."""

    def __init__(self, angles, p=0.5):
        self.angles = list(angles)
        self.p = p

    def __call__(self, x):
        x, mask = _split(x)
        if torch.rand(1) < self.p:
            angle = random.choice(self.angles)
            if angle % 90 != 0:
                raise ValueError("only right-angle rotations are supported (the reference uses 90/180/270)")
            k = (angle // 90) % 4
            x = torch.rot90(x, k, dims=(-2, -1))
            if mask is not None:
                mask = torch.rot90(mask, k, dims=(-2, -1))
        return x if mask is None else (x, mask)


def adjust_brightness(x, factor):
    return (x * factor).clamp(0.0, 1.0)


def adjust_gamma(x, gamma, gain=1.0):
    return (gain * x.pow(gamma)).clamp(0.0, 1.0)


class RandomGamma:
    """utils/transform.py:175-224.  Note the reference's quirk: a 3-channel input gets adjust_brightness(x, gamma)."""

    def __init__(self, gamma_limit=(0.5, 2.0), p=0.5, s2_max=10000):
        self.gamma_limit = gamma_limit
        self.p = p
        self.s2_max = s2_max

    def __call__(self, x):
        if torch.rand(1) < self.p:
            gamma = random.uniform(self.gamma_limit[0], self.gamma_limit[1])
            x = torch.clip(x, min=0) / self.s2_max
            cdim = x.dim() - 3
            if x.shape[cdim] == 3:
                x = adjust_brightness(x, gamma)
            elif x.dim() == 4:
                for i in range(x.shape[1]):                  # channel by channel like the reference (transform.py:220-221):
                    x[:, i:i + 1] = adjust_gamma(x[:, i:i + 1], gamma)      # bit-equal to it on CPU (fixture g10)
            else:
                # 3-D input with C != 3: the reference loops over range(x.shape[1]) but slices dim 0 (transform.py:207-208)
                x = x.clone()
                for i in range(x.shape[1]):
                    x[i:i + 1] = adjust_gamma(x[i:i + 1], gamma)
            x = x * self.s2_max
        return x


class RandomBrightness:
    """utils/transform.py:227-276"""

    def __init__(self, beta_limit=(0.666, 1.5), p=0.5):
        self.beta_limit = beta_limit
        self.p = p
        self.s2_max = 10000

    def __call__(self, x):
        if torch.rand(1) < self.p:
            beta = random.uniform(self.beta_limit[0], self.beta_limit[1])
            x = adjust_brightness(x / self.s2_max, beta) * self.s2_max
        return x


def default_train_transform():
    """The augmentation set of the reference trainer (run_train.py:386-402)."""
    return {
        "general": Compose([RandomVerticalFlip(p=0.5, allsame=True), RandomHorizontalFlip(p=0.5, allsame=True),
                            RandomRotationTransform(angles=[90, 180, 270], p=0.75)]),
        "S2": OwnCompose([RandomBrightness(p=0.9, beta_limit=(0.666, 1.5)), RandomGamma(p=0.9, gamma_limit=(0.6666, 1.5))]),
        "S1": Compose([]),
    }


def draw_fused_params(transform):
    """The parameters of one batch's augmentation, drawn with EXACTLY the generator consumption of applying ``transform`` the way the
    trainer does (utils/utils.py:130-214: the S2 transforms first, then the joint geometric ones) -- ``torch.rand(1)`` per coin, Python's
    ``random`` for the factors / the angle, in the same order -- for ``ops.augment_raw`` (one HIP launch instead of the per-op torch
    launches).  Returns ``{"beta", "gamma", "vflip", "hflip", "rot"}`` (None / False / 0 = not applied) or ``None`` when ``transform`` is
    not the reference trainer's set (per-sample coins, other transforms): the caller then applies the classes themselves."""
    if transform is None:
        return {"beta": None, "gamma": None, "vflip": False, "hflip": False, "rot": 0}
    s2 = transform.get("S2")
    gen = transform.get("general")
    s1 = transform.get("S1")
    s2t = list(getattr(s2, "transforms", [])) if s2 is not None else []
    gt = list(getattr(gen, "transforms", [])) if gen is not None else []
    if s1 is not None and list(getattr(s1, "transforms", [None])):
        return None
    if [type(t) for t in s2t] not in ([], [RandomBrightness, RandomGamma]):
        return None
    if [type(t) for t in gt] not in ([], [RandomVerticalFlip, RandomHorizontalFlip, RandomRotationTransform]):
        return None
    if gt and not (gt[0].allsame and gt[1].allsame and all(a % 90 == 0 for a in gt[2].angles)):
        return None
    if s2t and (s2t[0].s2_max != 10000 or s2t[1].s2_max != 10000):
        return None
    out = {"beta": None, "gamma": None, "vflip": False, "hflip": False, "rot": 0}
    if s2t:
        br, ga = s2t
        if torch.rand(1) < br.p:
            out["beta"] = random.uniform(br.beta_limit[0], br.beta_limit[1])
        if torch.rand(1) < ga.p:
            out["gamma"] = random.uniform(ga.gamma_limit[0], ga.gamma_limit[1])
    if gt:
        vf, hf, ro = gt
        out["vflip"] = bool(torch.rand(1) < vf.p)
        out["hflip"] = bool(torch.rand(1) < hf.p)
        if torch.rand(1) < ro.p:
            out["rot"] = (random.choice(ro.angles) // 90) % 4
    return out
