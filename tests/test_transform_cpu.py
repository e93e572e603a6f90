"""Augmentations + sample preparation (popcorn_amd/utils/transform.py, utils/utils.py) against hand-computed answers.

The reference classes (utils/transform.py:54-276) need torchvision, which this image lacks, so there is no golden
vector: the expected values below restate torchvision's published float-image definitions (adjust_brightness = clamp(f*x),
adjust_gamma = clamp(x**g)) and torch.flip / torch.rot90, and pin the RNG consumption order of the reference
(torch.rand(1) for the coin, random.uniform / random.choice for the value)."""
import random

import torch

from popcorn_amd.data import stats
from popcorn_amd.utils import transform as T
from popcorn_amd.utils.utils import apply_normalize, apply_transformations_and_normalize, default_dataset_stats


def test_brightness_known_answer():
    x = torch.tensor([0.0, 2000.0, 6000.0, 9000.0]).view(1, 4, 1, 1).repeat(1, 1, 2, 2)
    torch.manual_seed(0)
    random.seed(5)
    coin = torch.rand(1)
    beta = random.uniform(0.666, 1.5)
    torch.manual_seed(0)
    random.seed(5)
    y = T.RandomBrightness(p=1.1)(x.clone())
    assert coin < 1.1
    assert torch.allclose(y, (x / 10000 * beta).clamp(0, 1) * 10000)
    assert y.max() <= 10000.0


def test_brightness_not_applied_consumes_only_the_coin():
    x = torch.rand(2, 4, 3, 3) * 10000
    random.seed(1)
    state = random.getstate()
    torch.manual_seed(0)
    y = T.RandomBrightness(p=0.0)(x)
    assert y is x and random.getstate() == state          # no python draw when the coin says no
    torch.manual_seed(0)
    torch.rand(1)
    after = torch.rand(1)
    torch.manual_seed(0)
    T.RandomBrightness(p=0.0)(x)
    assert torch.equal(torch.rand(1), after)               # exactly one torch draw


def test_gamma_known_answer_and_clip():
    x = torch.tensor([-50.0, 0.0, 2500.0, 10000.0, 12000.0]).view(1, 5, 1, 1)
    random.seed(2)
    g = random.uniform(0.6666, 1.5)
    random.seed(2)
    y = T.RandomGamma(p=1.1, gamma_limit=(0.6666, 1.5))(x)
    want = (torch.clip(x, min=0) / 10000).pow(g).clamp(0, 1) * 10000
    assert torch.allclose(y, want)
    assert y[0, 0].item() == 0.0 and y[0, 4].item() == 10000.0


def test_gamma_three_channel_quirk_is_brightness():
    x = torch.full((1, 3, 2, 2), 4000.0)                   # reference: 3 channels -> adjust_brightness(x, gamma)
    random.seed(3)
    g = random.uniform(0.5, 2.0)
    random.seed(3)
    y = T.RandomGamma(p=1.1)(x)
    assert torch.allclose(y, (x / 10000 * g).clamp(0, 1) * 10000)


def test_flips_allsame_and_per_sample():
    x = torch.arange(2 * 1 * 3 * 4.0).view(2, 1, 3, 4)
    m = x.clone() + 100
    y, ym = T.RandomVerticalFlip(p=1.1, allsame=True)((x.clone(), m.clone()))
    assert torch.equal(y, x.flip(-2)) and torch.equal(ym, m.flip(-2))
    y, ym = T.RandomHorizontalFlip(p=1.1, allsame=True)((x.clone(), m.clone()))
    assert torch.equal(y, x.flip(-1)) and torch.equal(ym, m.flip(-1))
    assert torch.equal(T.RandomHorizontalFlip(p=0.0, allsame=True)(x.clone()), x)
    torch.manual_seed(4)
    sel = torch.rand(2) < 0.5
    torch.manual_seed(4)
    y = T.RandomVerticalFlip(p=0.5, allsame=False)(x.clone())
    want = x.clone()
    want[sel] = x.flip(-2)[sel]
    assert torch.equal(y, want)


def test_rotation_is_ccw_quarter_turns_with_expand():
    x = torch.arange(1 * 1 * 2 * 3.0).view(1, 1, 2, 3)
    m = -x
    for seed in range(6):
        random.seed(seed)
        angle = random.choice([90, 180, 270])
        random.seed(seed)
        y, ym = T.RandomRotationTransform([90, 180, 270], p=1.1)((x, m))
        assert torch.equal(y, torch.rot90(x, angle // 90, (-2, -1))) and torch.equal(ym, torch.rot90(m, angle // 90, (-2, -1)))
        assert y.shape[-2:] == ((3, 2) if angle != 180 else (2, 3))
    # counter-clockwise: the top-right element moves to the top-left
    y = torch.rot90(x, 1, (-2, -1))
    assert y[0, 0, 0, 0] == x[0, 0, 0, 2]


def test_apply_normalize_matches_constants():
    s = {"S2": torch.rand(2, 4, 3, 3) * 5000, "S1": torch.randn(2, 2, 3, 3) * 5 - 12}
    ref = {k: v.clone() for k, v in s.items()}
    out = apply_normalize(s, default_dataset_stats())
    for c in range(4):
        assert torch.allclose(out["S2"][:, c], (ref["S2"][:, c] - stats.S2_MEAN[c]) / stats.S2_STD[c])
    for c in range(2):
        assert torch.allclose(out["S1"][:, c], (ref["S1"][:, c] - stats.S1_MEAN[c]) / stats.S1_STD[c])


def test_pipeline_order_and_joint_geometry():
    """S2 augmentation sees RAW values, then normalisation, then input/admin_mask move together."""
    torch.manual_seed(11)
    random.seed(11)
    s = {"S2": torch.rand(2, 4, 6, 5) * 10000, "S1": torch.randn(2, 2, 6, 5), "admin_mask": torch.arange(60.0).view(2, 6, 5),
         "building_counts": torch.rand(2, 1, 6, 5)}
    ref = {k: v.clone() for k, v in s.items()}
    tf = T.default_train_transform()
    torch.manual_seed(7)
    random.seed(7)
    out = apply_transformations_and_normalize(s, tf, default_dataset_stats())
    # replay the draws by hand, in the reference's order
    torch.manual_seed(7)
    random.seed(7)
    s2 = ref["S2"]
    if torch.rand(1) < 0.9:
        s2 = (s2 / 10000 * random.uniform(0.666, 1.5)).clamp(0, 1) * 10000
    if torch.rand(1) < 0.9:
        s2 = (torch.clip(s2, min=0) / 10000).pow(random.uniform(0.6666, 1.5)).clamp(0, 1) * 10000
    x = torch.cat([apply_normalize({"S2": s2}, default_dataset_stats())["S2"],
                   apply_normalize({"S1": ref["S1"].clone()}, default_dataset_stats())["S1"]], 1)
    m = torch.cat([ref["admin_mask"].unsqueeze(1), ref["building_counts"]], 1)
    if torch.rand(1) < 0.5:
        x, m = x.flip(-2), m.flip(-2)
    if torch.rand(1) < 0.5:
        x, m = x.flip(-1), m.flip(-1)
    if torch.rand(1) < 0.75:
        k = random.choice([90, 180, 270]) // 90
        x, m = torch.rot90(x, k, (-2, -1)), torch.rot90(m, k, (-2, -1))
    assert torch.allclose(out["input"], x) and torch.equal(out["admin_mask"], m[:, 0]) and torch.equal(out["building_counts"], m[:, 1:2])
    assert out["admin_mask"].dim() == 3
