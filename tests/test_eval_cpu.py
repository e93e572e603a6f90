"""Evaluation-side host logic vs reference-generated fixtures (g7) and the oracle: patch indices, interior mask,
collate, augmentation consistency.  CPU only."""
import os

import numpy as np
import torch

from oracle import popcorn_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


def test_patch_indices_vs_reference_golden():
    from popcorn_amd.eval import get_patch_indices
    g = np.load(os.path.join(G, "g7_dataset_helpers.npz"))
    for key in [k for k in g.files if k.startswith("patch_indices/")]:
        _, hw, fs = key.split("/")
        h, w = map(int, hw.split("x"))
        four = fs == "fs1"
        ref = g[key]
        assert np.array_equal(get_patch_indices(h, w, 2048, 128, four).numpy(), ref), key
        assert np.array_equal(O.get_patch_indices(h, w, 2048, 128, four).numpy(), ref), key


def test_create_mask_vs_reference_golden():
    from popcorn_amd.eval import create_mask
    g = np.load(os.path.join(G, "g7_dataset_helpers.npz"))
    assert np.array_equal(create_mask(20, 30, 4).numpy(), g["create_mask/20x30_o4"])
    assert np.array_equal(O.create_mask(20, 30, 4), g["create_mask/20x30_o4"])


def test_collate_vs_reference_golden():
    from popcorn_amd.data.collate import Population_Dataset_collate_fn
    g = np.load(os.path.join(G, "g7_dataset_helpers.npz"))
    batch = []
    for i in range(3):
        batch.append({"S2": torch.from_numpy(g[f"collate/in{i}/S2"]), "S1": torch.from_numpy(g[f"collate/in{i}/S1"]),
                      "admin_mask": torch.from_numpy(g[f"collate/in{i}/admin_mask"]), "y": torch.tensor(float(10 + i)),
                      "img_coords": (i, 2 * i), "valid_coords": (i, i), "season": i % 4, "census_idx": torch.tensor([i + 3])})
    for fn in (Population_Dataset_collate_fn, O.collate_fn):
        out = fn(batch)
        for k in ("S2", "S1", "admin_mask", "y", "season", "census_idx"):
            assert np.array_equal(out[k].numpy(), g[f"collate/out/{k}"]), (fn.__name__, k)


def test_geometric_augment_keeps_input_and_mask_aligned():
    from popcorn_amd.data.collate import augment_geometric
    g = torch.Generator().manual_seed(3)
    B, H, W = 4, 6, 9
    mask = torch.arange(B * H * W, dtype=torch.float32).view(B, H, W)
    inp = mask.unsqueeze(1).repeat(1, 3, 1, 1).clone()
    for _ in range(8):
        a, m = augment_geometric(inp, mask, generator=g)
        assert a.shape[-2:] == m.shape[-2:]
        assert torch.equal(a[:, 0], m) and torch.equal(a[:, 2], m)
        assert torch.equal(torch.sort(m.reshape(B, -1), dim=1)[0], mask.reshape(B, -1))


def test_census_loop_oracle_equals_bincount():
    rng = np.random.default_rng(5)
    h, w, nreg = 40, 50, 6
    boundary = torch.from_numpy(rng.integers(0, nreg, size=(h, w)).astype(np.float32))
    pred = torch.from_numpy(rng.random((h, w)).astype(np.float32))
    idx = list(range(nreg))
    bbox = [(0, h, 0, w)] * nreg
    loop = O.convert_popmap_to_census_loop(pred, boundary, idx, bbox)
    binc = O.census_sums(pred.numpy(), boundary.numpy(), nreg)
    np.testing.assert_allclose(loop.numpy(), binc, rtol=1e-5)
