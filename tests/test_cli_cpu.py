"""CLI flag surface equals the reference's (names + defaults that the canonical recipes rely on).  CPU only."""


def test_train_flags_defaults_match_reference():
    from popcorn_amd.cli import train_parser
    a = train_parser().parse_args("-S2 -NIR -S1 -treg rwa -tregtrain rwa -occmodel -wd 1e-5 -senbuilds -pret --biasinit 0.9407".split())
    assert a.Sentinel1 and a.Sentinel2 and a.NIR and a.occupancymodel and a.sentinelbuildings and a.pretrained
    assert a.weightdecay == 1e-5 and a.biasinit == 0.9407
    # arguments/train.py defaults
    assert (a.weak_batch_size, a.learning_rate, a.loss, a.scale_regularization, a.lam_weak) == (2, 1e-4, ["log_l1_loss"], 0.01, 100.0)
    assert (a.limit1, a.limit2, a.limit3) == (9000000, 9000000, 13000000)
    assert (a.lr_step, a.lr_gamma, a.gradient_clip, a.num_epochs, a.seed) == (5, 0.75, 0.01, 100, 1600)


def test_eval_flags():
    from popcorn_amd.cli import eval_parser
    a = eval_parser().parse_args("-occmodel -senbuilds -S2 -NIR -S1 -treg rwa --fourseasons --resume a.pth b.pth".split())
    assert a.fourseasons and a.resume == ["a.pth", "b.pth"] and a.seed == 1610


def test_synthetic_dataset_and_collate_shapes():
    import torch
    from popcorn_amd.data.collate import Population_Dataset_collate_fn
    from popcorn_amd.data.dataset import SyntheticWeaksupDataset
    ds = SyntheticWeaksupDataset(8, seed=3)
    b = Population_Dataset_collate_fn([ds[0], ds[1], ds[2]])
    H = max(ds.hw[i][0] for i in range(3)); W = max(ds.hw[i][1] for i in range(3))
    assert b["S2"].shape == (3, 4, H, W) and b["S1"].shape == (3, 2, H, W) and b["admin_mask"].shape == (3, H, W)
    assert torch.equal(b["census_idx"], torch.tensor([1, 2, 3]))
    assert (b["admin_mask"][0, ds.hw[0][0]:, :] == -1).all()


def test_limit_regime_nests_like_the_reference():
    """run_train.py:191-198 with the defaults of arguments/train.py:34-36 (9e6 / 9e6 / 13e6)."""
    from popcorn_amd.cli import limit_regime, train_parser
    a = train_parser().parse_args([])
    assert (a.limit1, a.limit2, a.limit3, a.weak_batch_size) == (9000000, 9000000, 13000000, 2)
    L = (a.limit1, a.limit2, a.limit3)
    assert limit_regime(2 * 2100 * 2140, *L) == (False, False, False)          # 8.988e6 px: everything trains
    assert limit_regime(9000000, *L) == (False, False, False)                 # strict ">"
    assert limit_regime(2 * 2100 * 2150, *L) == (True, True, False)            # 9.03e6 px: head only (limit1 == limit2)
    assert limit_regime(13000001, *L) == (True, True, True)                   # skipped
    assert limit_regime(500, 100, 1000, 2000) == (True, False, False)         # decoder + head
    assert limit_regime(5000, 100, 10000, 2000) == (True, False, False)       # limit3 only applies beyond limit2


def test_every_tool_script_still_compiles():
    """tools/*.py are measurement scaffolding that DESIGN.md cites; nothing runs them on the CPU box, but each must at least parse
    against the current tree (a renamed op breaks them silently otherwise: the names they import from popcorn_amd are checked too)."""
    import ast
    import glob
    import importlib
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "tools", "*.py")))
    assert len(files) >= 30
    for f in files:
        tree = ast.parse(open(f).read(), filename=f)
        for node in ast.walk(tree):
            if isinstance(node, ast.ImportFrom) and node.module and node.module.startswith("popcorn_amd") and node.level == 0:
                mod = importlib.import_module(node.module)
                for alias in node.names:
                    assert hasattr(mod, alias.name) or importlib.util.find_spec(node.module + "." + alias.name) is not None, (f, node.module, alias.name)
