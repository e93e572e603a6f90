"""Replays the inputs of fixture g12_stitch.npz (the reference's own ``Trainer.test_target``, run_eval.py:71-203, driven by
tests/golden/make_golden.py: g12_stitch): windows from the reference's patch grid, ensemble members that are fixed per-pixel
functions of the normalised input.  Shared by the CPU (oracle) and GPU (HIP stitcher) tests."""
import os

import numpy as np
import torch

G = os.path.join(os.path.dirname(__file__), "golden")


def member_outputs(x, i):
    """ensemble member i of the fixture (make_golden.py: g12_stitch.Member): x = normalised (1,6,ps,ps) input"""
    pd_ = torch.relu(x[:, i % 6] * (0.5 + 0.25 * i) + x[:, (i + 3) % 6] * 0.125 + 0.75)
    sc = (x[:, (i + 1) % 6] * 0.5).abs() + 0.0625 * i
    return pd_, sc


def load_case(name, normalize):
    """-> dict(h, w, ips, ov, M, windows=[(x, y, popdense[M,ips,ips], scale[M,ips,ips])], ref maps ...).
    ``normalize``: (1,6,ps,ps) raw [S2(4), S1(2)] -> normalised input (the checker's or the product's restatement)."""
    g = np.load(os.path.join(G, "g12_stitch.npz"))
    h, w, ips, ov, M, fs = (int(v) for v in g[f"{name}/meta"])
    s2 = torch.from_numpy(g[f"{name}/s2"].astype(np.float32))
    s1 = torch.from_numpy(g[f"{name}/s1"].astype(np.float32))
    wins = []
    for x, y, s in g[f"{name}/windows"].tolist():
        raw = torch.cat([s2[s:s + 1, :, x:x + ips, y:y + ips], s1[s:s + 1, :, x:x + ips, y:y + ips]], 1)
        inp = normalize(raw)
        outs = [member_outputs(inp, i) for i in range(M)]
        wins.append((x, y, torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])))
    ref = {k: torch.from_numpy(g[f"{name}/{k}"]) for k in ("map", "std", "scale", "scale_std", "adjusted", "count", "boundary")}
    return dict(h=h, w=w, ips=ips, ov=ov, M=M, fourseasons=bool(fs), windows=wins, window_list=g[f"{name}/windows"], ref=ref,
                census_idx=g[f"{name}/census_idx"].tolist(), census_pop=g[f"{name}/census_pop"],
                metrics=dict(zip(g[f"{name}/metric_keys"].tolist(), g[f"{name}/metric_vals"].tolist())))


CASES = ("a", "b", "c")
