"""Drop-in boundary checks that need no GPU: registry, constructor, state-dict keys/shapes/order, seeded head
init, checkpoint round trip, loud failure on CPU tensors, C-ABI symbol export."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(seed=1600, **kw):
    from popcorn_amd.model import POPCORN
    torch.manual_seed(seed)
    args = dict(input_channels=6, feature_extractor="DDA", occupancymodel=True, pretrained=True, biasinit=0.9407,
                sentinelbuildings=True)
    args.update(kw)
    return POPCORN(**args)


def test_registry_matches_reference_kwargs():
    from popcorn_amd.model import Args, calculate_input_channels, get_model_kwargs, model_dict
    args = Args(Sentinel1=True, NIR=True, Sentinel2=True, feature_extractor="DDA", occupancymodel=True, pretrained=True,
                biasinit=0.9407, sentinelbuildings=True)
    kw = get_model_kwargs(args, "POPCORN")
    assert repr(sorted(kw.items())) == open(os.path.join(G, "g1_model_kwargs.txt")).read().strip()
    assert calculate_input_channels(args) == 6
    assert calculate_input_channels(args._replace(Sentinel2=False, NIR=False)) == 2
    assert calculate_input_channels(args._replace(Sentinel1=False)) == 4
    assert "POPCORN" in model_dict
    with pytest.raises(ValueError):
        get_model_kwargs(args, "NOPE")


def test_state_dict_keys_shapes_order_and_params():
    m = _build()
    lines = [l.rstrip("\n").split("\t") for l in open(os.path.join(G, "g1_state_dict_keys.txt"))]
    sd = m.state_dict()
    assert list(sd.keys()) == [l[0] for l in lines]                       # same keys, same ORDER
    pnames = {n for n, _ in m.named_parameters()}
    for k, shp, kind in lines:
        assert ",".join(map(str, sd[k].shape)) == shp, k
        assert (k in pnames) == (kind == "P"), k
    assert m.num_params == int(np.load(os.path.join(G, "g1_head_seed1600.npz"))["num_params"]) == 39799
    names, params = m.trainable()
    assert len(names) == 56 and sum(p.numel() for p in params) == 39298


def test_seeded_head_init_equals_reference_and_weights_loaded():
    m = _build(seed=1600)
    hd = np.load(os.path.join(G, "g1_head_seed1600.npz"))
    for k in hd.files:
        if k.startswith("head."):
            assert np.array_equal(m.state_dict()[k].numpy(), hd[k]), k    # same RNG stream as the reference ctor
    ck = np.load(os.path.join(G, "g1_dda_checkpoint.npz"))
    for pre in ("unetmodel.", "building_extractor."):
        for k in ck.files:
            if k != "step":
                assert np.array_equal(m.state_dict()[pre + k].numpy(), ck[k]), pre + k


def test_not_pretrained_reinit_keeps_running_stats():
    m = _build(pretrained=False)
    ck = np.load(os.path.join(G, "g1_dda_checkpoint.npz"))
    sd = m.state_dict()
    k = "sar_stream.inc.conv.conv.1."
    assert np.array_equal(sd["unetmodel." + k + "running_mean"].numpy(), ck[k + "running_mean"])      # popcorn.py:59-66
    assert torch.all(sd["unetmodel." + k + "weight"] == 1) and torch.all(sd["unetmodel." + k + "bias"] == 0)
    assert not np.array_equal(sd["unetmodel.sar_stream.inc.conv.conv.0.weight"].numpy(), ck["sar_stream.inc.conv.conv.0.weight"])
    # ConvTranspose2d is not an nn.Conv2d instance: left untouched by the reference's re-init
    assert np.array_equal(sd["unetmodel.sar_stream.up_seq.up2.up.weight"].numpy(), ck["sar_stream.up_seq.up2.up.weight"])
    assert np.array_equal(sd["building_extractor.sar_stream.inc.conv.conv.0.weight"].numpy(), ck["sar_stream.inc.conv.conv.0.weight"])


def test_checkpoint_roundtrip_reference_format(tmp_path):
    """run_train.py:445-456 / run_eval.py:243-257: {'model': state_dict, ...} round trip."""
    m = _build(seed=1)
    path = tmp_path / "last_model.pth"
    torch.save({"model": m.state_dict(), "epoch": 3, "iter": 10}, path)
    m2 = _build(seed=2)
    m2.load_state_dict(torch.load(path, weights_only=False)["model"])
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


def test_optimizer_grouping_by_name():
    """run_train.py:82-85 groups by the strings 'unetmodel' and 'head.6.*'."""
    m = _build()
    head_name = ["head.6.weight", "head.6.bias"]
    named = list(m.named_parameters())
    with_decay = [n for n, _ in named if n not in head_name and "unetmodel" not in n]
    unet_only = [n for n, _ in named if n not in head_name and "unetmodel" in n]
    no_decay = [n for n, _ in named if n in head_name and "unetmodel" not in n]
    assert len(no_decay) == 2 and len(unet_only) == 98 and len(with_decay) == 6 + 98
    torch.optim.Adam([{"params": [p for n, p in named if n in with_decay]}], lr=1e-4)


def test_cpu_tensors_fail_loudly_no_fallback():
    from popcorn_amd._lib import PopcornHipError
    m = _build()
    with pytest.raises(PopcornHipError):
        m({"input": torch.zeros(1, 6, 32, 32)}, padding=False)
    with pytest.raises(RuntimeError):
        m.unetmodel.sar_stream.inc(torch.zeros(1, 2, 8, 8))      # parameter containers have no torch forward
    with pytest.raises(NotImplementedError):
        _build(input_channels=0)
    m2 = _build(input_channels=2)                      # S1-only: 8 head inputs (popcorn.py:68-69)
    assert m2.head[0].weight.shape == (64, 8, 1, 1) and len(m2.trainable()[0]) == 32


def test_pad_geometry_matches_reference_rule():
    from popcorn_amd.model.popcorn import pad_geometry
    assert pad_geometry(100, 100, True) == (14, 14, 14, 14)
    assert pad_geometry(100, 100, False) == (14, 14, 14, 14)
    assert pad_geometry(131, 77, False) == (30, 31, 25, 26)
    assert pad_geometry(64, 96, False) == (0, 0, 0, 0)
    assert pad_geometry(2048, 2048, False) == (0, 0, 0, 0)


def test_c_abi_exports_every_declared_symbol():
    """The library loads on a CPU-only host and exports every function include/popcorn_hip.h declares."""
    from popcorn_amd import _lib as L
    lib = L.lib()
    hdr = open(os.path.join(ROOT, "include", "popcorn_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(pc_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in popcorn_hip.h but not exported"
    assert lib.pc_abi_version() == L.PC_ABI_VERSION == 9
    assert isinstance(lib.pc_device_count(), int)
    assert lib.pc_error_string(-1).decode().startswith("invalid")
    # struct layouts agree with the header (sizes of the C structs, from the compiler)
    assert ctypes.sizeof(L.PcSrc) == lib.pc_sizeof(0) and ctypes.sizeof(L.PcDst) == lib.pc_sizeof(1)
    assert ctypes.sizeof(L.PcBn) == lib.pc_sizeof(2)
    assert ctypes.sizeof(L.PcConvFwdDesc) == lib.pc_sizeof(3)
    assert ctypes.sizeof(L.PcAdamGroups) == lib.pc_sizeof(4)
    assert ctypes.sizeof(L.PcLevel2FwdDesc) == lib.pc_sizeof(5)


def test_graft_entry_build_passes_on_the_current_tree():
    """`__graft_entry__.build()` is the driver's "does it build" check: make (a no-op when the in-tree library is current), import,
    ABI assertion.  (Round 4 bumped the ABI twice; the assertion inside build() must follow the binding.)"""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    try:
        g = importlib.import_module("__graft_entry__")
        g.build()
    finally:
        sys.path.remove(root)
