"""popcorn_amd/data/feed.py: the double-buffered pinned-host feed of the fused step (counterpart of the reference's per-step
`to_cuda_inplace`, run_train.py:186 / utils/utils.py:22-27) and the measured choice of its copy stream."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _trainer():
    from popcorn_amd.model import POPCORN
    from popcorn_amd.train import FusedTrainStep
    torch.manual_seed(1600)
    m = POPCORN(6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
    return FusedTrainStep(m, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01, use_graph=True)


def test_pick_copy_stream_returns_a_side_stream_that_overlaps_and_is_cached():
    from popcorn_amd.data import feed
    feed._PICKED.clear()
    s = feed.pick_copy_stream(verbose=True)
    assert isinstance(s, torch.cuda.Stream) and s.cuda_stream != torch.cuda.current_stream().cuda_stream
    assert feed.pick_copy_stream() is s
    # the property it was picked for, re-measured: a pinned copy on it next to a spin kernel of the same length on the compute stream
    # takes clearly less than the two in a row
    import time
    n = 16 << 20
    host = torch.empty(n, dtype=torch.uint8).pin_memory()
    dst = torch.empty(n, dtype=torch.uint8, device="cuda")

    def copy_only():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(s):
            dst.copy_(host, non_blocking=True)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    copy_only()
    t_copy = min(copy_only() for _ in range(3))
    torch.cuda._sleep(1 << 20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    torch.cuda._sleep(1 << 20)
    torch.cuda.synchronize()
    spin = int(t_copy / ((time.perf_counter() - t0) / (1 << 20)))
    ev = torch.cuda.Event()
    both = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(s):
            dst.copy_(host, non_blocking=True)
            ev.record(s)
        torch.cuda._sleep(spin)
        torch.cuda.synchronize()
        both.append(time.perf_counter() - t0)
    assert min(both) < 1.6 * t_copy, (min(both), t_copy)          # serialised would be ~2x


@pytest.mark.parametrize("kind", ["raw", "split"])
def test_host_feed_equals_resident_steps_bit_for_bit(kind):
    """Five batches through HostFeed (pinned host -> idle static set on the copy stream -> that set's graph) give the losses and the
    final parameters of the same batches copied synchronously into ONE static set."""
    from popcorn_amd.data import stats
    from popcorn_amd.data.feed import HostFeed
    from popcorn_amd.data.synthetic import make_raw_batch
    B, H, W = 4, 100, 100
    b6 = list(stats.BAND6)
    batches = [make_raw_batch(B, H, W, seed=300 + i, device="cpu", region="disc") for i in range(5)]
    ref = _trainer()
    ref.raw_norm = (tuple(range(6)), stats.MEAN6, stats.STD6)

    def host_of(b, tr):
        packed = tr.pack_small(b["admin_mask"], b["y"], b["census_idx"]).pin_memory()
        if kind == "split":
            return {"_rawpacked": tr.pack_split(b["raw"][:, b6[:4]].round().to(torch.int32).to(torch.uint16).contiguous(),
                                                b["raw"][:, b6[4:]].contiguous()).pin_memory(), "_packed": packed}
        return {"raw": b["raw"][:, b6].contiguous().pin_memory(), "_packed": packed}

    one = ref.static_buffers(B, H, W, split=True) if kind == "split" else ref.static_buffers(B, H, W, raw_channels=6)
    want = []
    for i, b in enumerate(batches):
        for k, v in host_of(b, ref).items():
            one[k].copy_(v)
        torch.manual_seed(50 + i)
        want.append(ref.step(one).tolist())
    tr = _trainer()
    tr.raw_norm = (tuple(range(6)), stats.MEAN6, stats.STD6)
    feed = HostFeed(tr, B, H, W, kind=kind, raw_channels=6)
    got = []
    hosts = [host_of(b, tr) for b in batches]
    # the selection grid is drawn on the CPU generator when a step is ENQUEUED: step i of the feed is enqueued by call i + 1
    for i in range(len(hosts) + 1):
        if i >= 1:
            torch.manual_seed(50 + i - 1)
        if i < len(hosts):
            loss = feed.step(hosts[i])
        else:
            loss = feed.flush()
        if loss is not None:
            got.append(loss.tolist())
    torch.cuda.synchronize()
    assert feed.flush() is None
    assert got == want
    assert torch.equal(tr.flat_p, ref.flat_p)


@pytest.mark.parametrize("shape", [(2, 37, 52), (1, 64, 64), (3, 20, 33)])
def test_augment_raw_equals_the_transform_classes_and_their_generator_consumption(shape):
    """pc_augment_raw + draw_fused_params against the reference-named classes applied the way the trainer applies them
    (utils/utils.py:130-214: S2 transforms on the digital numbers, cat, normalise, joint geometric transform of input and admin_mask):
    over 24 seeds (every flip / rotation / brightness / gamma combination occurs) the raw tile normalises to the same input (the power
    function differs by an ulp between torch and the kernel), the admin mask is identical, and BOTH generators (torch's and Python's)
    end in the same state."""
    import random
    from popcorn_amd import ops
    from popcorn_amd.cli import normalize_sample, prepare_sample_fused
    from popcorn_amd.data import stats
    from popcorn_amd.utils.transform import default_train_transform
    B, H, W = shape
    tr = default_train_transform()
    seen = set()
    for seed in range(24):
        g = torch.Generator().manual_seed(seed)
        smp = {"S2": torch.randint(0, 10000, (B, 4, H, W), generator=g).float(), "S1": torch.randn(B, 2, H, W, generator=g) * 4 - 12,
               "admin_mask": torch.randint(0, 7, (B, H, W), generator=g).float(), "y": torch.rand(B, generator=g),
               "census_idx": torch.arange(1, B + 1)}
        torch.manual_seed(100 + seed)
        random.seed(100 + seed)
        ref = normalize_sample({k: v.clone() for k, v in smp.items()}, torch.device("cuda"), tr)
        end_t, end_r = torch.get_rng_state(), random.getstate()
        torch.manual_seed(100 + seed)
        random.seed(100 + seed)
        fast = prepare_sample_fused({k: v.cuda() for k, v in smp.items()}, tr)
        assert torch.equal(torch.get_rng_state(), end_t) and random.getstate() == end_r
        x = ops.select_normalize(fast["raw"], tuple(range(6)), stats.MEAN6, stats.STD6)
        assert x.shape == ref["input"].shape
        torch.testing.assert_close(x, ref["input"], rtol=2e-6, atol=2e-5)
        assert torch.equal(fast["admin_mask"], ref["admin_mask"])
        assert torch.equal(fast["census_idx"], ref["census_idx"]) and torch.equal(fast["y"], ref["y"])
        seen.add(tuple(x.shape[2:]) == (H, W))
    assert seen == {True, False} or H == W


def test_trainer_fed_one_batch_ahead_takes_the_reference_order_steps(tmp_path):
    """Trainer.train() through RegionFeed (pinned loader, copy stream, one-launch augmentation, raw input form) against the same trainer
    stepped the reference's way (synchronous copy, per-op augmentation launches, normalised input): same seeds -> same batches, same
    coins, same selection grids; after an epoch of variable-size regions the parameters agree to the rounding of the differing power
    function (1e-5 relative on the flat parameter vector; the two runs take identical optimisation decisions)."""
    import random
    from popcorn_amd.cli import Trainer, normalize_sample, train_parser, limit_regime
    argv = ("-S2 -NIR -S1 -occmodel -senbuilds -pret -wd 1e-5 --biasinit 0.9407 -lr 1e-4 --synthetic_regions 12 -wb 2 "
            f"--save_dir {tmp_path} -lt 100 -val 100 -e 1 --synthetic_hw_range 90 200 --save-model no").split()
    ta = Trainer(train_parser().parse_args(argv))
    ta.train()
    torch.cuda.synchronize()
    tb = Trainer(train_parser().parse_args(argv))
    tb.model.train()
    a = tb.args
    for sample in tb.loader:                       # the pre-round-6 loop body
        s = normalize_sample(sample, tb.device, tb.data_transform)
        n = s["input"].shape[0] * s["input"].shape[2] * s["input"].shape[3]
        e, u, k = limit_regime(n, a.limit1, a.limit2, a.limit3)
        tb.fused.step(s, encoder_no_grad=e, unet_no_grad=u)
    torch.cuda.synchronize()
    assert ta.fused.native_steps == 6 and int(ta.fused.step_count[0].item()) == int(tb.fused.step_count[0].item()) == 6
    pa, pb = ta.fused.flat_p, tb.fused.flat_p
    assert torch.isfinite(pa).all()
    assert (pa - pb).abs().max().item() <= 1e-5 * pb.abs().max().item()
