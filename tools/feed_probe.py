"""Where the end-to-end epoch of the trainer counterpart spends its time (bench.py: config3_epoch): the DataLoader alone (workers, pinned or not),
the loader + RegionFeed (copies, no step), the feed + augmentation, the full loop.    python3 tools/feed_probe.py [--workers 8]"""
import argparse
import json
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from popcorn_amd.cli import Trainer, prepare_sample_fused, train_parser  # noqa: E402
from popcorn_amd.data.collate import Population_Dataset_collate_fn  # noqa: E402
from popcorn_amd.data.feed import RegionFeed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=0)
    ap.add_argument("--regions", type=int, default=256)
    a = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="pc_probe_")
    argv = (f"-S2 -NIR -S1 -occmodel -senbuilds -pret -wd 1e-5 --biasinit 0.9407 -lr 1e-4 --synthetic_regions {a.regions} -wb 2 --save_dir {tmp} "
            f"-lt 1000000 -val 1000000 -e 1 --synthetic_hw_range 150 700 --save-model no -w {a.workers}").split()
    t = Trainer(train_parser().parse_args(argv))
    dev = t.device
    res = {}

    def timed(label, fn, reps=2):
        best = None
        for _ in range(reps + 1):            # first pass warms worker caches
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = fn()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        res[label] = {"ms_per_batch": round(best / n * 1e3, 3), "batches": n}
        print(label, res[label], flush=True)

    def loader_only():
        n = 0
        for _ in t.loader:
            n += 1
        return n
    timed("loader", loader_only)
    ds = t.loader.dataset
    plain = torch.utils.data.DataLoader(ds, batch_size=2, num_workers=a.workers, shuffle=True, collate_fn=Population_Dataset_collate_fn,
                                        drop_last=True, pin_memory=False, persistent_workers=a.workers > 0)

    def plain_only():
        n = 0
        for _ in plain:
            n += 1
        return n
    timed("loader (pin_memory=False)", plain_only)

    def feed_only():
        n = 0
        for _ in RegionFeed(t.loader, dev):
            n += 1
        return n
    timed("loader + RegionFeed (copies)", feed_only)

    def feed_plain():
        n = 0
        for _ in RegionFeed(plain, dev):
            n += 1
        return n
    timed("unpinned loader + RegionFeed (own staging)", feed_plain)

    def feed_aug():
        n = 0
        for s in RegionFeed(t.loader, dev):
            prepare_sample_fused(s, t.data_transform)
            n += 1
        return n
    timed("loader + RegionFeed + augment", feed_aug)

    from popcorn_amd.cli import limit_regime
    aa = t.args

    def feed_aug_step():
        n = 0
        for smp in RegionFeed(t.loader, dev):
            s = prepare_sample_fused(smp, t.data_transform)
            k = s["raw"].shape[0] * s["raw"].shape[2] * s["raw"].shape[3]
            e, u, skip = limit_regime(k, aa.limit1, aa.limit2, aa.limit3)
            t.fused.step(s, encoder_no_grad=e, unet_no_grad=u)
            n += 1
        return n
    timed("loader + RegionFeed + augment + fused.step", feed_aug_step)

    def staged_step():
        staged = [prepare_sample_fused(smp, t.data_transform) for smp in RegionFeed(t.loader, dev)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in staged:
            t.fused.step(s)
        torch.cuda.synchronize()
        res["resident fused.step"] = {"ms_per_batch": round((time.perf_counter() - t0) / len(staged) * 1e3, 3)}
        # device time of the augmentation launch alone
        smps = list(RegionFeed(t.loader, dev))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for smp in smps:
            prepare_sample_fused(smp, t.data_transform)
        e1.record()
        torch.cuda.synchronize()
        res["augment launch (device + host, back to back)"] = {"ms_per_batch": round(e0.elapsed_time(e1) / len(smps), 3)}
        return len(staged)
    staged_step()

    def train_loop():
        it0 = t.info["iter"]
        t.args.num_epochs = t.info["epoch"] + 1
        t.train()
        return t.info["iter"] - it0
    timed("Trainer.train()", train_loop)

    def collate_only():
        n = 0
        items = [ds[i] for i in range(len(ds))]
        t0 = time.perf_counter()
        for i in range(0, len(items) - 1, 2):
            Population_Dataset_collate_fn(items[i:i + 2])
            n += 1
        res["collate alone (main thread)"] = {"ms_per_batch": round((time.perf_counter() - t0) / n * 1e3, 3)}
        return n
    collate_only()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
