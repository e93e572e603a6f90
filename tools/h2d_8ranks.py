"""Feeding N data-parallel ranks from ONE host: N processes (one per rank, all on the box's one GPU), each streaming the fp32 6-band batch
of the default loader (B = 64 tiles of 6 x 100 x 100: 15.4 MB per step) from PINNED memory through its own copy stream, concurrently.
Prints the aggregate and per-rank sustained rates next to what N ranks need at the resident step rate (bench.py: h2d_gbps_needed_8_ranks).

    python tools/h2d_8ranks.py [--ranks 8] [--seconds 3] [--step-ms 1.67]

On the single-GPU test box all N streams share ONE PCIe link, so the aggregate here is a lower bound of what N links give on the 8-GPU
node; what the run does establish is that N concurrent pinned producers on one host do not collapse (host memory / IOMMU / driver locks)."""
import argparse
import json
import os
import sys
import time

import torch
import torch.multiprocessing as mp


def worker(rank, n, seconds, nbytes, barrier, q):
    torch.cuda.set_device(0)
    host = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    host.random_(0, 255)
    dev = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(2)]
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for i in range(4):
            dev[i & 1].copy_(host, non_blocking=True)
    st.synchronize()
    barrier.wait()
    t0 = time.perf_counter()
    k = 0
    with torch.cuda.stream(st):
        while time.perf_counter() - t0 < seconds:
            for i in range(8):
                dev[i & 1].copy_(host, non_blocking=True)
            st.synchronize()
            k += 8
    dt = time.perf_counter() - t0
    q.put((rank, k * nbytes / dt / 1e9))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--step-ms", type=float, default=1.67, help="resident step time: what one rank's feed must keep up with")
    ap.add_argument("--out", default=None)
    ap.add_argument("--feed", choices=["fp32", "split"], default="fp32",
                    help="split: S2 as uint16 digital numbers + S1 fp32 (160 KB per tile, the default feed of bench.py --gpus N since round 6)")
    a = ap.parse_args()
    nbytes = 64 * 6 * 100 * 100 * 4 if a.feed == "fp32" else 64 * 100 * 100 * (4 * 2 + 2 * 4)
    res = {}
    for n in sorted({1, a.ranks}):
        ctx = mp.get_context("spawn")
        barrier, q = ctx.Barrier(n), ctx.Queue()
        procs = [ctx.Process(target=worker, args=(r, n, a.seconds, nbytes, barrier, q)) for r in range(n)]
        for p in procs:
            p.start()
        rates = dict(q.get(timeout=300) for _ in range(n))
        for p in procs:
            p.join(timeout=60)
        res[f"ranks_{n}"] = {"aggregate_gbps": round(sum(rates.values()), 2), "per_rank_gbps": [round(rates[r], 2) for r in range(n)]}
    need1 = nbytes / (a.step_ms * 1e-3) / 1e9
    res["needed_per_rank_gbps"] = round(need1, 2)
    res[f"needed_{a.ranks}_ranks_gbps"] = round(need1 * a.ranks, 2)
    res["feed"] = (f"{'fp32 6-band' if a.feed == 'fp32' else 'uint16 S2 (4 bands) + fp32 S1 (2 bands)'}, B = 64 tiles of 100 x 100: "
                   f"{nbytes / 1e6:.1f} MB per step, step {a.step_ms} ms")
    res["note"] = "all ranks share the ONE PCIe link of the single-GPU box: a lower bound for N links"
    print(json.dumps(res))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    sys.path.insert(0, os.getcwd())
    main()
