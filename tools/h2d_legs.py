"""bench.py's three h2d legs (6-band fp32 / uint16 split / 15-band) with per-slot timing events (why is the
narrowest feed the slowest?).  For every step i:   c0 / c1 = before / after the H2D copies on the copy stream,   g0 / g1 = before / after
the replayed graph on the compute stream.  Reported per feed (medians over the timed steps, ms):

    copy        c1 - c0               duration of the step's H2D copies (raw tile + the packed small buffer)
    graph       g1 - g0               duration of the replayed step while the NEXT copy runs beside it
    gap         g0(i) - g1(i-1)       compute stream idle between two steps
    copy_lag    c0(i+1) - g1(i-1)     from "set consumed" to the copy's start   (event propagation, copy stream)
    wait_lag    g0(i) - c1(i)         from "copy done" to the graph's start     (> 0: the compute stream was waiting for the copy)
    step        g1(i) - g1(i-1)       = graph + gap

plus three control runs on the same graphs: resident (no events, no copies), events only (no copies), copies only (compute does not wait).
GPU only:   gpurun -- 'python tools/h2d_legs.py --out gpurun_out/r5_h2d_legs.json'"""
import argparse
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402
from popcorn_amd.data import stats  # noqa: E402
from popcorn_amd.data.synthetic import make_raw_batch  # noqa: E402
from popcorn_amd.model import POPCORN  # noqa: E402
from popcorn_amd.train import FusedTrainStep  # noqa: E402


def med(v):
    return round(statistics.median(v), 4) if v else None


def leg(tr, batch, feed_kind, nsteps, timing=True, copy_mode="events", priority=0, modes=("feed", "resident", "events_only", "copies_only", "inline")):
    """copy_mode: 'events' = the bench's double-buffered feed with the copy on a side stream (two cross-stream event edges per step);
    'inline' = the copies on the COMPUTE stream in front of the graph (no cross-stream edge, no overlap)."""
    B, _, H, W = batch["raw"].shape
    keep = tr.raw_norm
    packed = tr.pack_small(batch["admin_mask"].cpu(), batch["y"].cpu(), batch["census_idx"].cpu()).pin_memory()
    b6 = list(stats.BAND6)
    if feed_kind == "split":
        tr.raw_norm = (tuple(range(6)), stats.MEAN6, stats.STD6)
        host = {"_rawpacked": tr.pack_split(batch["raw"][:, b6[:4]].round().to(torch.int32).cpu().to(torch.uint16).contiguous(),
                                            batch["raw"][:, b6[4:]].contiguous().cpu()).pin_memory(), "_packed": packed}
        sets = [tr.static_buffers(B, H, W, split=True, slot=sl) for sl in (0, 1)]
    else:
        raw = batch["raw"] if feed_kind == "15" else batch["raw"][:, b6].contiguous()
        if feed_kind != "15":
            tr.raw_norm = (tuple(range(6)), stats.MEAN6, stats.STD6)
        host = {"raw": raw.cpu().pin_memory(), "_packed": packed}
        sets = [tr.static_buffers(B, H, W, raw_channels=raw.shape[1], slot=sl) for sl in (0, 1)]
    nbytes = sum(v.numel() * v.element_size() for v in host.values())
    copied = [torch.cuda.Event() for _ in range(2)]
    consumed = [torch.cuda.Event() for _ in range(2)]
    if priority == "pick":
        from popcorn_amd.data.feed import pick_copy_stream
        cs = pick_copy_stream(verbose=True)
    else:
        cs = torch.cuda.Stream(priority=priority)
    cur = torch.cuda.current_stream()
    out = {"feed": feed_kind, "host_bytes_per_step": nbytes, "copy_stream_priority": priority, "copy_stream_handle": hex(cs.cuda_stream)}

    def run(n, mode, rec=None):
        # mode: "feed" | "resident" | "events_only" | "copies_only" | "inline"
        for e in consumed:
            e.record(cur)

        def feed(i):
            s = i & 1
            with torch.cuda.stream(cs):
                cs.wait_event(consumed[s])
                if rec is not None:
                    rec["c0"][i].record(cs)
                if mode != "events_only":
                    for k, v in host.items():
                        sets[s][k].copy_(v, non_blocking=True)
                if rec is not None:
                    rec["c1"][i].record(cs)
                copied[s].record(cs)
        if mode in ("feed", "events_only", "copies_only"):
            feed(0)
        for i in range(n):
            s = i & 1
            if mode in ("feed", "events_only", "copies_only") and i + 1 < n:
                feed(i + 1)
            if mode in ("feed", "events_only"):
                cur.wait_event(copied[s])
            if mode == "inline":
                for k, v in host.items():
                    sets[s][k].copy_(v, non_blocking=True)
            if rec is not None:
                rec["g0"][i].record(cur)
            tr.step(sets[s])
            if rec is not None:
                rec["g1"][i].record(cur)
            if mode != "resident" and mode != "inline":
                consumed[s].record(cur)

    for mode in modes:
        run(8, mode)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(nsteps, mode)
        torch.cuda.synchronize()
        out["ms_per_step_" + mode] = round((time.perf_counter() - t0) / nsteps * 1e3, 4)
    if timing:
        n = nsteps
        rec = {k: [torch.cuda.Event(enable_timing=True) for _ in range(n)] for k in ("c0", "c1", "g0", "g1")}
        run(8, "feed")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(n, "feed", rec)
        torch.cuda.synchronize()
        out["ms_per_step_feed_with_timing_events"] = round((time.perf_counter() - t0) / n * 1e3, 4)
        lo = 8
        el = lambda a, b: a.elapsed_time(b)  # noqa: E731
        out["events_ms"] = {
            "copy": med([el(rec["c0"][i], rec["c1"][i]) for i in range(lo, n)]),
            "graph": med([el(rec["g0"][i], rec["g1"][i]) for i in range(lo, n)]),
            "gap": med([el(rec["g1"][i - 1], rec["g0"][i]) for i in range(lo, n)]),
            "copy_lag": med([el(rec["g1"][i - 1], rec["c0"][i + 1]) for i in range(lo, n - 1)]),
            "wait_lag": med([el(rec["c1"][i], rec["g0"][i]) for i in range(lo, n)]),
            "step": med([el(rec["g1"][i - 1], rec["g1"][i]) for i in range(lo, n)]),
            "copy_start_after_graph_start": med([el(rec["g0"][i], rec["c0"][i + 1]) for i in range(lo, n - 1)]),
        }
    tr.raw_norm = keep
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/r5_h2d_legs.json")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--priority", default="0", help="priority of the copy stream (0 / -1), or 'pick' = popcorn_amd.data.feed.pick_copy_stream")
    ap.add_argument("--alias-scan", type=int, default=0, help="N: the 15-band leg on N freshly created copy streams in a row (which ones serialise?)")
    a = ap.parse_args()
    torch.manual_seed(1600)
    m = POPCORN(6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
    m.set_precision(a.precision)
    tr = FusedTrainStep(m, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01, use_graph=True)
    batch = make_raw_batch(64, 100, 100, seed=1600, device="cuda")
    res = {"precision": a.precision, "steps": a.steps, "legs": []}
    if a.alias_scan:
        # warm clocks first
        leg(tr, batch, "15", a.steps, timing=False, modes=("resident",))
        scan = []
        for prio in (0, -1):
            for j in range(a.alias_scan):
                r = leg(tr, batch, "15", a.steps, timing=False, priority=prio, modes=("feed",))
                scan.append({"priority": prio, "stream": r["copy_stream_handle"], "ms_per_step_feed": r["ms_per_step_feed"]})
                print(json.dumps(scan[-1]), flush=True)
        res["alias_scan_15_band"] = scan
    for order in (("6", "split", "15"), ("15", "split", "6")):          # both orders: is it the leg or its place in the sequence?
        for kind in order:
            r = leg(tr, batch, kind, a.steps, priority=a.priority if a.priority == "pick" else int(a.priority))
            r["order"] = "".join(o + " " for o in order).strip()
            res["legs"].append(r)
            print(json.dumps(r), flush=True)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
