"""Pre-flight of a multi-GPU scaling run: run it on the N-GPU node BEFORE `bench.py --gpus N`, as

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29533 tools/scale_preflight.py [--steps 200]

(functional check on one GPU: POPCORN_DIST_BACKEND=gloo with N = 2).  Rank 0 prints one JSON object:
  * `ranks_seen` / `rccl_ranks_seen`: distinct (rank, device, PCI bus id) triples gathered through the job's own backend -- must equal N;
  * `collectives_us`: the step's two collectives ALONE (the 157 KB flat-gradient SUM all-reduce and the 16-byte statistics all-reduce),
    median of 200 back-to-back calls each -- the per-step cost data parallelism adds over xGMI;
  * `dp_graph_ab`: `--steps` optimisation steps with both collectives captured inside ONE HIP graph (POPCORN_DP_ONE_GRAPH=1) and as THREE
    graphs with the collectives between them (=0), each compared with the SINGLE-PROCESS trajectory on the concatenated batch
    (max relative parameter distance; the data-parallel step reproduces the single-process gradient by construction: global normalisers,
    one SUM all-reduce, clip after the reduce) + ms per step of either form;
  * `ok`: all ranks present, both forms finite and within 1e-3 (relative to the largest parameter) of the single-process parameters after
    `--steps` Adam steps: rounding-order differences grow to ~1e-5 in a dozen steps; a lost or stale collective shows as >> 1e-3."""
import argparse
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class _LocalReducer:
    """No collectives: the single-process reference trajectory, computed inside the distributed job by every rank alike."""
    active, capture_failed, world = False, False, 1

    def reduce_stats(self, t):
        return t

    def reduce_grads(self, t):
        return t

    def global_batch(self, b):
        return b

    def capturable(self):
        return False

    def all_agree(self, ok):
        return bool(ok)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--batch", type=int, default=8, help="tiles per rank (small: the check is about equality, not speed)")
    a = ap.parse_args()
    import torch.distributed as dist
    from popcorn_amd.data.synthetic import make_raw_batch
    from popcorn_amd.distributed import FlatReducer, init_from_env
    from popcorn_amd.model import POPCORN
    from popcorn_amd.train import FusedTrainStep

    rank, local_rank, world = init_from_env()
    ndev = torch.cuda.device_count()
    local_rank %= max(1, ndev)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    out = {"world": world, "backend": dist.get_backend() if dist.is_initialized() else None}
    pr = torch.cuda.get_device_properties(dev)
    mine = f"{rank}:{local_rank}:{getattr(pr, 'pci_bus_id', '?')}"
    seen = [None] * world
    if world > 1:
        dist.all_gather_object(seen, mine)
    else:
        seen = [mine]
    out["devices"] = seen
    out["ranks_seen"] = len(set(seen))
    out["rccl_ranks_seen"] = out["ranks_seen"] if out["backend"] == "nccl" else None
    out["distinct_devices"] = len({s.split(":", 1)[1] for s in seen})

    # ---- the two collectives alone
    red = FlatReducer()
    flat = torch.randn(39298, device=dev)
    stats = torch.zeros(2, device=dev, dtype=torch.float64)
    coll = {}
    for name, fn in (("grad_allreduce_157KB", lambda: red.reduce_grads(flat)), ("stats_allreduce_16B", lambda: red.reduce_stats(stats))):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(200):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e6)
        coll[name] = round(statistics.median(ts), 1)
    out["collectives_us"] = coll

    # ---- one-graph vs three-graph data-parallel step against the single-process trajectory
    def trajectory(one_graph, single):
        os.environ["POPCORN_DP_ONE_GRAPH"] = "1" if one_graph else "0"
        torch.manual_seed(1600)
        m = POPCORN(6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).to(dev)
        B = a.batch * (world if single else 1)
        tr = FusedTrainStep(m, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01, use_graph=True, reducer=_LocalReducer() if single else FlatReducer())
        full = make_raw_batch(a.batch * world, 100, 100, seed=77, device=dev)
        sl = slice(0, B) if single else slice(rank * a.batch, (rank + 1) * a.batch)
        smp = tr.static_buffers(B, 100, 100, raw_channels=full["raw"].shape[1])
        smp["raw"].copy_(full["raw"][sl]); smp["admin_mask"].copy_(full["admin_mask"][sl]); smp["census_idx"].copy_(full["census_idx"][sl])
        smp["y"].copy_(full["y"][sl])
        torch.manual_seed(5)
        tr.step(smp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps - 1):
            tr.step(smp)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / max(a.steps - 1, 1) * 1e3
        form = None if tr._graphs is None else ("one" if len(tr._graphs[3]) == 1 else "split")
        return tr.flat_p.clone(), ms, form
    ab = {}
    ref = None
    ref, ms1, _ = trajectory(False, True)               # every rank computes it (same seeds): no broadcast needed
    for og in (0, 1):
        p, ms, form = trajectory(bool(og), False)
        d = ((p - ref).abs().max() / ref.abs().max()).item()
        t = torch.tensor([d, ms], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ab[f"POPCORN_DP_ONE_GRAPH={og}"] = {"graph_form": form, "max_rel_param_distance_to_single_process": t[0].item(), "ms_per_step": round(t[1].item(), 4),
                                             "finite": bool(torch.isfinite(p).all().item())}
    out["dp_graph_ab"] = ab
    out["single_process_ms_per_step"] = round(ms1, 4)
    out["steps"] = a.steps
    out["ok"] = bool(out["ranks_seen"] == world and all(v["finite"] and v["max_rel_param_distance_to_single_process"] <= 1e-3 for v in ab.values()))
    if world > 1:
        dist.barrier()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    sys.exit(0 if out["ok"] else 4)


if __name__ == "__main__":
    main()
