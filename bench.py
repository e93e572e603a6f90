#!/usr/bin/env python3
"""bench.py -- training throughput of the POPCORN hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched once per GPU by torch.distributed.run)

One "step" = one full optimisation step of the reference recipe (run_train.py:186-238) on a batch of synthetic
15-band 100x100 tiles already resident in HBM: band select + normalise -> frozen building-extractor U-Net ->
sparsity mask -> trainable dual-stream U-Net forward -> sparse head -> log-L1 loss + scale regulariser -> backward
(head, U-Net dgrad/wgrad) -> [N > 1: one RCCL all-reduce of the flat gradient] -> clip_grad_norm_(0.01) -> Adam.
fp32 throughout (fp32 MFMA).  Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (dominant
kernel, timed live with events) and, at N = 1, `cpu_baseline` (the CPU oracle = "port", timed on the host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_TRAIN_PER_TILE = 1_732_423_680          # SURVEY.md section 8d / BASELINE.md section 2 (all 10^4 px selected)
FP32_MATRIX_PEAK = 157.3e12                  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="tiles per GPU per step (BASELINE config: 64)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--cpu-iters", type=int, default=200)
    return ap.parse_args()


def dominant_kernel_roofline(torch, trainer, sample, reps=10):
    """The kernel with the largest share of the step (profiles/r1_*_kernel_stats.csv) is `head_bwd_pc_kernel`: the
    backward of the sparse 16-64-64-64-1 head (forward recompute + data gradients + weight gradients on fp32 MFMA).
    Timed live: `reps` back-to-back launches between two events on the launch stream, on the step's real tensors.
    Algorithmic flops per launch = 56,064 flop per selected pixel x selected pixels."""
    from popcorn_amd import ops
    m = trainer.model
    X = sample["input"]
    B, _, H, W = X.shape
    eng_u, eng_b = m.engines()
    with torch.no_grad():
        building = eng_b.building_score(X, m.p)
        feats, _ = eng_u.forward(X, 14, 14, H + 28, W + 28, save=False)
    g_pc = torch.ones(B, device=X.device)
    gsc = torch.full((1,), 1e-3, device=X.device)
    grads = [torch.empty_like(t) for t in m.head_tensors()]
    g_feat = torch.empty(B, 16, H + 28, W + 28, device=X.device)
    nsel = B * H * W                      # bench regions cover the tile: every pixel is selected

    def run():
        ops.head_bwd(feats, 14, 14, H, W, m.head_tensors(), building, mask=None, admin_mask=sample["admin_mask"],
                     census_idx=sample["census_idx"], g_popcount=g_pc, g_scale_const=gsc, grads=grads, g_feat=g_feat,
                     feat_bn=eng_u.feat_bn())
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    dur = e0.elapsed_time(e1) * 1e-3 / reps       # includes the 67 MB zero-fill kernel + the 8 us reduce kernel of the call
    flops = 56064.0 * nsel
    achieved = flops / dur / 1e12
    traffic = None                              # HBM bytes per launch from the committed PMC pass (cannot be read live)
    try:
        with open(os.path.join(ROOT, "profiles", "r1_pmc_head_bwd.json")) as fh:
            traffic = json.load(fh)["traffic_bytes"] if (B, H, W) == (64, 100, 100) else None
    except OSError:
        pass
    return {"bound": "mfma", "kernel": "head_bwd_pc_kernel (sparse head backward, producer/consumer waves, fp32 MFMA 16x16x4; + zero fill + reduce kernels of the same call)",
            "achieved": round(achieved, 3), "peak": FP32_MATRIX_PEAK / 1e12, "unit": "TFLOP/s",
            "frac": round(achieved * 1e12 / FP32_MATRIX_PEAK, 4), "traffic": traffic,
            "launch_us": round(dur * 1e6, 2), "alg_flop_per_launch": flops, "units_per_launch": nsel,
            "unit_def": "selected pixel, 56,064 flop"}


def conv_kernel_roofline(torch, B, reps=5, nsets=4):
    """Second roofline object: the most frequent heavy launch of the step, the grouped 3x3 conv 8->8 @128x128 (4 problems =
    2 networks x 2 streams, B tiles each), timed live with events over rotating buffer sets (cold Infinity Cache).  It sits
    at the ridge of this chip (18 flop / compulsory byte), so both views are given: algorithmic bytes against the HBM peak
    and algorithmic flops against the fp32-matrix peak."""
    from popcorn_amd import ops, _lib as L
    sets = []
    for _ in range(nsets):
        probs = []
        for _ in range(4):
            bias = torch.zeros(8, device="cuda")
            probs.append({"a": torch.randn(B, 8, 128, 128, device="cuda"), "w": torch.randn(8, 8, 3, 3, device="cuda") * 0.1,
                          "bn": L.bn(bias), "out": torch.empty(B, 8, 128, 128, device="cuda"), "_keep": bias})
        sets.append(probs)
    for s in sets:
        ops.conv3x3_fwd_group(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    with torch.cuda.stream(cap):
        with torch.cuda.graph(g, stream=cap):
            for _ in range(reps):
                for s in sets:
                    ops.conv3x3_fwd_group(s)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    dur = e0.elapsed_time(e1) * 1e-3 / (reps * nsets)
    nbytes = 4 * B * 128 * 128 * 4 * (8 + 8)                 # compulsory: read 8 channels, write 8 channels, fp32
    flops = 4 * B * 128 * 128 * 2 * 9 * 8 * 8
    return {"bound": "hbm", "kernel": "conv3x3_mfma_kernel<8,8,fwd> grouped x4 (3x3 conv + BN + ReLU, 8->8 @128x128)",
            "achieved": round(nbytes / dur / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(nbytes / dur / 8e12, 4),
            "traffic": None, "launch_us": round(dur * 1e6, 2), "alg_bytes_per_launch": nbytes,
            "mfma_view": {"achieved_tflops": round(flops / dur / 1e12, 2), "peak": FP32_MATRIX_PEAK / 1e12,
                          "frac": round(flops / dur / FP32_MATRIX_PEAK, 4)}}


def cpu_baseline(torch, sd, cpu_batch, iters):
    """The oracle (CPU restatement of the reference path, oracle/popcorn_oracle.py) timed on the host cores on a
    bounded sample of the same workload: `iters` train steps (fwd + loss + bwd + clip + Adam) at B=cpu_batch."""
    from oracle import popcorn_oracle as O
    from popcorn_amd.data.synthetic import make_raw_batch
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 32))          # oneDNN stops scaling (and oversubscription thrashes) beyond a few dozen threads
    torch.set_num_threads(cores)
    batch = make_raw_batch(cpu_batch, 100, 100, seed=1600)
    sample = {"input": O.select_normalize(batch["raw"]), "admin_mask": batch["admin_mask"],
              "census_idx": batch["census_idx"], "y": batch["y"]}
    params, state = dict(sd), {}

    def one():
        loss, out, grads, _ = O.train_step_grads(params, dict(sample))
        _, clipped = O.clip_grad_norm(grads, 0.01)
        params.update(O.adam_step(params, clipped, state, lr=1e-4, weight_decay=1e-5))

    t0 = time.perf_counter()
    one()                                   # warm-up (also sizes the bounded sample)
    t_warm = time.perf_counter() - t0
    budget = 15.0                           # seconds of CPU work for the timed sample
    n = max(1, min(iters, int(budget / max(t_warm, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(n):
        one()
    dt = time.perf_counter() - t0
    return {"value": round(cpu_batch * n / dt, 2), "unit": "patches/s", "cores": cores, "kind": "port",
            "sample": f"{n} train steps x B={cpu_batch} synthetic 100x100 tiles (oracle = torch-CPU fp32 restatement, "
                      f"{cores} threads of {avail} available), {dt:.1f} s"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from popcorn_amd.distributed import FlatReducer, init_from_env
    from popcorn_amd.data import stats
    from popcorn_amd.data.synthetic import make_raw_batch
    from popcorn_amd import ops
    from popcorn_amd.model import Args, get_model_kwargs, model_dict
    from popcorn_amd.train import FusedTrainStep

    rank, local_rank, world = init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path)")
    local_rank %= max(1, torch.cuda.device_count())     # (functional runs with more ranks than GPUs share a device)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    B = args.batch

    # rwa recipe (README.md:187): -S2 -NIR -S1 -occmodel -senbuilds -pret -wd 1e-5 --biasinit 0.9407
    margs = Args(Sentinel1=True, NIR=True, Sentinel2=True, feature_extractor="DDA", occupancymodel=True, pretrained=True,
                 biasinit=0.9407, sentinelbuildings=True)
    torch.manual_seed(1600)
    model = model_dict["POPCORN"](**get_model_kwargs(margs, "POPCORN")).to(dev)
    sd_cpu = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    trainer = FusedTrainStep(model, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01, loss=("log_l1_loss",), lam=(1.0,),
                             scale_regularization=0.01, lam_weak=100.0, reducer=FlatReducer(),
                             use_graph=not args.no_graph)
    batch = make_raw_batch(B, 100, 100, seed=1600 + rank, device=dev)           # resident in HBM before timing
    # the loader side of the step writes into the tensors the captured graph reads (no per-step input copies)
    sample = trainer.static_buffers(B, 100, 100) if not args.no_graph else \
        {"input": torch.empty(B, 6, 100, 100, device=dev), "admin_mask": torch.empty(B, 100, 100, device=dev),
         "census_idx": torch.empty(B, dtype=torch.int64, device=dev), "y": torch.empty(B, device=dev)}
    sample["admin_mask"].copy_(batch["admin_mask"])
    sample["census_idx"].copy_(batch["census_idx"])
    sample["y"].copy_(batch["y"])

    def step():
        ops.select_normalize(batch["raw"], stats.BAND6, stats.MEAN6, stats.STD6, out=sample["input"])
        return trainer.step(sample)

    torch.manual_seed(1600 + rank)
    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    loss_val = float(loss[0].item())
    if not (loss_val == loss_val) or abs(loss_val) == float("inf"):
        raise SystemExit(f"non-finite loss {loss_val}")

    if rank == 0:
        tiles = B * world * args.steps
        value = tiles / dt
        res = {
            "metric": "training patches/s (15-band 100x100)", "value": round(value, 1), "unit": "patches/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"config[1]/[2]: batch={B} synthetic S1+S2 100x100 tiles per GPU, full train step "
                                   "(building extractor + DDA dual-stream U-Net + sparse head fwd/bwd, log-L1 loss, "
                                   "clip 0.01, Adam, rwa flags), every pixel of every tile selected",
                       "global_batch": B * world, "tile": "15x100x100 -> 6x100x100 (128x128 internal)",
                       "parallelism": f"dp{world}", "graph": not args.no_graph},
            "final_loss": round(loss_val, 6),
            "step_tflops": round(value * FLOP_TRAIN_PER_TILE / 1e12, 3),
            "step_frac_of_fp32_mfma_peak": round(value * FLOP_TRAIN_PER_TILE / (FP32_MATRIX_PEAK * world), 4),
        }
        res["roofline"] = dominant_kernel_roofline(torch, trainer, sample)
        res["roofline_conv"] = conv_kernel_roofline(torch, B)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(torch, sd_cpu, args.cpu_batch, args.cpu_iters)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
