"""Sliding-window inference throughput (BASELINE config 5 shape: 2048x2048 windows, overlap 128, one season) on one GPU.

    python3 tools/bench_eval.py [H W [members [fp32|bf16]]]
"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from popcorn_amd.eval import evaluate_raster, get_patch_indices
from popcorn_amd.model import Args, get_model_kwargs, model_dict

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 5888)
members = int(sys.argv[3]) if len(sys.argv) > 3 else 1
prec = sys.argv[4] if len(sys.argv) > 4 else "fp32"
dev = torch.device("cuda")
margs = Args(Sentinel1=True, NIR=True, Sentinel2=True, feature_extractor="DDA", occupancymodel=True, pretrained=True,
             biasinit=0.9407, sentinelbuildings=True)
torch.manual_seed(1600)
models = [model_dict["POPCORN"](**get_model_kwargs(margs, "POPCORN")).to(dev).set_precision(prec) for _ in range(members)]
raster = torch.randn(1, 6, H, W, device=dev)
nwin = get_patch_indices(H, W).shape[0]
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mean, std, smean, sstd = evaluate_raster(models, raster)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{prec} {H}x{W}, {members} member(s): {nwin} windows in {dt * 1e3:.1f} ms = {nwin / dt:.1f} windows/s, "
          f"{H * W / dt / 1e6:.1f} Mpx/s of map, {nwin * 2048 * 2048 / dt / 1e6:.1f} Mpx/s through the network", flush=True)
print("finite:", bool(torch.isfinite(mean).all()))
