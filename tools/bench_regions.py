"""bench.py's `config3_regions` leg alone (BASELINE config 3 at the reference's real geometry: weak_batch_size = 2 census regions of
varying size, eager steps through the native executor): one JSON object on stdout.

    python tools/bench_regions.py [--native 0|1]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.getcwd())
ap = argparse.ArgumentParser()
ap.add_argument("--native", default="1")
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
os.environ["POPCORN_NATIVE_STEP"] = a.native
import torch                                                  # noqa: E402
import bench                                                  # noqa: E402
from popcorn_amd.model import Args                            # noqa: E402

margs = Args(Sentinel1=True, NIR=True, Sentinel2=True, feature_extractor="DDA", occupancymodel=True, pretrained=True,
             biasinit=0.9407, sentinelbuildings=True)
res = bench.config3_regions_leg(torch, margs, torch.device("cuda:0"), steps=a.steps)
print(json.dumps(res))
