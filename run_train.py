#!/usr/bin/env python3
"""Counterpart of the reference's run_train.py (same flags): python run_train.py -S2 -NIR -S1 -occmodel -senbuilds -pret
-wd 1e-5 --biasinit 0.9407   (rwa recipe, README.md:187), on the synthetic PopulationDataset-shaped loader."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from popcorn_amd.cli import run_train  # noqa: E402

if __name__ == "__main__":
    run_train()
