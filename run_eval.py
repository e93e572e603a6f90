#!/usr/bin/env python3
"""Counterpart of the reference's run_eval.py (same flags): ensemble sliding-window inference, census aggregation,
dasymetric adjustment, metrics -- on a synthetic raster."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from popcorn_amd.cli import run_eval  # noqa: E402

if __name__ == "__main__":
    run_eval()
