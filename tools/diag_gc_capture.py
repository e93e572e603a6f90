"""Diagnostic: the GPU test suite with the captures' garbage-collector guard REPLACED by gc.DEBUG_COLLECTABLE (prints every object a
collection in the middle of a graph capture finds unreachable) -- what `train.py: _capturing` protects against.
    python tools/diag_gc_capture.py [pytest args]      (default: tests -m gpu -q -x)"""
import contextlib
import gc
import os
import sys


def main():
    sys.path.insert(0, os.getcwd())
    import torch
    import popcorn_amd.train as T

    @contextlib.contextmanager
    def raw_capturing(g, **kw):          # the pre-guard behaviour + the collector's own report
        with torch.cuda.graph(g, **kw):
            gc.set_debug(gc.DEBUG_COLLECTABLE)
            try:
                yield
            finally:
                gc.set_debug(0)

    T._capturing = raw_capturing
    import pytest
    return pytest.main(sys.argv[1:] or ["tests", "-m", "gpu", "-q", "-x"])


if __name__ == "__main__":
    sys.exit(main())
