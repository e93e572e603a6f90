import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_terminal_summary(terminalreporter):
    """Every gradient comparison that passed through the tie adjudication instead of the flat 2e-4 bar (tests/tie_adjudication.py):
    which test, how many decisions differed, and how far either fp32 side is from the fp64 oracle."""
    try:
        from tests.tie_adjudication import ADJUDICATED
    except Exception:
        return
    tr = terminalreporter
    tr.write_sep("-", f"tie adjudications: {len(ADJUDICATED)} comparison(s) passed through a proven decision flip")
    for r in ADJUDICATED:
        tr.write_line(f"  {r['test']}: worst {r['worst']:.2e}, flips {r['flips']}, w_hip {r['w_hip']:.2e}, w_ref {r['w_ref']:.2e}")
    try:
        from tests.tie_adjudication import SHARED
    except Exception:
        return
    if SHARED:
        tr.write_sep("-", f"shared decisions: {len(SHARED)} gradient comparison(s) against the fp64 oracle on the HIP side of every ReLU / arg-max decision, "
                          f"worst residual {max(r['residual'] for r in SHARED):.2e} (bar 1e-4)")
        for r in SHARED:
            tr.write_line(f"  {r['test']}: {r['residual']:.2e} ({r['tensor']}), differing sites {r['flips']}")
