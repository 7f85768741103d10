import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests must never be skipped silently on a GPU box: they are only
    skipped when no HIP device is visible at all (the CPU container)."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    """The largest error every attention parity check measured, per kernel / route (tests/parity_tolerance.py)."""
    import parity_tolerance as tolerance  # tests/ is on sys.path (pytest's rootdir-relative imports)
    if not tolerance.MEASURED:
        return
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_errors.json"), "w") as f:
        json.dump(dict(bar="1e-3 * max(1, |ref|) (+ 2^-8 |ref| for 16-bit outputs)", measured=tolerance.MEASURED), f,
                  indent=1, sort_keys=True)
    print("\nattention parity, largest error per kernel / route (fraction of the bar, absolute):")
    for what, m in sorted(tolerance.MEASURED.items()):
        print(f"  {what:58s} {m['max_frac_of_bar']:.3f}  {m['max_abs']:.2e}  ({m['checks']} checks)")


@pytest.fixture(scope="session")
def golden_masks():
    with open(os.path.join(GOLDEN, "masks.json")) as f:
        return json.load(f)["cases"]


@pytest.fixture(scope="session")
def golden_selection():
    with open(os.path.join(GOLDEN, "selection.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_attention():
    with open(os.path.join(GOLDEN, "attention.json")) as f:
        meta = json.load(f)
    arrays = np.load(os.path.join(GOLDEN, "attention.npz"))
    return meta, arrays
