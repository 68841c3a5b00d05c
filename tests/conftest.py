"""pytest configuration: registers the `gpu` marker and puts the package + oracle on sys.path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "flow-sim_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def fixture_paths(host_evaluated=False):
    """tests/golden/*.npz that describe a solver run (geometry, boundaries, depth/flow history).  Left out: the two
    data-only fixtures (rmse_curve, result_summaries) and - unless asked for - the runs whose boundary is a Python
    plugin without a device form (gerd_gates, storage_callable_rc): the oracle cannot restate an arbitrary callable,
    those are pinned to the reference directly through the mirror API (tests/test_gpu_dropin.py)."""
    import glob
    import json
    import numpy as np
    out = []
    for path in sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))):
        meta = json.loads(str(np.load(path)["meta"]))
        if "kind" in meta or "cases" in meta:          # data-only fixtures; the multi-case sweep (tests/test_random_sweep.py)
            continue
        if ("host_evaluated" in meta) != host_evaluated:
            continue
        out.append(path)
    return out
