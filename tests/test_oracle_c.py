"""Pins the C restatement (oracle/preissmann_oracle.c, own banded LU) to the golden vectors the
reference produced, and cross-checks it against the numpy oracle on fresh inputs."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, fixture_paths
from oracle import c_oracle as CO
from oracle import preissmann_oracle as O

# the C restatement covers the trapezoid family with closed-form boundaries; polyline channels and the
# brentq-based general LumpedStorage are pinned by the numpy oracle (test_oracle_irregular.py, test_oracle_golden.py)
FIXTURES = [p for p in fixture_paths() if not os.path.basename(p).startswith(("irr_", "storage_curve_"))]


def rel_err(got, want, floor):
    return float(np.max(np.abs(got - want) / np.maximum(np.abs(want), floor)))


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_c_oracle_matches_reference_fixtures(path):
    fx, meta = O.load_fixture(path)
    for mem in (range(meta["B"]) if meta.get("B") else [None]):
        p = O.problem_from_fixture(fx, meta, mem)
        out = CO.run(p)
        pick = lambda k, nd: fx[k][mem] if mem is not None and fx[k].ndim > nd else fx[k]
        assert out["status"] == 0
        assert rel_err(out["depth"], pick("depth", 2), 1e-3) <= 1e-8
        assert rel_err(out["flow"], pick("flow", 2), 1.0) <= 1e-8
        assert np.array_equal(out["iters"], pick("iters", 1))


def test_c_and_numpy_oracles_agree_on_fresh_inputs():
    from synth import rect_problem
    for seed in (21, 22):
        p = rect_problem(700, seed=seed, n_steps=5)
        a, b = O.newton_run(p), CO.run(p)
        assert rel_err(b["depth"], a["depth"], 1e-3) <= 1e-10
        assert rel_err(b["flow"], a["flow"], 1.0) <= 1e-10
        assert np.array_equal(a["iters"], b["iters"])
