"""Pins the CPU oracle (oracle/preissmann_oracle.py) to the golden vectors that
oracle/gen_golden.py produced by running the reference itself.

Tolerance: max |dh|/max(|h|,1e-3) and |dQ|/max(|Q|,1) <= 1e-8 with identical
per-step Newton iteration counts.  The oracle calls the same SuperLU routine as the reference,
so it actually lands within 1e-11.
"""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, fixture_paths
from oracle import preissmann_oracle as O

FIXTURES = fixture_paths()
TOL = 1e-8


def rel_err(got, want, floor):
    return float(np.max(np.abs(got - want) / np.maximum(np.abs(want), floor)))


def members(meta):
    return list(range(meta["B"])) if meta.get("B") else [None]


def pick(fx, name, mem, base_ndim):
    a = fx[name]
    return a[mem] if mem is not None and a.ndim > base_ndim else a


def test_fixtures_present():
    names = {os.path.basename(p) for p in FIXTURES}
    for need in ("akbari.npz", "example.npz", "gerd.npz", "gerd_ensemble.npz", "synthetic_rect_64.npz",
                 "synthetic_rect_512.npz", "synthetic_trap_64.npz", "bc_stage_fixed.npz",
                 "bc_trap_poly.npz", "bc_compound_normal.npz", "c3_4096.npz", "c5_512.npz", "gerd_full.npz",
                 "bc_us_fixed_ds_flow.npz", "bc_us_rating_ds_stage.npz", "bc_us_normal_ds_stage.npz"):
        assert need in names


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_first_iteration_residual_and_jacobian(path):
    """R[2N] and the 8N-4 Jacobian entries of step 1 / iteration 1 (preissmann.py:61-99)."""
    fx, meta = O.load_fixture(path)
    if "R0" not in fx.files:
        pytest.skip("ensemble fixture keeps only hydrographs")
    for mem in members(meta):
        p = O.problem_from_fixture(fx, meta, mem)
        store = {"Y_prev": None} if p.ds.storage is not None else None
        R, data, _ = O.assemble(p, p.h0, p.Q0, p.h0, p.Q0, 1, store)
        R0 = pick(fx, "R0", mem, 1)
        assert R.shape == R0.shape
        scale = max(1.0, float(np.max(np.abs(R0))))
        assert np.max(np.abs(R - R0)) <= 1e-10 * scale
        if "J0" not in fx.files:
            continue                     # benchmark-size fixtures keep the residual vector only
        J0 = pick(fx, "J0", mem, 1)
        assert data.shape == J0.shape
        # polyline nodes: dA/dh and dR/dA are central differences with dh = 1e-6 (cross_section.py:523-538),
        # their cancellation noise (~1e-10 relative, summation order) enters the Jacobian
        assert rel_err(data, J0, 1e-6) <= (2e-8 if "geo_irr_npts" in fx.files else 1e-9)


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_hydrographs_and_iteration_counts(path):
    """Full depth/flow[nt,N] history and Newton counts (preissmann.py:101-177)."""
    fx, meta = O.load_fixture(path)
    for mem in members(meta):
        p = O.problem_from_fixture(fx, meta, mem)
        out = O.newton_run(p)
        d, f, it = pick(fx, "depth", mem, 2), pick(fx, "flow", mem, 2), pick(fx, "iters", mem, 1)
        assert out["status"] == 0
        assert rel_err(out["depth"], d, 1e-3) <= TOL
        assert rel_err(out["flow"], f, 1.0) <= TOL
        assert np.array_equal(out["iters"], it)


def test_known_answers_from_survey():
    """Spot values quoted in SURVEY.md section 8c (akbari outflow, example negative outflow)."""
    fx, meta = O.load_fixture(os.path.join(GOLDEN, "akbari.npz"))
    out = O.newton_run(O.problem_from_fixture(fx, meta))
    assert abs(out["flow"][8, -1] - 294.1373705485) < 1e-8
    assert abs(out["depth"][20, -1] - 0.8687656972) < 1e-9
    fx, meta = O.load_fixture(os.path.join(GOLDEN, "example.npz"))
    out = O.newton_run(O.problem_from_fixture(fx, meta))
    assert abs(out["flow"][1, -1] - (-999.9613412523)) < 1e-7
    assert abs(out["depth"][24, -1] - 69.9641357393) < 1e-8
    assert abs(out["storage_stage"][0] - 8.7019590909) < 1e-8
    assert np.allclose(out["storage_stage"], fx["storage_stage"][:, 1], rtol=1e-10, atol=0)


def test_gerd_rating_curve_probe():
    """The flattened Roseires blend reproduces the reference's discharge / dQ_dz samples."""
    fx, meta = O.load_fixture(os.path.join(GOLDEN, "gerd.npz"))
    p = O.problem_from_fixture(fx, meta)
    q = np.array([O.rating_discharge(p.ds, s) for s in fx["rating_probe_stage"]])
    dq = np.array([O.rating_dQ_dz(p.ds, s) for s in fx["rating_probe_stage"]])
    assert rel_err(q, fx["rating_probe_Q"], 1.0) <= 1e-9
    assert rel_err(dq, fx["rating_probe_dQ"], 1.0) <= 1e-6      # central difference of a ~1e4 value
