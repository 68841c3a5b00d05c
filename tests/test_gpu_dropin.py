"""The whole drop-in path on the GPU: case scripts written against the reference's API
(Channel / Boundary / ... / PreissmannSolver.run) reproduce the reference's depth/flow history.
Tolerance 1e-8 relative (north_star), identical Newton iteration counts."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import preissmann_oracle as O
import case_builders as CB

pytestmark = pytest.mark.gpu
TOL = 1e-8


def rel(got, want, floor):
    return float(np.max(np.abs(got - want) / np.maximum(np.abs(want), floor)))


@pytest.mark.parametrize("name", sorted(CB.BUILDERS))
def test_case_through_reference_api(name):
    fx, meta = O.load_fixture(os.path.join(GOLDEN, name + ".npz"))
    solver, tol = CB.BUILDERS[name]()
    solver.run(tolerance=tol, verbose=0)
    assert solver.depth.shape == (meta["nt"], meta["N"])
    assert rel(solver.depth, fx["depth"], 1e-3) <= TOL
    assert rel(solver.flow, fx["flow"], 1.0) <= TOL
    assert np.array_equal(solver.iterations, fx["iters"])
    assert rel(solver.unknowns, fx["final_unknowns"], 1e-3) <= 1e-7     # Newton vector seeding the next level
    # derived fields of prepare_results exist and are consistent
    assert solver.level.shape == solver.depth.shape and np.all(solver.area > 0)
    np.testing.assert_allclose(solver.velocity * solver.area, solver.flow, rtol=1e-12)


def test_example_storage_stage_hydrograph():
    fx, meta = O.load_fixture(os.path.join(GOLDEN, "example.npz"))
    solver, tol = CB.example()
    solver.run(tolerance=tol, verbose=0)
    st = solver.channel.downstream_boundary.lumped_storage
    got = np.array(st.stage_hydrograph)[1:, 1]                 # [0] is the level-0 entry prepare_results inserts
    assert rel(got, fx["storage_stage"][:, 1], 1e-3) <= TOL
    assert abs(solver.flow[1, -1] - (-999.9613412523)) < 1e-6   # the reference's negative first outflow


def test_synthetic_trapezoid_power_rating():
    fx, meta = O.load_fixture(os.path.join(GOLDEN, "synthetic_trap_64.npz"))
    for mem in range(meta["B"]):
        solver, tol = CB.synthetic_trap(mem)
        solver.run(tolerance=tol, verbose=0)
        assert rel(solver.depth, fx["depth"][mem], 1e-3) <= TOL
        assert rel(solver.flow, fx["flow"][mem], 1.0) <= TOL
        assert np.array_equal(solver.iterations, fx["iters"][mem])


def test_non_convergence_raises_like_the_reference(capsys):
    solver, tol = CB.akbari()
    with pytest.raises(ValueError, match="Convergence within 1 iterations couldn't be achieved."):
        solver.run(tolerance=1e-12, verbose=0, max_iter=1)
    assert "subcritical" in capsys.readouterr().out              # check_criticality ran first


def test_manning_ensemble_batch_matches_sequential_reference_runs():
    """BASELINE configs[3] semantics at small size: 8 members of the gerd_roseires n-study in one
    device batch == 8 separate runs of the reference (tests/golden/gerd_ensemble.npz)."""
    from cases.gerd_roseires.n_calibrate import member_setup
    from flowsim_amd.ensemble import run_manning_ensemble
    from cases.gerd_roseires import settings as S
    fx, meta = O.load_fixture(os.path.join(GOLDEN, "gerd_ensemble.npz"))
    ns = fx["n_members"]
    solvers = [member_setup(float(n))[0] for n in ns]
    lead = solvers[0]
    lead.channel.member_ics = np.stack([s.channel.initial_conditions for s in solvers])
    res = run_manning_ensemble(lead, ns, tolerance=S.tolerance)
    assert np.all(res["status"] == 0)
    for i in range(len(ns)):
        hy = res["hydrographs"][:, :, i]
        for col, (arr, node, floor) in enumerate(((fx["depth"], 0, 1e-3), (fx["flow"], 0, 1.0),
                                                  (fx["depth"], -1, 1e-3), (fx["flow"], -1, 1.0))):
            assert rel(hy[:, col], arr[i][:, node], floor) <= TOL, (i, col)
        assert np.array_equal(res["iterations"][:, i], fx["iters"][i])


@pytest.mark.parametrize("name", ["akbari", "example", "bc_compound_normal", "gerd", "irr_levee", "irr_mixed",
                                  "storage_curve_poly_losses", "storage_curve_power_trap", "storage_curve_closed"])
def test_derived_fields_kernel_matches_reference_post_processing(name):
    """fs_batch_derive (HIP, elementwise) vs the reference's Solver.prepare_results output."""
    fx, meta = O.load_fixture(os.path.join(GOLDEN, name + ".npz"))
    solver, tol = CB.BUILDERS[name]()
    solver.run(tolerance=tol, verbose=0)
    assert solver._derived is not None
    for field in ("level", "area", "top_width", "froude_number", "velocity", "wave_celerity"):
        assert rel(getattr(solver, field), fx["derived_" + field], 1e-6) <= TOL, field
    np.testing.assert_allclose(solver.amplitude, fx["derived_amplitude"], rtol=0, atol=1e-8 * np.abs(fx["depth"]).max())
    np.testing.assert_allclose(solver.peak_amplitude, fx["derived_peak_amplitude"], rtol=0, atol=1e-8 * np.abs(fx["depth"]).max())
    if "derived_storage_outflow" in fx.files:
        assert rel(solver.storage_stage, fx["derived_storage_stage"], 1e-3) <= TOL
        np.testing.assert_allclose(solver.storage_outflow, fx["derived_storage_outflow"], rtol=1e-6, atol=1e-4)


@pytest.mark.parametrize("name", ["akbari", "gerd", "example"])
def test_newton_residual_trace_matches_reference(name, capsys):
    """run(verbose=3) prints the reference's '>> Iteration #i: Error = ...' lines; the norms of every
    Newton iteration of every level follow the reference's (quadratic convergence amplifies
    rounding, so late iterates are compared against the size of the first residual)."""
    fx, meta = O.load_fixture(os.path.join(GOLDEN, name + ".npz"))
    solver, tol = CB.BUILDERS[name]()
    solver.run(tolerance=tol, verbose=3)
    out = capsys.readouterr().out
    assert out.count(">> Iteration #") == int(fx["iters"].sum())
    lv, val = fx["norm_level"], fx["norm_value"]
    for k in range(1, meta["nt"]):
        ref = val[lv == k]
        got = solver.residual_norms[k, :len(ref)]
        assert np.all(solver.residual_norms[k, len(ref):] == 0)
        np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-9 * ref[0] + 1e-12)
