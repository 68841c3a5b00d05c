"""The whole drop-in path on the GPU: case scripts written against the reference's API
(Channel / Boundary / ... / PreissmannSolver.run) reproduce the reference's depth/flow history.
Tolerance 1e-8 relative (north_star), identical Newton iteration counts."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import preissmann_oracle as O
import case_builders as CB
from flowsim_amd import PreissmannBatch, _abi as A

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


@pytest.mark.parametrize("name", sorted(CB.HOST_ROW_BUILDERS))
def test_plugins_without_a_device_form_run_through_host_rows(name):
    """The plugin contract is "any object with discharge(stage, time) / dQ_dz" (rating_curve.py:32-63, :132-147) and a
    LumpedStorage may carry any rating curve (lumped_storage.py:24-35).  Plugins without a device form - operated gates
    that move with time and their own history, a Python callable as reservoir outlet - run with their boundary row
    evaluated on the host before every Newton iteration (FS_BC_HOST_ROW, one kernel launch per iteration); the rest of
    the iteration is the same kernel.  Against fixtures the reference produced with the same plugins."""
    fx, meta = O.load_fixture(os.path.join(GOLDEN, name + ".npz"))
    solver, tol = CB.HOST_ROW_BUILDERS[name]()
    assert not solver.channel.downstream_boundary.has_device_form()
    solver.run(tolerance=tol, verbose=0)
    assert rel(solver.depth, fx["depth"], 1e-3) <= TOL
    assert rel(solver.flow, fx["flow"], 1.0) <= TOL
    assert np.array_equal(solver.iterations, fx["iters"])
    assert rel(solver.unknowns, fx["final_unknowns"], 1e-3) <= 1e-7
    if name == "gerd_gates":       # the gates did move: the outflow jumps by an order of magnitude and comes back
        q = solver.flow[:, -1]
        assert q[26] < 2200 and q[27] > 15000 and q[36] < 2200
    elif name == "gerd_gates_long":
        # 2 409 nodes: more than a table kernel keeps on chip - the multi-pass kernel with its iteration budget (fs_long.hpp), one
        # Newton iteration per launch, the gate row evaluated by the mirror's RoseiresRatingCurve in between
        assert solver.number_of_nodes == 2409 and solver.kernel_entry["long_reach"] == 1 and solver.kernel_entry["boundary_class"] == -1
        q = solver.flow[:, -1]
        assert q[26] < 2200 and q[27] > 15000                      # the gates open here as they do at dx = 1 000 m
    else:
        st = solver.channel.downstream_boundary.lumped_storage
        got = np.array(st.stage_hydrograph)[1:, 1]
        assert rel(got, fx["storage_stage"][:, 1], 1e-3) <= TOL


def test_one_iteration_per_launch_equals_the_fused_loop(monkeypatch):
    """fs_batch_iterate (one Newton iteration per launch, iteration count carried across launches) against fs_batch_step
    (the whole loop in one launch) on device-evaluated boundaries, both through the instantiation fs_batch_iterate uses
    (boundary class -1): the same kernel code, so the same bits."""
    from fixture_batch import batch_from_problems
    for name, B in (("gerd", 1), ("storage_curve_poly_losses", 1), ("irr_mixed", 1), ("gerd_ensemble", 8)):
        fx, meta = O.load_fixture(os.path.join(GOLDEN, name + ".npz"))
        probs = [O.problem_from_fixture(fx, meta, m) for m in (range(B) if meta.get("B") else [None])]
        override = [float(p.geo["n_main"][0]) for p in probs] if B > 1 else None
        mode = "irregular" if name.startswith("irr") else "table"
        n = min(probs[0].nt - 1, 8)
        with batch_from_problems(probs, mode=mode, n_main_override=override) as a, \
                batch_from_problems(probs, mode=mode, n_main_override=override) as b:
            launches = 0
            while b.level < n:
                b.iterate()
                launches += 1
            monkeypatch.setenv("FS_KERNEL_INDEX", str(b.kernel_index()))
            a.step(n)
            monkeypatch.delenv("FS_KERNEL_INDEX")
            assert a.kernel_index() == b.kernel_index()
            its = a.iterations(0, n + 1)
            assert launches == int(its.max(axis=1).sum())          # members wait for the slowest one of a level
            assert np.array_equal(its, b.iterations(0, n + 1))
            assert np.array_equal(a.hydrographs(0, n + 1), b.hydrographs(0, n + 1))
            for x, y in zip(a.state() + a.guess() + a.history_arrays(0, n + 1), b.state() + b.guess() + b.history_arrays(0, n + 1)):
                assert np.array_equal(x, y)
            assert np.all(b.status() == 0)


def test_restart_continues_bit_exactly():
    """A run stopped at level k and continued in a NEW batch from (state, Newton start vector, reservoir stage) gives
    the bits of the uninterrupted single launch: after the first level the start vector differs from the state (SURVEY
    F2), so both travel (fs_batch_restart).  Chunked stepping of one batch likewise."""
    from fixture_batch import batch_from_problems
    for name in ("example", "synthetic_rect_512", "gerd", "storage_curve_power_trap", "irr_mixed", "synthetic_trap_64"):
        fx, meta = O.load_fixture(os.path.join(GOLDEN, name + ".npz"))
        probs = [O.problem_from_fixture(fx, meta, m) for m in (range(meta["B"]) if meta.get("B") else [None])]
        if name == "synthetic_trap_64":
            probs = probs[:1]                                   # TABLE geometry is shared by the batch
        nt = min(probs[0].nt, 13)
        k = nt // 2
        with batch_from_problems(probs, history=False) as a:
            a.step(nt - 1)
            want = a.hydrographs(0, nt), a.iterations(0, nt), a.state(), a.guess()
        with batch_from_problems(probs, history=False) as b:
            b.step(k)
            snap = b.state() + b.guess() + (b.storage_stage(),)
            b.step(nt - 1 - k)
            assert np.array_equal(b.hydrographs(0, nt), want[0]) and np.array_equal(b.iterations(0, nt), want[1])
        assert not np.array_equal(snap[0], snap[2])                 # state != Newton start vector
        with batch_from_problems(probs, history=False) as c:
            c.restart(k, *snap)
            assert c.level == k
            c.step(nt - 1 - k)
            assert np.all(c.status() == 0)
            assert np.array_equal(c.hydrographs(k, nt - k), want[0][k:])
            assert np.array_equal(c.iterations(k + 1, nt - 1 - k), want[1][k + 1:])
            for x, y in zip(c.state() + c.guess(), want[2] + want[3]):
                assert np.array_equal(x, y)


def test_bed_level_monte_carlo_in_one_launch_equals_members_run_singly():
    """SURVEY 8b per-reach geometry: 4 096 members of cases/gerd_roseires, each with its own perturbed bed profile and
    bankfull depths (fs_batch_set_geometry_table_per_reach), in ONE launch; eight of them run alone - through the shared-table
    entry point, as a batch of one - give the same bits."""
    from fixture_batch import boundary_spec
    fx, meta = O.load_fixture(os.path.join(GOLDEN, "gerd.npz"))
    p = O.problem_from_fixture(fx, meta)
    B, K = 4096, 4
    rng = np.random.default_rng(20260304)
    x = np.linspace(0.0, 1.0, p.N)
    bump = 0.15 * np.sin(2 * np.pi * (rng.uniform(0.5, 3.0, (B, 1)) * x[None] + rng.uniform(0, 1, (B, 1)))) * rng.uniform(0, 1, (B, 1))
    geo = {k: np.broadcast_to(p.geo[k], (B, p.N)).copy() for k in A.GEO_ROWS}
    geo["z_bed"] += bump
    geo["h_bf"] *= 1.0 + 0.1 * rng.uniform(-1, 1, (B, 1))
    geo["b_main"] *= 1.0 + 0.05 * rng.uniform(-1, 1, (B, 1))

    def run(tables, n):
        with PreissmannBatch(n, p.N, K + 1, section_mode="table") as b:
            b.set_scheme(p.theta, p.dt, p.dx, p.tol, p.max_iter)
            b.set_geometry_table(tables)
            b.set_boundary(A.UPSTREAM, boundary_spec(p.us, p.nt)); b.set_boundary(A.DOWNSTREAM, boundary_spec(p.ds, p.nt))
            b.set_state(np.broadcast_to(p.h0, (n, p.N)), np.broadcast_to(p.Q0, (n, p.N)))
            b.step(K)
            return b.hydrographs(0, K + 1), b.iterations(0, K + 1), b.state(), b.status(), b.kernel_index()
    hyd, its, (h, Q), st, kidx = run(geo, B)
    assert np.all(st == 0)
    assert np.ptp(hyd[K, 0, :]) > 1e-4                                       # the members do differ
    for r in (0, 1, 7, 63, 64, 1000, 4094, 4095):
        hyd1, its1, (h1, Q1), st1, kidx1 = run({k: geo[k][r] for k in A.GEO_ROWS}, 1)
        assert kidx1 == kidx and st1[0] == 0
        assert np.array_equal(hyd1[:, :, 0], hyd[:, :, r]) and np.array_equal(its1[:, 0], its[:, r])
        assert np.array_equal(h1[0], h[r]) and np.array_equal(Q1[0], Q[r])


def test_restart_guards():
    """what fs_batch_restart refuses or restricts: a storage boundary without the reservoir stage of the restart level (the
    run would silently go on from stage 0), and - for a batch that keeps a history - ranges that begin before the restart
    level (those rows were not restored; amplitudes then refer to the restart state)"""
    from fixture_batch import batch_from_problems
    from flowsim_amd._abi import FlowsimError
    fx, meta = O.load_fixture(os.path.join(GOLDEN, "example.npz"))              # fixed depth behind a LumpedStorage
    p = O.problem_from_fixture(fx, meta)
    k, nt = 5, 12
    with batch_from_problems([p], history=True) as a:
        a.step(nt - 1)
        want_h, want_Q = a.history_arrays(0, nt)
        want = a.derive(k, nt - k)
    with batch_from_problems([p], history=True) as b:
        b.step(k)
        snap = b.state() + b.guess()
        stage = b.storage_stage()
    with batch_from_problems([p], history=True) as c:
        with pytest.raises(FlowsimError, match="reservoir stage"):
            c.restart(k, *snap)                                                   # storage stage missing
        c.restart(k, *snap, stage)
        c.step(nt - 1 - k)
        assert np.all(c.status() == 0)
        h, Q = c.history_arrays(k, nt - k)
        assert np.array_equal(h, want_h[k:]) and np.array_equal(Q, want_Q[k:])    # the continued history, bit for bit
        with pytest.raises(FlowsimError, match="restarted at level"):
            c.history_arrays(0, nt)
        with pytest.raises(FlowsimError, match="restarted at level"):
            c.derive(0, nt)
        got = c.derive(k, nt - k)
        for name in ("level", "area", "top_width", "froude_number", "velocity", "wave_celerity"):
            assert np.array_equal(got[name], want[name]), name
        # amplitudes refer to the restart state (row 0 of the restarted batch), not to the initial condition
        np.testing.assert_allclose(got["amplitude"], want_h[k:] - want_h[k][None], rtol=0, atol=1e-12)


def test_a_new_state_closes_a_level_left_open_by_iterate():
    """fs_batch_iterate leaves its level open until every reach has accepted it; both state setters start over"""
    from fixture_batch import batch_from_problems
    fx, meta = O.load_fixture(os.path.join(GOLDEN, "akbari.npz"))              # steady-state initial condition: one depth, one flow
    p = O.problem_from_fixture(fx, meta)
    with batch_from_problems([p], mode="table") as b:
        assert b.iterate() == 1                                       # one Newton iteration of level 1: still open
        with pytest.raises(Exception, match="fs_batch_iterate"):
            b.step(1)
        b.set_state_uniform(float(p.h0[0]), float(p.Q0[0]))
        b.step(1)                                                     # accepted again
        b.iterate()
        b.set_state(p.h0[None], p.Q0[None])
        b.step(2)
        assert b.level == 2 and np.all(b.status() == 0)


def test_calibration_rmse_curve_matches_the_reference():
    """cases/gerd_roseires/n_calibrate - the ten-member Manning-n study as ONE device batch - against
    the curve the reference's own loop produced (n_calibrate.py:55-67 over model.run; tests/golden/rmse_curve.npz;
    SURVEY F9 quotes 5.85 / 2.65 / 1.74 at n = 0.02 / 0.0422 / 0.06)."""
    from cases.gerd_roseires.n_calibrate import rmse_curve, H_target, Q_gauge
    fx = np.load(os.path.join(GOLDEN, "rmse_curve.npz"))
    assert np.array_equal(H_target, fx["H_target"]) and np.array_equal(Q_gauge, fx["Q"])
    got = np.array(rmse_curve(fx["n_values"]))
    np.testing.assert_allclose(got, fx["rmse"], rtol=1e-8, atol=0)
    assert abs(got[0] - 5.85) < 5e-3 and abs(got[5] - 2.65) < 5e-3 and abs(got[-1] - 1.74) < 5e-3


@pytest.mark.parametrize("name", ["akbari", "example"])
def test_result_summary_text_matches_the_reference(name, tmp_path):
    """Solver.save_results' text summary (solver.py:188-233): the .txt the mirror writes next to its result tables is,
    character for character, the one the reference wrote for the same case (tests/golden/result_summaries.npz)."""
    fx = np.load(os.path.join(GOLDEN, "result_summaries.npz"))
    solver, tol = CB.BUILDERS[name]()
    solver.run(tolerance=tol, verbose=0)
    solver.save_results(str(tmp_path), "results.xlsx")
    assert open(os.path.join(tmp_path, "results.txt")).read() == str(fx[name])


def test_derived_fields_can_stay_on_the_device():
    """fs_batch_derive_device: the prepare_results kernel with its outputs left in HBM (buffers owned and reused by the
    handle); the host-copy entry point is the same kernel plus the download."""
    import ctypes
    from fixture_batch import batch_from_problems
    fx, meta = O.load_fixture(os.path.join(GOLDEN, "akbari.npz"))
    p = O.problem_from_fixture(fx, meta)
    with batch_from_problems([p]) as b:
        b.step(p.nt - 1)
        ptr1 = b.derive_device(0, p.nt, fields=2 | 8)             # area, Froude number
        assert set(ptr1) == {"area", "froude_number"} and all(ptr1.values())
        ptr2 = b.derive_device(0, p.nt, fields=2 | 8)
        assert ptr1 == ptr2                                      # no reallocation for a call of the same size
        host = b.derive(0, p.nt, fields=("area", "froude_number"))
        assert rel(host["area"][:, 0], fx["derived_area"], 1e-6) <= TOL
        assert rel(host["froude_number"][:, 0], fx["derived_froude_number"], 1e-6) <= TOL


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("name", ["akbari", "example", "gerd"])
def test_derived_fields_in_batches_of_any_size(name, dtype):
    """the post-processing kernel moves 16 bytes per access (2 doubles / 4 floats per thread) when the batch's B*N
    elements allow it and falls back to single elements otherwise: B = 1 .. 4 copies of a reach (N = 30, 21, 121, so both
    paths and the ragged last thread occur in either precision) give, reach by reach, the bits of the batch of one, and
    that one the reference's prepare_results arrays"""
    from fixture_batch import batch_from_problems
    if dtype == "f32" and name != "akbari":
        pytest.skip("fp32: the rectangular case only (N = 30: B*N % 4 is 2, 0, 2, 0)")
    fx, meta = O.load_fixture(os.path.join(GOLDEN, name + ".npz"))
    p = O.problem_from_fixture(fx, meta)
    nlev = 6
    fields = ("level", "area", "top_width", "froude_number", "velocity", "wave_celerity", "amplitude", "peak_amplitude")
    ref = None
    for B in (1, 2, 3, 4):
        with batch_from_problems([p] * B, dtype=dtype) as b:
            b.step(nlev - 1)
            assert np.all(b.status() == 0)
            d = b.derive(0, nlev)
        assert set(d) == set(fields)
        if ref is None:
            ref = d
            if dtype == "f64":
                for f in ("area", "froude_number", "velocity"):
                    assert rel(d[f][:, 0], fx["derived_" + f][:nlev], 1e-6) <= TOL, f
        for f in fields:
            for j in range(B):
                got = d[f][j] if f == "peak_amplitude" else d[f][:, j]
                want = ref[f][0] if f == "peak_amplitude" else ref[f][:, 0]
                assert np.array_equal(got, want), (f, B, j)


def test_batches_on_two_devices_in_one_process():
    """every entry point switches to its batch's device and back: a batch on device 1 configured and stepped
    while device 0 is current, next to a batch on device 0"""
    from flowsim_amd import _abi as A
    if A.device_count() < 2:
        pytest.skip("one GPU visible")
    from fixture_batch import batch_from_problems
    import fixture_batch as FB
    from flowsim_amd import PreissmannBatch
    fx, meta = O.load_fixture(os.path.join(GOLDEN, "akbari.npz"))
    p = O.problem_from_fixture(fx, meta)
    real = PreissmannBatch.__init__
    out = []
    for dev in (0, 1):
        def init(self, *a, **k):
            k["device"] = dev
            real(self, *a, **k)
        FB.PreissmannBatch.__init__ = init
        try:
            b = batch_from_problems([p])
        finally:
            FB.PreissmannBatch.__init__ = real
        out.append(b)
    for b in out:
        b.step(p.nt - 1)
    for b in out:
        h, Q = b.history_arrays()
        assert rel(Q[:, 0], fx["flow"], 1.0) <= TOL
        b.close()
