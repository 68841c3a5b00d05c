"""GPU parity: the HIP path (through the C ABI) against the golden vectors produced by the
reference and against the CPU oracle on the same inputs.

Tolerance (BASELINE.json north_star / SURVEY 8c): relative stage / discharge error
max |dh|/max(|h|,1e-3), |dQ|/max(|Q|,1) <= 1e-8 in fp64, identical Newton iteration counts.
"""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, fixture_paths
from oracle import preissmann_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-8
FIXTURES = fixture_paths()


def boundary_kind(bc):
    from fixture_batch import boundary_spec
    return boundary_spec(bc, 4).kind


def is_rect(p):
    from fixture_batch import is_rect_uniform
    return is_rect_uniform(p)


def rel_err(got, want, floor):
    return float(np.max(np.abs(got - want) / np.maximum(np.abs(want), floor)))


def problems_of(path):
    fx, meta = O.load_fixture(path)
    mem = list(range(meta["B"])) if meta.get("B") else [None]
    return fx, meta, [O.problem_from_fixture(fx, meta, m) for m in mem]


def golden(fx, name, i, B, base_ndim):
    a = fx[name]
    return a[i] if B and a.ndim > base_ndim else a


def run_and_compare(path, mode, shape=None, monkeypatch=None):
    from fixture_batch import batch_from_problems
    fx, meta, probs = problems_of(path)
    B = meta.get("B")
    if "geo_irr_npts" in fx.files and mode == "table":
        mode = "irregular"                     # polyline nodes: the TABLE mode cannot express them
    if shape:
        monkeypatch.setenv("FS_KERNEL_SHAPE", shape)
    override = None
    if os.path.basename(path).startswith("gerd_ensemble"):
        override = [float(p.geo["n_main"][0]) for p in probs]      # all sections share one n_main per member
        for p in probs:
            assert np.ptp(p.geo["n_main"]) < 1e-15
    # the TABLE geometry is shared by the whole batch: members with their own channel run one by one
    shared = override is not None or mode != "table" or all(
        all(np.array_equal(p.geo[k], probs[0].geo[k]) for k in O.GEO_KEYS) for p in probs)
    groups = [list(range(len(probs)))] if shared else [[i] for i in range(len(probs))]
    info = None
    for grp in groups:
        with batch_from_problems([probs[i] for i in grp], mode=mode, n_main_override=override) as b:
            b.step(probs[0].nt - 1)
            assert np.all(b.status() == 0), b.status()
            h, Q = b.history_arrays()
            its = b.iterations()
            hyd = b.hydrographs()
            for j, i in enumerate(grp):
                d, f, it = golden(fx, "depth", i, B, 2), golden(fx, "flow", i, B, 2), golden(fx, "iters", i, B, 1)
                assert rel_err(h[:, j], d, 1e-3) <= TOL, (path, i)
                assert rel_err(Q[:, j], f, 1.0) <= TOL, (path, i)
                assert np.array_equal(its[:, j], it), (path, i, its[:, j], it)
                assert np.array_equal(hyd[:, 0, j], h[:, j, 0]) and np.array_equal(hyd[:, 3, j], Q[:, j, -1])
            info = b.kernel_info()
    return info


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_golden_hydrographs_table_mode(path):
    _, meta = O.load_fixture(path)
    # the table kernels reach 2049 nodes; the one longer fixture is a rectangular prismatic channel (uniform-geometry mode)
    run_and_compare(path, "table" if meta["N"] <= 2049 else "rect_uniform")


@pytest.mark.parametrize("name", ["akbari", "example", "synthetic_rect_64", "synthetic_rect_512", "bc_stage_fixed"])
def test_golden_hydrographs_rect_fast_path(name):
    run_and_compare(os.path.join(GOLDEN, name + ".npz"), "rect_uniform")


@pytest.mark.parametrize("shape", ["2,1", "4,1", "8,1", "16,1", "16,2", "16,4"])
def test_every_kernel_shape_on_small_reach(shape, monkeypatch):
    """Same answers whatever the cells-per-lane / waves-per-reach split (padding, cross-wave fold)."""
    m, w = map(int, shape.split(","))
    for name in ("synthetic_rect_64", "akbari"):
        path = os.path.join(GOLDEN, name + ".npz")
        fx, meta = O.load_fixture(path)
        if 64 * m * w < meta["N"] - 1:
            continue
        info = run_and_compare(path, "rect_uniform", shape, monkeypatch)
        assert (info["cells_per_thread"], info["waves_per_reach"]) == (m, w)


@pytest.mark.parametrize("shape", ["2,1", "4,1", "8,1", "8,4"])
def test_table_mode_kernel_shapes(shape, monkeypatch):
    m, w = map(int, shape.split(","))
    for name in ("bc_compound_normal", "gerd"):
        path = os.path.join(GOLDEN, name + ".npz")
        fx, meta = O.load_fixture(path)
        if 64 * m * w < meta["N"] - 1:
            continue
        run_and_compare(path, "table", shape, monkeypatch)


@pytest.mark.parametrize("shape", ["2,1", "8,1", "8,4"])
def test_irregular_mode_kernel_shapes(shape, monkeypatch):
    """Polyline sections (IrregularSection): single thalweg, levee with sub-channel conveyance and
    composite roughness, trapezoid/polyline mix with curvature - in every kernel shape."""
    for name in ("irr_single", "irr_levee", "irr_mixed"):
        run_and_compare(os.path.join(GOLDEN, name + ".npz"), "irregular", shape, monkeypatch)


@pytest.mark.parametrize("name", ["irr_single", "irr_levee", "irr_mixed"])
def test_irregular_derived_fields(name):
    """fs_batch_derive on polyline nodes against the reference's Solver.prepare_results output."""
    from fixture_batch import batch_from_problems
    fx, meta, probs = problems_of(os.path.join(GOLDEN, name + ".npz"))
    with batch_from_problems(probs) as b:
        b.step(probs[0].nt - 1)
        d = b.derive(0, probs[0].nt)
    for f in ("area", "top_width", "froude_number", "velocity", "wave_celerity", "level"):
        assert rel_err(d[f][:, 0], fx["derived_" + f], 1e-6) <= TOL, f


def test_irregular_mode_rejects_bad_input():
    from flowsim_amd import PreissmannBatch
    from flowsim_amd._abi import FlowsimError
    with pytest.raises(FlowsimError, match="fp64 only"):
        PreissmannBatch(1, 8, 4, dtype="f32", section_mode="irregular")
    fx, meta, probs = problems_of(os.path.join(GOLDEN, "irr_single.npz"))
    geo = dict(probs[0].geo)
    with PreissmannBatch(1, probs[0].N, 4, section_mode="irregular") as b:
        bad = dict(geo); bad["irr_x"] = geo["irr_x"][:, ::-1].copy()
        with pytest.raises(FlowsimError, match="ascending|same shape"):
            b.set_geometry_irregular(bad)
        bad = dict(geo); bad["z_bed"] = geo["z_bed"] + 0.5
        with pytest.raises(FlowsimError, match="min\\(z\\)"):
            b.set_geometry_irregular(bad)


def test_oracle_agreement_on_fresh_inputs():
    """Seeded inputs that are not in any fixture: HIP path vs CPU oracle (rect channel, N=300)."""
    from fixture_batch import batch_from_problems
    from synth import rect_problem
    probs = [rect_problem(300, seed=s, n_steps=6) for s in range(5)]
    with batch_from_problems(probs) as b:
        b.step(6)
        h, Q = b.history_arrays()
        its = b.iterations()
    for i, p in enumerate(probs):
        out = O.newton_run(p)
        assert rel_err(h[:, i], out["depth"], 1e-3) <= TOL
        assert rel_err(Q[:, i], out["flow"], 1.0) <= TOL
        assert np.array_equal(its[:, i], out["iters"])


def test_batch_invariance_bitwise():
    """Reach i inside a mixed batch == reach i alone, a rerun, and chunked stepping == one launch: all bit for bit.
    (The two lanes that share a node move their copies of it by the same bits - the shared node's p comes from the same
    two numbers on both sides - so what a launch writes back at its end is exactly what the registers would have
    carried on.)"""
    from fixture_batch import batch_from_problems
    from synth import rect_problem
    for N in (1000, 4096, 300):
        probs = [rect_problem(N, seed=s, n_steps=4) for s in range(6)]
        with batch_from_problems(probs) as b:
            b.step(4)
            h_all, Q_all = b.history_arrays()
            it_all = b.iterations()
        with batch_from_problems(probs) as b:
            b.step(4)
            h_again, Q_again = b.history_arrays()
        assert np.array_equal(h_all, h_again) and np.array_equal(Q_all, Q_again)
        for i in (0, 3, 5):
            with batch_from_problems([probs[i]]) as b1:
                b1.step(1); b1.step(2); b1.step(1)
                h1, Q1 = b1.history_arrays()
                assert np.array_equal(b1.iterations()[:, 0], it_all[:, i])
            assert np.array_equal(h1[:, 0], h_all[:, i]) and np.array_equal(Q1[:, 0], Q_all[:, i])


def test_steady_state_is_a_fixed_point():
    """Uniform flow at normal depth with constant inflow: one Newton iteration per level, state unchanged
    (property (v))."""
    from fixture_batch import batch_from_problems
    from synth import rect_problem
    probs = [rect_problem(4096, seed=s, n_steps=3, steady=True) for s in range(3)]
    with batch_from_problems(probs, history=False) as b:
        b.step(3)
        h, Q = b.state()
        its = b.iterations()
        assert np.all(b.status() == 0)
    for i, p in enumerate(probs):
        assert np.max(np.abs(h[i] - p.h0)) <= 1e-9 * p.h0[0]
        assert np.max(np.abs(Q[i] - p.Q0)) <= 1e-9 * p.Q0[0]
    assert np.all(its[1:] == 1)


def test_full_width_reach_against_oracle():
    """N = 4096 (BASELINE configs[2] node count): 2 reaches x 2 levels against the oracle."""
    from fixture_batch import batch_from_problems
    from synth import rect_problem
    probs = [rect_problem(4096, seed=s, n_steps=2) for s in (11, 12)]
    with batch_from_problems(probs) as b:
        b.step(2)
        h, Q = b.history_arrays()
        its = b.iterations()
        info = b.kernel_info()
    assert info["waves_per_reach"] == 4 and info["cells_per_thread"] == 16
    for i, p in enumerate(probs):
        out = O.newton_run(p)
        assert rel_err(h[:, i], out["depth"], 1e-3) <= TOL
        assert rel_err(Q[:, i], out["flow"], 1.0) <= TOL
        assert np.array_equal(its[:, i], out["iters"])


def test_api_misuse_is_reported_not_computed():
    """Argument checks of the C ABI: the messages of the reference where it has them
    (boundary.py:33, :83-87), otherwise a plain description; nothing runs with a half-configured batch."""
    from flowsim_amd import BoundarySpec, PreissmannBatch
    from flowsim_amd import _abi as A
    import ctypes as C
    E = A.FlowsimError
    for bad in ((0, 30, 5), (1, 1, 5), (1, 30, 1), (1, 40000, 5)):        # (reaches beyond 32 768 nodes: no kernel, not even the multi-pass one)
        with pytest.raises(E, match="n_reaches >= 1|no kernel instantiation"):
            PreissmannBatch(*bad)
    with PreissmannBatch(2, 40, 6) as b:
        lib = b._lib
        with pytest.raises(E, match="must be set first"):
            b.step(1)
        with pytest.raises(E, match="dt, dx, tolerance > 0"):
            b.set_scheme(0.6, -1.0, 100.0, 1e-6, 10)
        b.set_scheme(0.6, 600.0, 250.0, 1e-6, 50)
        with pytest.raises(E, match="width and Manning n must be positive"):
            b.set_geometry_uniform([100.0, -1.0], [0.03, 0.03], [1.0, 1.0], [0.0, 0.0])
        with pytest.raises(E, match="another section_mode"):
            b.set_geometry_table({k: np.zeros(40) for k in A.GEO_ROWS})
        b.set_geometry_uniform([100.0, 120.0], [0.03, 0.03], [1.0, 1.0], [0.0, 0.0])
        assert lib.fs_batch_set_bc(b._h, 2, A.BC_FIXED_DEPTH, None, 0, 0, None) != 0        # bad side
        assert b"side" in lib.fs_last_error()
        assert lib.fs_batch_set_bc(b._h, A.UPSTREAM, 99, None, 0, 0, None) != 0
        assert lib.fs_last_error() == b"Invalid boundary condition."
        assert lib.fs_batch_set_bc(b._h, A.UPSTREAM, A.BC_FLOW_HYDROGRAPH, None, 0, 0, None) != 0   # no hydrograph
        assert lib.fs_last_error() == b"Insufficient arguments for boundary condition."
        p = (C.c_double * 5)(1.0e5, 1.0, 0.0, 50.0, 0.0)
        assert lib.fs_batch_set_bc(b._h, A.UPSTREAM, A.BC_STORAGE, p, 5, 0, None) != 0             # storage upstream
        assert b"downstream only" in lib.fs_last_error()
        b.set_boundary(A.UPSTREAM, BoundarySpec(A.BC_FLOW_HYDROGRAPH, {}, np.full(6, 200.0)))
        b.set_boundary(A.DOWNSTREAM, BoundarySpec(A.BC_FIXED_DEPTH, dict(initial_depth=2.0)))
        b.set_state_uniform([2.0, 2.0], [200.0, 200.0])
        with pytest.raises(E, match="n_steps must be >= 1"):
            b.step(0)
        with pytest.raises(E, match="past max_levels"):
            b.step(6)
        b.step(5)
        assert b.level == 5 and np.all(b.status() == 0)
        with pytest.raises(E, match="past max_levels"):
            b.step(1)
    with PreissmannBatch(1, 40, 6) as nb:
        with pytest.raises(E):
            nb.history_arrays()                    # no FS_FLAG_HISTORY on this batch
        with pytest.raises(E, match="FS_SEC_TABLE or FS_SEC_IRREGULAR"):
            nb.set_boundary(A.DOWNSTREAM, BoundarySpec(A.BC_HOST_ROW))       # host rows: table / polyline modes only
        with pytest.raises(E, match="no field requested|without FS_FLAG_HISTORY"):
            nb.derive_device(0, 2, fields=0)
    # the one-iteration-per-launch entry points (host-evaluated boundary rows)
    from fixture_batch import batch_from_problems
    fx, meta = O.load_fixture(os.path.join(GOLDEN, "bc_compound_normal.npz"))
    p = O.problem_from_fixture(fx, meta)
    with batch_from_problems([p], mode="table") as tb:
        with pytest.raises(E, match="not an FS_BC_HOST_ROW boundary"):
            tb.set_host_rows(A.DOWNSTREAM, 0.0, 1.0, 0.0)
        tb.set_boundary(A.DOWNSTREAM, BoundarySpec(A.BC_HOST_ROW))
        with pytest.raises(E, match="advances with fs_batch_iterate"):
            tb.step(1)
        with pytest.raises(E, match="level out of range"):
            tb.restart(p.nt, p.h0, p.Q0, p.h0, p.Q0)
        it = tb.boundary_iterate()
        assert it.shape == (4, 1) and it[0, 0] == p.h0[0] and it[3, 0] == p.Q0[-1]
    with batch_from_problems([p], mode="table") as tb:
        tb.iterate()                               # a level opened with iterate() must be closed with it
        if tb.level == 0:
            with pytest.raises(E, match="must be closed with it first"):
                tb.step(1)


def test_eight_wave_shape_on_a_full_width_reach(monkeypatch):
    """M = 8, W = 8 (two waves per SIMD; level-0 tree records kept in registers and handed to the
    pair lane by quad_perm in the down-sweep): not the default shape, selected here through the override."""
    from fixture_batch import batch_from_problems
    from oracle import c_oracle
    from synth import rect_problem
    monkeypatch.setenv("FS_KERNEL_SHAPE", "8,8")
    probs = [rect_problem(4096, seed=s, n_steps=2) for s in (21, 22)]
    with batch_from_problems(probs) as b:
        b.step(2)
        h, Q = b.history_arrays()
        its = b.iterations()
        info = b.kernel_info()
    assert (info["cells_per_thread"], info["waves_per_reach"]) == (8, 8)
    for i, p in enumerate(probs):
        out = c_oracle.run(p)
        assert rel_err(h[:, i], out["depth"], 1e-3) <= TOL and rel_err(Q[:, i], out["flow"], 1.0) <= TOL
        assert np.array_equal(its[:, i], out["iters"])


def test_general_storage_row_needs_a_table_or_polyline_batch():
    """The uniform-geometry kernels are compiled without the general reservoir row (its Brent iteration is the one
    out-of-line call of the boundary code and costs every kernel that carries it ~100 registers): such a batch
    is refused when it is stepped, with the way out in the message; table mode takes the same channel."""
    from fixture_batch import batch_from_problems
    from flowsim_amd import _abi as A
    fx, meta, probs = problems_of(os.path.join(GOLDEN, "storage_curve_closed.npz"))
    p = probs[0]
    assert boundary_kind(p.ds) == A.BC_STORAGE_CURVE
    with batch_from_problems([p], mode="table") as b:
        b.step(3)
        assert b.status()[0] == 0
    if is_rect(p):
        with batch_from_problems([p], mode="rect_uniform") as b:
            with pytest.raises(A.FlowsimError, match="FS_SEC_TABLE"):
                b.step(3)


def test_max_iter_and_status_reporting():
    """max_iter exhausted -> status 1 and the level is not advanced (preissmann.py:124-126)."""
    from fixture_batch import batch_from_problems
    from synth import rect_problem
    p = rect_problem(200, seed=3, n_steps=2)
    p.max_iter = 1
    with batch_from_problems([p]) as b:
        b.step(2)
        assert b.status()[0] == 1
        assert b.iterations()[1, 0] == 1


@pytest.mark.parametrize("N", [2, 3, 5, 64, 65, 66, 127, 128, 129, 512, 513, 1024, 1025, 2048, 2049, 4095, 4096])
def test_ragged_node_counts_against_the_c_oracle(N):
    """Smallest (one cell), off-by-one around every lane / wave capacity and the largest supported
    reach (4096 nodes): padding cells, the last-lane bookkeeping and the cross-wave fold."""
    from fixture_batch import batch_from_problems
    from oracle import c_oracle
    from synth import rect_problem
    probs = [rect_problem(N, seed=100 + N + s, n_steps=3) for s in range(2)]
    with batch_from_problems(probs) as b:
        b.step(3)
        assert np.all(b.status() == 0)
        h, Q = b.history_arrays()
        its = b.iterations()
    for i, p in enumerate(probs):
        out = c_oracle.run(p)
        assert rel_err(h[:, i], out["depth"], 1e-3) <= TOL, N
        assert rel_err(Q[:, i], out["flow"], 1.0) <= TOL, N
        assert np.array_equal(its[:, i], out["iters"]), N


@pytest.mark.parametrize("N", [4097, 6000, 8192, 16384, 32768])
@pytest.mark.parametrize("mode", ["rect_uniform", "table", "trap_uniform", "rect_uniform-multipass", "trap_uniform-multipass"])
def test_reaches_longer_than_the_lane_grid_against_the_c_oracle(N, mode, monkeypatch):
    """The reference has no limit on the number of nodes (solver.py:34-38, :53-55).  Beyond what one workgroup keeps on chip
    (4 096 rows for the uniform section modes, 2 048 for tables): uniform sections - a TEAM of workgroups, each holding 4 096 rows
    on chip, meeting once per Newton iteration through device memory (fs_kernel.hpp, round 4); tables, and uniform sections with
    FS_NO_TEAM=1 - the multi-pass kernel of fs_long.hpp (state in HBM / L2, two sweeps per Newton iteration).  Same bar as
    everywhere: the pivoted C oracle to 1e-8 with identical Newton counts; three reaches per batch, chunked stepping equal to one
    launch bit for bit."""
    from fixture_batch import batch_from_problems, hetero_batch_from_problems
    from flowsim_amd import _abi as A_
    from oracle import c_oracle
    from synth import rect_problem
    multipass = mode.endswith("-multipass")
    mode = mode.split("-")[0]
    if multipass:
        monkeypatch.setenv("FS_NO_TEAM", "1")
    if mode == "table" and N > 16384:
        pytest.skip("tables stop at 16 384 nodes")
    probs = [rect_problem(N, seed=900 + N % 97 + s, n_steps=3) for s in range(3)]
    if mode == "trap_uniform":
        for p in probs:
            p.geo["m_main"][:] = 1.5
            # (the rectangle's normal depth is not the trapezoid's: the run starts with a small adjustment wave - fine)
    # (a table is shared by a batch unless every reach brings its own: the three channels differ, so they do)
    with (hetero_batch_from_problems(probs) if mode == "table" else batch_from_problems(probs, mode=mode)) as b:
        b.step(3)
        assert np.all(b.status() == 0)
        info = b.kernel_info()
        assert 64 * info["cells_per_thread"] * info["waves_per_reach"] < N         # more than one workgroup holds
        e = A_.kernel_table()[b.kernel_index()]
        assert (e["team"], e["long_reach"]) == ((0, 1) if (multipass or mode == "table") else (1, 0)), e
        h, Q = b.history_arrays()
        its = b.iterations()
        hyd = b.hydrographs()
    for i, p in enumerate(probs):
        out = c_oracle.run(p)
        assert rel_err(h[:, i], out["depth"], 1e-3) <= TOL, (N, mode)
        assert rel_err(Q[:, i], out["flow"], 1.0) <= TOL, (N, mode)
        assert np.array_equal(its[:, i], out["iters"]), (N, mode)
    with batch_from_problems(probs[1:2], mode=mode, history=False) as c:          # one reach alone, stepped level by level
        for _ in range(3):
            c.step(1)
        assert np.array_equal(c.hydrographs()[:, :, 0], hyd[:, :, 1])


def test_long_reaches_of_different_lengths_in_one_batch():
    """Seven channels of 4 097 ... 9 999 nodes in ONE batch (per-reach node counts, steps and boundary kinds) on the multi-pass
    kernel: its state loads and stores run with consecutive lanes on consecutive nodes and are transposed through LDS, so the
    clamped copies beyond a reach's last node, passes that are half empty and waves that hold no node at all each occur here -
    seeded lengths around the pass (2 048 rows) and wave (512 rows) boundaries.  Pivoted C oracle, 1e-8, equal Newton counts."""
    from fixture_batch import hetero_batch_from_problems
    from oracle import c_oracle
    from synth import rect_problem
    rng = np.random.default_rng(4242)
    lengths = [4097, 9999, 2048 * 3 + 1, 2048 * 4, 512 * 9 + 1, int(rng.integers(4200, 9000)), int(rng.integers(4200, 9000))]
    probs = []
    for s, N in enumerate(lengths):
        q = rect_problem(N, seed=1300 + s, n_steps=3, dt=float(rng.choice([300.0, 600.0, 900.0])), dx=float(rng.choice([125.0, 250.0])),
                         theta=float(rng.uniform(0.55, 0.9)))
        if s % 2:
            q.geo["m_main"][:] = 1.0
        probs.append(q)
    with hetero_batch_from_problems(probs) as b:
        b.step(3)
        assert np.all(b.status() == 0)
        from flowsim_amd import _abi as A_
        assert A_.kernel_table()[b.kernel_index()]["long_reach"] == 1
        h, Q = b.history_arrays()
        its = b.iterations()
    for i, q in enumerate(probs):
        out = c_oracle.run(q)
        assert rel_err(h[:, i, :q.N], out["depth"], 1e-3) <= TOL, lengths[i]
        assert rel_err(Q[:, i, :q.N], out["flow"], 1.0) <= TOL, lengths[i]
        assert np.array_equal(its[:, i], out["iters"]), lengths[i]


def test_gerd_roseires_at_a_spatial_step_of_25_m():
    """cases/gerd_roseires refined to dx = 25 m: 4 817 nodes of compound sections with curvature and the gate curve - more than
    a table kernel keeps on chip; against the C oracle over the first levels"""
    from cases.gerd_roseires.model import build as build_gerd
    from fixture_batch import batch_from_problems
    from flowsim_amd.hydromodel.preissmann import boundary_to_spec
    from oracle import c_oracle
    solver, _ = build_gerd(inflow_hyd_func=None, sim_duration=4 * 3600, spatial_step=25)
    ch = solver.channel
    N = solver.number_of_nodes
    assert N > 4096
    nt = 4
    from flowsim_amd import PreissmannBatch, _abi as A
    us, ds = boundary_to_spec(ch.upstream_boundary, nt, solver.time_step), boundary_to_spec(ch.downstream_boundary, nt, solver.time_step)
    with PreissmannBatch(1, N, nt, section_mode="table", history=True) as b:
        b.set_scheme(solver.theta, solver.time_step, solver.spatial_step, 1e-6, 100)
        b.set_geometry_table(ch.node_geometry)
        b.set_boundary(A.UPSTREAM, us); b.set_boundary(A.DOWNSTREAM, ds)
        b.set_state(ch.initial_conditions[:, 0], ch.initial_conditions[:, 1])
        b.step(nt - 1)
        assert b.status()[0] == 0
        h, Q = b.history_arrays(0, nt)
        its = b.iterations(0, nt)[:, 0]
    geo = {k: np.asarray(ch.node_geometry[k], dtype=np.float64) for k in O.GEO_KEYS}
    p = O.Problem(geo=geo, h0=ch.initial_conditions[:, 0].copy(), Q0=ch.initial_conditions[:, 1].copy(),
                  us=O.BC("flow_hydrograph", bed_level=float(geo["z_bed"][0]), target=np.asarray(us.target, dtype=np.float64)[:nt]),
                  ds=O.BC("rating_curve", bed_level=float(ds.params["bed_level"]), rc_type="blend",
                          rc=dict(initial_stage=ds.params["stage0"], buffer=ds.params["buffer"], dY=ds.params["dY"],
                                  low=[ds.params["lo0"], ds.params["lo1"], ds.params["lo2"]], high=[ds.params["hi0"], ds.params["hi1"], ds.params["hi2"]])),
                  theta=solver.theta, dt=float(solver.time_step), dx=float(solver.spatial_step), nt=nt, tol=1e-6)
    out = c_oracle.run(p)
    assert out["status"] == 0
    assert rel_err(h[:, 0], out["depth"], 1e-3) <= TOL and rel_err(Q[:, 0], out["flow"], 1.0) <= TOL
    assert np.array_equal(its, out["iters"])


def test_failing_reach_does_not_disturb_its_neighbours():
    """One reach of a batch runs out of Newton iterations (its inflow jumps by three orders of
    magnitude); the others finish, bit-identical to a batch without it, and each status is its own."""
    from fixture_batch import batch_from_problems
    from synth import rect_problem
    probs = [rect_problem(300, seed=40 + s, n_steps=4) for s in range(4)]
    bad = rect_problem(300, seed=44, n_steps=4)
    bad.us.target = bad.us.target.copy()
    bad.us.target[2:] *= -5.0e3                      # flow reversal far beyond the channel: no convergence
    bad.max_iter = probs[0].max_iter = 12
    for p in probs:
        p.max_iter = 12
    with batch_from_problems(probs) as b:
        b.step(4)
        ref_h, ref_Q = b.history_arrays()
        assert np.all(b.status() == 0)
    mixed = probs[:2] + [bad] + probs[2:]
    with batch_from_problems(mixed) as b:
        b.step(4)
        st = b.status()
        h, Q = b.history_arrays()
        its = b.iterations()
    assert st[2] != 0 and np.all(np.delete(st, 2) == 0), st
    keep = [0, 1, 3, 4]
    assert np.array_equal(h[:, keep], ref_h) and np.array_equal(Q[:, keep], ref_Q)
    assert its[1, 2] > 0 and np.all(its[3:, 2] == 0)      # the failing reach stopped at the level that failed


# ---------------------------------------------------------------------------------------------
# uniform trapezoid mode (BASELINE configs[4] geometry) and fp32
# ---------------------------------------------------------------------------------------------
def trap_batch(fx, meta, dtype="f64", tol=None, history=True):
    from flowsim_amd import BoundarySpec, PreissmannBatch
    from flowsim_amd import _abi as A
    B, N, nt = meta["B"], meta["N"], meta["nt"]
    prm = fx["params"]                      # [B, (b, m, n, S0, Q_base, h_n, rc_a, rc_b)]
    b = PreissmannBatch(B, N, nt, dtype=dtype, section_mode="trap_uniform", history=history)
    b.set_scheme(meta["theta"], meta["dt"], meta["dx"], meta["tolerance"] if tol is None else tol, 100)
    L = (N - 1) * meta["dx"]
    b.set_geometry_uniform(prm[:, 0], prm[:, 2], prm[:, 3] * L, np.zeros(B), side_slope=prm[:, 1])
    b.set_boundary(A.UPSTREAM, BoundarySpec(A.BC_FLOW_HYDROGRAPH, {}, fx["us_target"].T))
    b.set_boundary(A.DOWNSTREAM, BoundarySpec(A.BC_RATING_POWER, dict(a=prm[:, 6], b=prm[:, 7], stage_shift=np.zeros(B),
                                                                    bed_level=np.zeros(B))))
    b.set_state(fx["initial_conditions"][:, :, 0], fx["initial_conditions"][:, :, 1])
    return b


def test_uniform_trapezoid_mode_matches_reference():
    """Per-reach trapezoids + per-reach power rating curves in one batch (4 different channels)."""
    fx, meta = O.load_fixture(os.path.join(GOLDEN, "synthetic_trap_64.npz"))
    with trap_batch(fx, meta) as b:
        b.step(meta["nt"] - 1)
        assert np.all(b.status() == 0)
        h, Q = b.history_arrays()
        its = b.iterations()
    for i in range(meta["B"]):
        assert rel_err(h[:, i], fx["depth"][i], 1e-3) <= TOL
        assert rel_err(Q[:, i], fx["flow"][i], 1.0) <= TOL
        assert np.array_equal(its[:, i], fx["iters"][i])


def test_fp32_trapezoid_tracks_the_fp64_reference():
    """fp32 arithmetic (BASELINE configs[4]): tolerance scaled to 1e-3 as SURVEY 8d prescribes; the
    hydrographs stay within 5e-4 relative of the fp64 reference (this is a throughput mode, the
    1e-8 bar applies to fp64)."""
    fx, meta = O.load_fixture(os.path.join(GOLDEN, "synthetic_trap_64.npz"))
    with trap_batch(fx, meta, dtype="f32", tol=1e-3) as b:
        b.step(meta["nt"] - 1)
        assert np.all(b.status() == 0)
        h, Q = b.history_arrays()
    for i in range(meta["B"]):
        assert rel_err(h[:, i], fx["depth"][i], 1e-3) <= 5e-4
        assert rel_err(Q[:, i], fx["flow"][i], 1.0) <= 5e-4


def test_fp32_rectangular_tracks_the_fp64_reference():
    from fixture_batch import batch_from_problems
    fx, meta, probs = problems_of(os.path.join(GOLDEN, "synthetic_rect_512.npz"))
    for p in probs:
        p.tol = 1e-3
    with batch_from_problems(probs, dtype="f32") as b:
        b.step(probs[0].nt - 1)
        assert np.all(b.status() == 0)
        h, Q = b.history_arrays()
    for i in range(len(probs)):
        assert rel_err(h[:, i], fx["depth"][i], 1e-3) <= 5e-4
        assert rel_err(Q[:, i], fx["flow"][i], 1.0) <= 5e-4


# Reaches of the C5 population (flowsim_amd.synthetic.c5_reach_parameters: seed, global index) on which ROUND 1's
# elimination - 2x2 blocks in the order of the classical double sweep, no pivoting - met singular pivot blocks: steep
# and shallow, uniform flow within a few per cent of the depth h* ~ (5/3) S0 dx at which the block {momentum row of
# cell i, continuity row of cell i+1} is singular (fp32: exact zeros, FS_NAN; it needed a perturbation of the Jacobian).
# The scalar tridiagonal system the solve works on since round 2 (fs_device.hpp) has no such block; these reaches stay
# as the regression test of that.
SINGULAR_PIVOT_REACHES = [(4, 506130), (9, 111108), (3, 154763), (3, 180059), (3, 628857), (6, 722298),
                          (10, 1010160), (20260214, 89004), (20260214, 163455), (9, 848778), (9, 972785)]


def _singular_pivot_problems(N, n_steps, tol):
    from flowsim_amd.synthetic import c5_reach_parameters
    from synth import trap_problem
    probs = []
    for seed, idx in SINGULAR_PIVOT_REACHES:
        b, m, n, S0, Qb = (float(v[0]) for v in c5_reach_parameters(idx, 1, seed))
        probs.append(trap_problem(b, m, n, S0, Qb, N, n_steps, tol=tol))
    return probs


@pytest.mark.parametrize("shape", ["8,1", "16,4"])
def test_fp32_rides_through_a_singular_pivot_block(shape, monkeypatch):
    """fp32, 512 nodes, the whole hydrograph: every reach converges at every level - no pivoting, no perturbation -
    and stays within 5e-3 of the fp64 run of the same reach (the fp32 mode stops Newton at a residual norm of
    1e-3)."""
    from fixture_batch import batch_from_problems
    monkeypatch.setenv("FS_KERNEL_SHAPE", shape)
    n_steps = 36
    out = {}
    for dtype, tol in (("f64", 1e-6), ("f32", 1e-3)):
        probs = _singular_pivot_problems(512, n_steps, tol)
        with batch_from_problems(probs, mode="trap_uniform", dtype=dtype, history=False) as b:
            for k in (4, n_steps - 4):             # two launches, as bench.py runs it
                b.step(k)
            assert np.all(b.status() == 0), (dtype, b.status())
            out[dtype] = (b.hydrographs(), b.iterations(1, n_steps))
    assert out["f32"][1].max() <= 5 and out["f64"][1].max() <= 6
    hy32, hy64 = out["f32"][0], out["f64"][0]
    assert np.max(np.abs(hy32 - hy64) / np.maximum(np.abs(hy64), 1e-3)) <= 5e-3


def test_fp64_near_singular_pivot_block_against_the_c_oracle():
    """The same reaches in fp64 against the partially pivoted banded LU of the C oracle."""
    from fixture_batch import batch_from_problems
    from oracle import c_oracle
    probs = _singular_pivot_problems(500, 6, 1e-6)     # 8 cells per lane, ragged, rating-curve instantiation
    with batch_from_problems(probs, mode="trap_uniform") as b:
        b.step(6)
        assert np.all(b.status() == 0)
        h, Q = b.history_arrays()
        its = b.iterations()
    for i, p in enumerate(probs):
        ref = c_oracle.run(p)
        assert rel_err(h[:, i], ref["depth"], 1e-3) <= TOL, SINGULAR_PIVOT_REACHES[i]
        assert rel_err(Q[:, i], ref["flow"], 1.0) <= TOL, SINGULAR_PIVOT_REACHES[i]
        assert np.array_equal(its[:, i], ref["iters"]), SINGULAR_PIVOT_REACHES[i]


@pytest.mark.parametrize("N", [2, 64, 65, 66, 128, 129, 130, 257, 513])
def test_trapezoid_reaches_that_fill_the_wave_exactly(N):
    """General-section kernels take a lane's last node from its right neighbour (FS_SHARE_NODE); lane 63 has none
    and evaluates the node itself when the reach fills the wave to the last cell (N - 1 = 64 M), one cell
    less leaves a padding cell there, one more moves to the next shape."""
    from fixture_batch import batch_from_problems
    from oracle import c_oracle
    from synth import trap_problem
    probs = [trap_problem(30.0 + 10 * i, 1.5, 0.03, 4e-4, 60.0 + 25 * i, N, 4, dt=900.0, dx=400.0) for i in range(3)]
    with batch_from_problems(probs, mode="trap_uniform") as b:
        b.step(4)
        assert np.all(b.status() == 0)
        h, Q = b.history_arrays()
        its = b.iterations()
    for i, p in enumerate(probs):
        ref = c_oracle.run(p)
        assert rel_err(h[:, i], ref["depth"], 1e-3) <= TOL, N
        assert rel_err(Q[:, i], ref["flow"], 1.0) <= TOL, N
        assert np.array_equal(its[:, i], ref["iters"]), N


def test_flow_regime_grid_against_the_c_oracle():
    """Bed slope x spatial step x base flow, from backwater-resolved grids (h / h* = 250) to kinematic ones
    (h / h* = 0.02, Froude up to 0.8): the unpivoted tree elimination against partially pivoted LU on both
    sides of the depth h* where round 1's pivot block changed sign (tools/scan_regimes.py prints the table)."""
    from fixture_batch import batch_from_problems
    from oracle import c_oracle
    from synth import trap_problem
    for S0 in (1e-4, 1e-3, 5e-3):
        for dx in (100.0, 500.0, 2000.0):
            for q in (0.3, 1.0, 4.0):
                p = trap_problem(40.0, 2.0, 0.03, S0, q * 40.0, 130, 4, dx=dx)
                ref = c_oracle.run(p)
                with batch_from_problems([p], mode="trap_uniform") as b:
                    b.step(4)
                    assert b.status()[0] == 0 and ref["status"] == 0, (S0, dx, q)
                    h, Q = b.history_arrays()
                    its = b.iterations()[:, 0]
                assert rel_err(h[:, 0], ref["depth"], 1e-3) <= TOL, (S0, dx, q)
                assert rel_err(Q[:, 0], ref["flow"], 1.0) <= TOL, (S0, dx, q)
                assert np.array_equal(its, ref["iters"]), (S0, dx, q)


@pytest.mark.parametrize("case", ["trap_512", "trap_500", "rect_512"])
def test_kernels_compiled_for_a_boundary_pair_match_the_general_ones(case, monkeypatch):
    """Instantiations with the downstream kind fixed at compile time (BCK >= 2, fs_kernel.hpp) against the
    general kernels of the same shape (FS_KERNEL_GENERAL=1): same rows, same source - equal iteration counts,
    hydrographs equal to a few ulp (the compiler contracts a row inlined next to one kind differently from
    the same row next to nine), and fewer registers (which is the point of them)."""
    from fixture_batch import batch_from_problems
    if case.startswith("rect"):
        _, _, probs = problems_of(os.path.join(GOLDEN, "synthetic_rect_512.npz"))
        mode = "rect_uniform"
    elif case.startswith("trap"):
        probs = _singular_pivot_problems(int(case[5:]), 5, 1e-6)[:4]
        mode = "trap_uniform"
    else:
        _, _, probs = problems_of(os.path.join(GOLDEN, case + ".npz"))
        probs, mode = probs[:1], "table"
    monkeypatch.setenv("FS_KERNEL_SHAPE", "2,1" if mode == "table" else "8,1")
    n = min(probs[0].nt - 1, 6)
    out = []
    for general in ("1", "0"):
        monkeypatch.setenv("FS_KERNEL_GENERAL", general)
        with batch_from_problems(probs, mode=mode, history=False) as b:
            b.step(n)
            assert np.all(b.status() == 0)
            out.append((b.hydrographs(), b.iterations(), b.kernel_info()))
    (hy_g, it_g, k_g), (hy_p, it_p, k_p) = out
    assert k_p["vgprs"] < k_g["vgprs"], (k_g, k_p)          # a different kernel did run
    assert np.array_equal(it_g, it_p)
    assert np.max(np.abs(hy_g - hy_p) / np.maximum(np.abs(hy_g), 1e-3)) <= 1e-13


@pytest.mark.parametrize("case", ["synthetic_rect_512", "gerd", "irr_mixed", "trap_512"])
def test_batches_without_history_run_the_same_numbers(case, monkeypatch):
    """Batches created without FS_FLAG_HISTORY / FS_FLAG_TRACE get the instantiations compiled without those
    stores (DIAG = false, fs_entries.hpp): same hydrographs to a few ulp, same iteration counts."""
    from fixture_batch import batch_from_problems
    if case.startswith("trap"):
        probs, mode = _singular_pivot_problems(512, 5, 1e-6)[:3], "trap_uniform"
    else:
        _, _, probs = problems_of(os.path.join(GOLDEN, case + ".npz"))
        mode = "auto"
        if case in ("gerd", "irr_mixed"):
            probs = probs[:1]
            monkeypatch.setenv("FS_KERNEL_SHAPE", "2,1")
    n = min(probs[0].nt - 1, 6)
    out = []
    for history in (True, False):
        with batch_from_problems(probs, mode=mode, history=history) as b:
            b.step(n)
            assert np.all(b.status() == 0)
            out.append((b.hydrographs(), b.iterations()))
    assert np.array_equal(out[0][1], out[1][1])
    assert np.max(np.abs(out[0][0] - out[1][0]) / np.maximum(np.abs(out[0][0]), 1e-3)) <= 1e-13


def test_full_size_batch_properties():
    """BASELINE configs[2] size (65 536 reaches x 4 096 nodes, fp64) through size-independent
    properties: (i) 256 distinct channels replicated 256 times give bitwise identical copies,
    (ii) identical to the 256-reach batch, (iii) every reach converges, (iv) mass balance of the
    routed wave: inflow - outflow volume equals the change of storage to 1e-9 of the inflow volume."""
    from flowsim_amd import BoundarySpec, PreissmannBatch
    from flowsim_amd import _abi as A
    from flowsim_amd.synthetic import c3_reach_parameters, inflow_table, normal_depth_rect
    N, K, dt, dx = 4096, 6, 600.0, 250.0
    b0, n0, S0, Qb0 = c3_reach_parameters(0, 256)
    hn0 = normal_depth_rect(b0, n0, S0, Qb0)
    L = (N - 1) * dx

    def run(rep):
        b, n, S, Qb, hn = (np.tile(v, rep) for v in (b0, n0, S0, Qb0, hn0))
        B = 256 * rep
        bt = PreissmannBatch(B, N, K + 1, section_mode="rect_uniform")
        bt.set_scheme(0.6, dt, dx, 1e-6, 100)
        bt.set_geometry_uniform(b, n, S * L, np.zeros(B))
        bt.set_boundary(A.UPSTREAM, BoundarySpec(A.BC_FLOW_HYDROGRAPH, {}, inflow_table(Qb, K + 1, dt)))
        bt.set_boundary(A.DOWNSTREAM, BoundarySpec(A.BC_NORMAL_DEPTH, dict(bed_slope=S, bed_level=np.zeros(B))))
        bt.set_state_uniform(hn, Qb)
        bt.step(K)
        out = bt.hydrographs(), bt.iterations(), bt.status(), bt.state()
        bt.close()
        return out

    hyd1, it1, st1, (h1, Q1) = run(1)
    hyd, it, st, (h, Q) = run(256)
    assert np.all(st == 0) and np.all(st1 == 0)
    assert np.array_equal(hyd.reshape(K + 1, 4, 256, 256), np.broadcast_to(hyd1[:, :, None, :], (K + 1, 4, 256, 256)))
    assert np.array_equal(it.reshape(K + 1, 256, 256), np.broadcast_to(it1[:, None, :], (K + 1, 256, 256)))
    assert np.array_equal(h.reshape(256, 256, N), np.broadcast_to(h1[None], (256, 256, N)))
    # continuity of the scheme: trapezoidal inflow/outflow volumes vs stored volume (rectangular: A = b h)
    vol_in = dt * (0.5 * (hyd1[:-1, 1] + hyd1[1:, 1])).sum(axis=0)
    vol_out = dt * (0.5 * (hyd1[:-1, 3] + hyd1[1:, 3])).sum(axis=0)
    w = np.full(N, dx); w[0] = w[-1] = dx / 2
    stored = (b0[:, None] * (h1 - hn0[:, None]) * w[None, :]).sum(axis=1)
    th = 0.6     # the box scheme conserves the theta-weighted fluxes; compare at the scheme's own weighting
    q_in = th * hyd1[1:, 1] + (1 - th) * hyd1[:-1, 1]
    q_out = th * hyd1[1:, 3] + (1 - th) * hyd1[:-1, 3]
    net = dt * (q_in - q_out).sum(axis=0)
    assert np.max(np.abs(net - stored) / vol_in) <= 1e-9


def test_sixteen_full_width_reaches_against_the_c_oracle():
    """N = 4096, 16 different channels, 6 levels: HIP path vs the compiled CPU oracle."""
    from fixture_batch import batch_from_problems
    from oracle import c_oracle as CO
    from synth import rect_problem
    probs = [rect_problem(4096, seed=100 + s, n_steps=6) for s in range(16)]
    with batch_from_problems(probs, mode="rect_uniform") as b:
        b.step(6)
        h, Q = b.history_arrays()
        its = b.iterations()
        assert np.all(b.status() == 0)
    for i, p in enumerate(probs):
        out = CO.run(p)
        assert rel_err(h[:, i], out["depth"], 1e-3) <= TOL
        assert rel_err(Q[:, i], out["flow"], 1.0) <= TOL
        assert np.array_equal(its[:, i], out["iters"])
