"""Ragged batches (fs_batch_set_reach_nodes) in the places the round-3 review found them treated as full ones, and the bound on
the polyline stage tables.

  * derive_fields_kernel (Solver.prepare_results, solver.py:65-127) interpolated the bed of RECT / TRAP_UNIFORM reaches over the
    batch's N instead of the reach's own node count (the step kernels use the reach's own: Geometry::init) and wrote results for
    slots past a reach's end from history that was never written;
  * fs_batch_set_state filled the downstream half of the level-0 hydrograph row from column N - 1 (caller padding);
  * fs_batch_restart's "storage needs its stage" guard looked at reach 0's kind only;
  * fs_batch_set_geometry_irregular(_per_reach) staged stage tables of any size."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import preissmann_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-8


def rel(a, b, floor):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def _uniform_batch(probs, mode, N, node_counts=None, pad_value=-777.0, nodes_after_state=False):
    """one RECT / TRAP_UNIFORM batch of prismatic problems with (optionally) ragged node counts; the state rows are padded with a
    sentinel no kernel may ever look at"""
    from flowsim_amd import BoundarySpec, PreissmannBatch
    from flowsim_amd import _abi as A
    B, p0 = len(probs), probs[0]
    b = PreissmannBatch(B, N, p0.nt, section_mode=mode, history=True)
    b.set_scheme(p0.theta, p0.dt, p0.dx, p0.tol, p0.max_iter)
    b.set_geometry_uniform([p.geo["b_main"][0] for p in probs], [p.geo["n_main"][0] for p in probs], [p.geo["z_bed"][0] for p in probs],
                           [p.geo["z_bed"][-1] for p in probs], side_slope=[p.geo["m_main"][0] for p in probs] if mode == "trap_uniform" else None)
    if node_counts is not None and not nodes_after_state:
        b.set_reach_nodes(node_counts)
    b.set_boundary(A.UPSTREAM, BoundarySpec(A.BC_FLOW_HYDROGRAPH, {}, np.stack([p.us.target for p in probs], axis=1)))
    b.set_boundary(A.DOWNSTREAM, BoundarySpec(A.BC_NORMAL_DEPTH, dict(bed_slope=np.array([p.ds.bed_slope for p in probs]), bed_level=np.zeros(B))))
    pad = lambda a: np.concatenate([a, np.full(N - len(a), pad_value)])
    b.set_state(np.stack([pad(p.h0) for p in probs]), np.stack([pad(p.Q0) for p in probs]))
    if node_counts is not None and nodes_after_state:
        b.set_reach_nodes(node_counts)
    return b


@pytest.mark.parametrize("mode", ["rect_uniform", "trap_uniform"])
def test_derived_fields_of_a_ragged_uniform_batch(mode):
    from synth import rect_problem
    lengths = [37, 90, 128, 61]
    probs = []
    for s, n in enumerate(lengths):
        q = rect_problem(n, seed=700 + s, n_steps=4)
        if mode == "trap_uniform":
            q.geo["m_main"][:] = 1.5
        probs.append(q)
    N = max(lengths)
    fields = ("level", "area", "top_width", "froude_number", "velocity", "wave_celerity", "amplitude")
    with _uniform_batch(probs, mode, N, lengths) as b:
        b.step(4)
        assert np.all(b.status() == 0)
        h, Q = b.history_arrays(0, 5)
        d = b.derive(0, 5)
    for r, (n, q) in enumerate(zip(lengths, probs)):
        with _uniform_batch([q], mode, n) as one:                 # the reach alone, as a full batch of its own length
            one.step(4)
            d1 = one.derive(0, 5)
            h1, _ = one.history_arrays(0, 5)
        assert np.array_equal(h[:, r, :n], h1[:, 0]), (mode, n)         # (the step kernels were right before: same bits)
        # the bed the oracle interpolates over the reach's OWN nodes (cross_section.py:887-900), not over the batch's N
        bed = q.geo["z_bed"][0] * (1.0 - np.arange(n) / (n - 1)) + q.geo["z_bed"][-1] * (np.arange(n) / (n - 1))
        assert np.max(np.abs(d["level"][:, r, :n] - h[:, r, :n] - bed[None, :])) <= 1e-12 * max(1.0, bed[0]), (mode, n)
        for f in fields:
            assert np.array_equal(d[f][:, r, :n], d1[f][:, 0]), (mode, n, f)
            assert np.all(d[f][:, r, n:] == 0.0), (mode, n, f)            # slots past the reach's end: zero, not derived garbage
        assert np.array_equal(d["peak_amplitude"][r, :n], d1["peak_amplitude"][0]) and np.all(d["peak_amplitude"][r, n:] == 0.0)


@pytest.mark.parametrize("nodes_after_state", [False, True])
def test_level0_hydrograph_row_holds_each_reachs_last_node(nodes_after_state):
    from synth import rect_problem
    lengths = [20, 64, 33]
    probs = [rect_problem(n, seed=800 + s, n_steps=2, steady=False) for s, n in enumerate(lengths)]
    for q in probs:                                          # a sloping initial profile: the last node differs from its neighbours
        q.h0 = q.h0 * (1.0 + 0.05 * np.arange(q.N) / q.N)
    with _uniform_batch(probs, "rect_uniform", 64, lengths, nodes_after_state=nodes_after_state) as b:
        row = b.hydrographs(0, 1)[0]
        for r, q in enumerate(probs):
            assert row[0, r] == q.h0[0] and row[1, r] == q.Q0[0]
            assert row[2, r] == q.h0[-1] and row[3, r] == q.Q0[-1], (r, row[2, r], q.h0[-1])


def test_restart_with_per_reach_kinds_needs_the_storage_stage_whichever_reach_has_the_storage():
    from fixture_batch import hetero_batch_from_problems
    from flowsim_amd import _abi as A
    # two channels of the reference's random sweep: reach 0 ends in a fixed depth, reach 1 in a LumpedStorage (closed form)
    sweep = {i: (fx, m) for i, fx, m in O.sweep_cases(os.path.join(GOLDEN, "random_sweep.npz"))}
    p_plain, p_store = (O.problem_from_fixture(*sweep[i]) for i in (14, 11))
    assert p_plain.ds.storage is None and p_store.ds.storage is not None and min(p_plain.nt, p_store.nt) >= 4
    with hetero_batch_from_problems([p_plain, p_store]) as b:
        b.step(2)
        assert np.all(b.status() == 0)
        h, Q = b.state(); hg, Qg = b.guess(); Y = b.storage_stage()
        with hetero_batch_from_problems([p_plain, p_store]) as c:
            with pytest.raises(A.FlowsimError, match="storage"):
                c.restart(2, h, Q, hg, Qg)                             # reach 1 would go on from stage 0
            c.restart(2, h, Q, hg, Qg, storage_stage=Y)
            c.step(1)
            b.step(1)
            assert np.all(c.status() == 0)
            assert np.array_equal(c.hydrographs(3, 1), b.hydrographs(3, 1))


def test_stage_tables_beyond_the_bound_fall_back_to_the_edge_walk(monkeypatch):
    from fixture_batch import batch_from_problems
    fx, meta = O.load_fixture(os.path.join(GOLDEN, "irr_levee.npz"))
    p = O.problem_from_fixture(fx, meta)
    p.nt = 9
    res = {}
    for cap in (None, "4096"):
        if cap:
            monkeypatch.setenv("FS_POLY_TABLE_MAX_BYTES", cap)
        with batch_from_problems([p], mode="irregular") as b:
            assert b.poly_tables() == (0 if cap else 1)
            b.step(p.nt - 1)
            assert np.all(b.status() == 0)
            res[cap] = b.history_arrays(0, p.nt) + (b.iterations(0, p.nt),)
    for h, Q, its in res.values():
        assert rel(h[:, 0], fx["depth"][:9], 1e-3) <= TOL and rel(Q[:, 0], fx["flow"][:9], 1.0) <= TOL
        assert np.array_equal(its[:, 0], fx["iters"][:9])


def test_a_ragged_batch_of_long_uniform_reaches_on_the_team_kernel():
    """Reaches of 4 097 ... 16 384 nodes in ONE rectangular batch (per-reach node counts): every reach a team of four workgroups
    (fs_kernel.hpp, TEAM), of which a shorter reach leaves one, two or three with nothing but identity rows - they still post,
    wait and decide with the others.  Pivoted C oracle, 1e-8, identical Newton counts; the sentinel padding is never read."""
    from flowsim_amd import _abi as A
    from oracle import c_oracle
    from synth import rect_problem
    lengths = [4097, 9999, 16384, 6145, 12289]
    probs = [rect_problem(n, seed=1500 + s, n_steps=3) for s, n in enumerate(lengths)]
    with _uniform_batch(probs, "rect_uniform", max(lengths), lengths) as b:
        b.step(3)
        e = A.kernel_table()[b.kernel_index()]
        assert e["team"] == 1 and e["long_reach"] == 0
        assert np.all(b.status() == 0), b.status()
        h, Q = b.history_arrays(0, 4)
        its = b.iterations(0, 4)
        hyd = b.hydrographs(0, 4)
    for r, (n, q) in enumerate(zip(lengths, probs)):
        ref = c_oracle.run(q)
        assert rel(h[:, r, :n], ref["depth"], 1e-3) <= TOL and rel(Q[:, r, :n], ref["flow"], 1.0) <= TOL, n
        assert np.array_equal(its[:, r], ref["iters"]), n
        assert np.array_equal(hyd[:, 2, r], h[:, r, n - 1]) and np.array_equal(hyd[:, 3, r], Q[:, r, n - 1])


def test_team_kernel_is_deterministic_and_chunk_invariant_at_size():
    """1 024 reaches x 16 384 nodes (4 096 workgroups meeting once per Newton iteration through device memory, sixteen rounds of the
    chip): two runs give the same bits, stepping level by level gives the same bits as one launch, every reach converges with the
    counts of the first run, and three of the reaches agree with the pivoted C oracle."""
    from flowsim_amd import BoundarySpec, PreissmannBatch
    from flowsim_amd import _abi as A
    from flowsim_amd.synthetic import c3_reach_parameters, inflow_table, normal_depth_rect
    from oracle import c_oracle
    B, N, K = 1024, 16384, 4
    b_, n_, S0, Qb = c3_reach_parameters(0, B); hn = normal_depth_rect(b_, n_, S0, Qb); L = (N - 1) * 250.0

    def run(chunks):
        with PreissmannBatch(B, N, K + 1, section_mode="rect_uniform", monitor=False) as bt:
            bt.set_scheme(0.6, 600.0, 250.0, 1e-6, 100); bt.set_geometry_uniform(b_, n_, S0 * L, np.zeros(B))
            bt.set_boundary(A.UPSTREAM, BoundarySpec(A.BC_FLOW_HYDROGRAPH, {}, inflow_table(Qb, K + 1, 600.0)))
            bt.set_boundary(A.DOWNSTREAM, BoundarySpec(A.BC_NORMAL_DEPTH, dict(bed_slope=S0, bed_level=np.zeros(B))))
            bt.set_state_uniform(hn, Qb)
            for c in chunks:
                bt.step(c)
            assert A.kernel_table()[bt.kernel_index()]["team"] == 1
            assert np.all(bt.status() == 0)
            return bt.hydrographs(0, K + 1), bt.iterations(0, K + 1), bt.state()
    one, again, chunked = run([K]), run([K]), run([1] * K)
    for other in (again, chunked):
        assert np.array_equal(one[0], other[0]) and np.array_equal(one[1], other[1])
        assert np.array_equal(one[2][0], other[2][0]) and np.array_equal(one[2][1], other[2][1])
    tgt = inflow_table(Qb, K + 1, 600.0)
    for r in (0, 511, 1023):
        geo = {k: np.zeros(N) for k in O.GEO_KEYS}
        geo["b_main"][:] = b_[r]; geo["n_main"][:] = n_[r]; geo["n_left"][:] = n_[r]; geo["n_right"][:] = n_[r]
        geo["z_bed"] = S0[r] * L * (1 - np.arange(N) / (N - 1))
        p = O.Problem(geo=geo, h0=np.full(N, hn[r]), Q0=np.full(N, Qb[r]), us=O.BC("flow_hydrograph", bed_level=S0[r] * L, target=tgt[:, r].copy()),
                      ds=O.BC("normal_depth", bed_level=0.0, bed_slope=float(S0[r])), theta=0.6, dt=600.0, dx=250.0, nt=K + 1, tol=1e-6)
        ref = c_oracle.run(p)
        assert np.array_equal(one[1][:, r], ref["iters"])
        assert rel(one[0][:, 2, r], ref["depth"][:, -1], 1e-3) <= TOL and rel(one[0][:, 3, r], ref["flow"][:, -1], 1.0) <= TOL
        assert rel(one[2][0][r], ref["depth"][-1], 1e-3) <= TOL and rel(one[2][1][r], ref["flow"][-1], 1.0) <= TOL


def test_one_general_reservoir_per_reach():
    """FS_BC_STORAGE_CURVE with per-reach parameters (round 4: the round-3 review's "general reservoir as a per-reach kind"): the three
    reference fixtures with a general LumpedStorage behind them - area curve + polynomial outflow curve + entrance losses on a rectangle,
    area curve + power outflow curve + losses on a trapezoid, a closed reservoir - as ONE batch, every reach its own channel table and its own
    reservoir (scalars, area curve, outflow curve, losses); brentq in the reference, Brent in the kernel.  Each against its fixture: 1e-8,
    identical Newton counts, the reservoir stages included."""
    from fixture_batch import boundary_spec, merge_specs
    from flowsim_amd import PreissmannBatch
    from flowsim_amd import _abi as A
    names = ("storage_curve_poly_losses", "storage_curve_power_trap", "storage_curve_closed")
    fxs = [O.load_fixture(os.path.join(GOLDEN, n + ".npz")) for n in names]
    probs = [O.problem_from_fixture(fx, meta) for fx, meta in fxs]
    p0, B = probs[0], len(probs)
    assert all((p.N, p.nt, p.theta, p.dt, p.dx, p.tol) == (p0.N, p0.nt, p0.theta, p0.dt, p0.dx, p0.tol) for p in probs)
    with PreissmannBatch(B, p0.N, p0.nt, section_mode="table", history=True) as b:
        b.set_scheme(p0.theta, p0.dt, p0.dx, p0.tol, p0.max_iter)
        b.set_geometry_table({k: np.stack([p.geo[k] for p in probs]) for k in A.GEO_ROWS})
        b.set_boundary(A.UPSTREAM, merge_specs([boundary_spec(p.us, p.nt) for p in probs], B))
        ds = merge_specs([boundary_spec(p.ds, p.nt) for p in probs], B)
        assert ds.kind == A.BC_STORAGE_CURVE and ds.params["curve"].ndim == 3 and len(set(ds.params["rc_type"])) == 3
        b.set_boundary(A.DOWNSTREAM, ds)
        b.set_state(np.stack([p.h0 for p in probs]), np.stack([p.Q0 for p in probs]))
        b.step(p0.nt - 1)
        assert np.all(b.status() == 0), b.status()
        h, Q = b.history_arrays(0, p0.nt)
        its = b.iterations(0, p0.nt)
        stages = b.storage_stages(0, p0.nt)
    for r, ((fx, meta), p) in enumerate(zip(fxs, probs)):
        assert rel(h[:, r], fx["depth"], 1e-3) <= TOL and rel(Q[:, r], fx["flow"], 1.0) <= TOL, names[r]
        assert np.array_equal(its[:, r], fx["iters"]), names[r]
        if "storage_stage" in fx.files:
            assert rel(stages[1:, r], fx["storage_stage"][:, 1], 1e-3) <= TOL, names[r]


def test_general_reservoirs_behind_some_reaches_of_a_batch():
    """FS_BC_STORAGE_CURVE as a PER-REACH kind next to the closed-form ones (fs_batch_set_bc_per_reach_wide): twelve reference fixtures
    as ONE batch - three general reservoirs with area curves of different lengths (one with none), the constant-area reservoir of
    cases/example (BASELINE C1), a rating curve, stage hydrographs, a fixed depth, normal depth on a compound channel and on
    cases/akbari_firoozi (C2); boundary kinds that differ at both ends, 17 ... 41 nodes, seven (theta, dt, dx) triples, two tolerances
    (every reach its own: fs_batch_set_reach_tolerance), 21 ... 49 levels.  Each reach against its fixture: 1e-8, identical Newton
    counts (C1's 13 iterations of the first level among them), reservoir stages."""
    from fixture_batch import boundary_spec, hetero_batch_from_problems
    from flowsim_amd import _abi as A
    names = ("storage_curve_poly_losses", "bc_trap_poly", "storage_curve_closed", "bc_compound_normal", "bc_stage_fixed",
             "storage_curve_power_trap", "bc_us_fixed_ds_flow", "bc_us_normal_ds_stage", "bc_us_rating_ds_stage", "storage_curve_poly_losses",
             "example", "akbari")        # BASELINE configs C1 and C2: tolerance 1e-4 next to the others' 1e-6 (fs_batch_set_reach_tolerance)
    fxs = [O.load_fixture(os.path.join(GOLDEN, n + ".npz")) for n in names]
    probs = [O.problem_from_fixture(fx, meta) for fx, meta in fxs]
    # the fixtures' area curves all have 61 points: one gets two more beyond its last stage, at the last area - np.interp clamps there
    # (lumped_storage.py:152-157), so the reservoir is the same one and the rows of the batch now differ in length
    c = np.asarray(probs[5].ds.storage["curve"], dtype=np.float64)
    probs[5].ds.storage["curve"] = np.concatenate([c, [[c[-1, 0] + 10.0, c[-1, 1]], [c[-1, 0] + 25.0, c[-1, 1]]]])
    kinds = [boundary_spec(p.ds, p.nt).kind for p in probs]
    curves = [len(boundary_spec(p.ds, p.nt).params.get("curve", ())) for p, k in zip(probs, kinds) if k == A.BC_STORAGE_CURVE]
    assert kinds.count(A.BC_STORAGE_CURVE) == 4 and len(set(kinds)) >= 5 and len(set(curves)) >= 2, (kinds, curves)
    assert len({p.N for p in probs}) >= 4 and len({(p.theta, p.dt, p.dx) for p in probs}) >= 7 and len({p.tol for p in probs}) == 2
    L = max(p.nt for p in probs)
    with hetero_batch_from_problems(probs, mode="table", history=True) as b:
        # level by level: a reach whose fixture has ended goes on with its last target (its rows beyond are not looked at)
        b.step(L - 1)
        st = b.status()
        h, Q = b.history_arrays(0, L)
        its = b.iterations(0, L)
        stages = b.storage_stages(0, L)
        e = A.kernel_table()[b.kernel_index()]
        assert e["boundary_class"] == -1
    for r, ((fx, meta), p) in enumerate(zip(fxs, probs)):
        assert st[r] == 0 or p.nt < L, (names[r], st[r])
        assert rel(h[:p.nt, r, :p.N], fx["depth"], 1e-3) <= TOL and rel(Q[:p.nt, r, :p.N], fx["flow"], 1.0) <= TOL, names[r]
        assert np.array_equal(its[:p.nt, r], fx["iters"]), names[r]
        if "storage_stage" in fx.files:
            assert rel(stages[1:p.nt, r], fx["storage_stage"][:, 1], 1e-3) <= TOL, names[r]


def test_per_reach_kinds_are_checked_reach_by_reach():
    """fs_batch_set_bc_per_reach(_wide): what the reference raises per Boundary object (boundary.py:33, :83-87) comes back per reach"""
    from flowsim_amd import BoundarySpec, PreissmannBatch
    from flowsim_amd import _abi as A
    E = A.FlowsimError
    nd = BoundarySpec(A.BC_NORMAL_DEPTH, dict(bed_slope=1e-3, bed_level=0.0))
    flow = BoundarySpec(A.BC_FLOW_HYDROGRAPH, {}, np.full(4, 10.0))
    curve = np.stack([np.linspace(0.0, 10.0, 5), np.linspace(1e4, 2e4, 5)], axis=1)
    sc = lambda c: BoundarySpec(A.BC_STORAGE_CURVE, dict(min_stage=0.0, Y_min=0.0, Y_max=10.0, bed_level=0.0, surface_area=0.0, curve=c))
    with PreissmannBatch(3, 16, 4, section_mode="rect_uniform") as b:          # uniform sections: no kernel with the general rows
        with pytest.raises(E, match="FS_SEC_TABLE or FS_SEC_IRREGULAR"):
            b.set_boundary_per_reach(A.DOWNSTREAM, [nd, BoundarySpec(A.BC_HOST_ROW), nd])
        with pytest.raises(E, match="FS_SEC_TABLE or FS_SEC_IRREGULAR"):
            b.set_boundary_per_reach(A.DOWNSTREAM, [nd, nd, sc(curve)])
    with PreissmannBatch(3, 16, 4, section_mode="table") as b:
        with pytest.raises(E, match="downstream only"):
            b.set_boundary_per_reach(A.UPSTREAM, [flow, sc(curve), flow])
        with pytest.raises(E, match="must be increasing"):
            b.set_boundary_per_reach(A.DOWNSTREAM, [nd, sc(curve[::-1]), nd])
        with pytest.raises(E, match="n_curve 0 or >= 2"):
            b.set_boundary_per_reach(A.DOWNSTREAM, [nd, sc(curve[:1]), nd])
        with pytest.raises(E, match="Insufficient arguments"):              # a closed reservoir without an area: boundary.py:83
            b.set_boundary_per_reach(A.DOWNSTREAM, [nd, sc(np.empty((0, 2))), nd])
        with pytest.raises(E, match="Invalid boundary condition"):          # boundary.py:33
            b.set_boundary_per_reach(A.DOWNSTREAM, [nd, BoundarySpec(17), nd])
        with pytest.raises(E, match="Insufficient arguments"):              # a hydrograph kind without its table: boundary.py:87
            b.set_boundary_per_reach(A.UPSTREAM, [BoundarySpec(A.BC_FLOW_HYDROGRAPH), nd, nd])
        b.set_boundary_per_reach(A.DOWNSTREAM, [nd, sc(curve), BoundarySpec(A.BC_HOST_ROW)])        # and the three together are accepted
        b.set_boundary_per_reach(A.UPSTREAM, [flow, flow, flow])
        with pytest.raises(E, match="not an FS_BC_HOST_ROW boundary"):
            b.set_host_rows(A.UPSTREAM, np.zeros(3), np.ones(3), np.zeros(3))
        b.set_host_rows(A.DOWNSTREAM, np.zeros(3), np.ones(3), np.zeros(3))


def test_a_team_that_misses_a_member_gives_the_reach_up_instead_of_spinning_on(monkeypatch):
    """The exit condition of the team kernel's wait.  FS_TEAM_TEST_DROP=1 launches one workgroup too few: the last reach's team (two members
    at 8 192 nodes) waits for a member that never starts.  After about eight seconds of polling the waiting member ends its reach with
    FS_TEAM_STALL; the other reaches complete as if nothing had happened, the launch returns, and the next launch on the handle works."""
    import time
    from flowsim_amd import _abi as A
    from synth import rect_problem
    probs = [rect_problem(8192, seed=1700 + s, n_steps=2) for s in range(4)]
    with _uniform_batch(probs, "rect_uniform", 8192) as good:
        good.step(2)
        want = good.hydrographs(0, 3)
        assert np.all(good.status() == 0) and A.kernel_table()[good.kernel_index()]["team"] == 1
    with _uniform_batch(probs, "rect_uniform", 8192) as b:
        monkeypatch.setenv("FS_TEAM_TEST_DROP", "1")
        t0 = time.time()
        b.step(2)
        waited = time.time() - t0
        monkeypatch.delenv("FS_TEAM_TEST_DROP")
        st = b.status()
        assert list(st) == [0, 0, 0, A.TEAM_STALL], st
        assert waited < 120.0, waited                             # a bounded wait, not a hang (no lower bound: the boxes' wall clocks run slow)
        assert np.array_equal(b.hydrographs(0, 3)[:, :, :3], want[:, :, :3])
    with _uniform_batch(probs, "rect_uniform", 8192) as c:       # nothing of it sticks to the device
        c.step(2)
        assert np.all(c.status() == 0) and np.array_equal(c.hydrographs(0, 3), want)
