"""Polyline (IrregularSection) reaches LONGER than one lane grid on preissmann_long_kernel<double, IRREGULAR, 4, 4, -1>.

Round 3 only forced a 13-node fixture onto that entry: one pass, so its multi-pass sweeps, the LDS transposes of the state and
the per-pass stage-table lookups never ran against an oracle.  Here: two channels of 3 000 and 9 000 nodes (3 and 9 passes of
1 024 rows) in ONE batch with per-reach polylines and stage tables - one valley with a single thalweg, one with a secondary
channel behind a levee (two wetted runs at low stages: the sub-section walk of cross_section.py:372-447) - against
oracle/preissmann_oracle.py (polyline nodes: oracle/irregular_oracle.py, pinned to the reference's IrregularSection by the probe
tables and the irr_* / sweep fixtures): 1e-8 with identical Newton counts, and chunked stepping equal to one launch bit for bit."""
import numpy as np
import pytest

from oracle import preissmann_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-8


def rel_err(got, want, floor):
    return float(np.max(np.abs(got - want) / np.maximum(np.abs(want), floor)))


def long_polyline_problem(N, levee, seed, n_steps):
    """A valley whose cross-section changes along the reach: every node its own polyline, blended station by station between an
    upstream and a downstream shape (what interpolate_cross_section does for two IrregularSections of equal vertex count,
    cross_section.py:932-968), on a bed of slope S0."""
    from oracle import irregular_oracle as IO
    rng = np.random.default_rng(seed)
    S0, dx = 2.5e-4, 120.0
    if levee:       # main channel, levee crest at 2.6 m, secondary channel behind it
        xa = np.array([0, 8, 12, 28, 32, 44, 50, 62, 70, 84.0]); za = np.array([7, 3.2, 0.5, 0.0, 0.7, 2.6, 1.1, 1.3, 2.9, 7.0])
        xb = xa * 1.12; zb = za * 0.94
        lim = (2, 5)
    else:
        xa = np.array([0, 10, 14, 30, 34, 60, 66, 80.0]); za = np.array([8, 3.0, 0.4, 0.0, 0.6, 2.5, 2.8, 8.0])
        xb = np.array([0, 12, 18, 33, 41, 58, 70, 90.0]); zb = np.array([7.5, 2.6, 0.3, 0.0, 0.5, 2.0, 2.6, 7.5])
        lim = (2, 5)
    P = len(xa)
    w = (np.arange(N) / (N - 1))[:, None]
    # a slow undulation on top of the blend so that neighbouring nodes sit in different stage-table intervals here and there
    wob = 1.0 + 0.03 * np.sin(np.arange(N) / 37.0)[:, None]
    x = (1 - w) * xa[None, :] + w * xb[None, :]
    zshape = ((1 - w) * za[None, :] + w * zb[None, :]) * wob
    bed = S0 * dx * (N - 1 - np.arange(N))
    z = bed[:, None] + zshape
    geo = {k: np.zeros(N) for k in O.GEO_KEYS}
    geo["z_bed"] = z.min(axis=1)
    geo["n_main"] = 0.030 + 0.004 * w[:, 0]; geo["n_left"] = np.full(N, 0.05); geo["n_right"] = 0.06 - 0.005 * w[:, 0]
    geo["irr_x"], geo["irr_z"] = x, z
    geo["irr_npts"] = np.full(N, P, dtype=np.int32)
    geo["irr_limits"] = np.stack([x[:, lim[0]], x[:, lim[1]]], axis=1)
    h0 = 2.0 if levee else 1.8           # (levee: below the crest - both channels carry water, two wetted runs)
    rough = lambda i: (geo["n_left"][i], geo["n_main"][i], geo["n_right"][i], *geo["irr_limits"][i])
    # the flow the downstream section carries at that depth (normal depth): near-uniform start, the Newton loop does the rest
    Q0 = float(IO.friction(x[-1], z[-1], rough(N - 1), h0, 1.0)[3] * np.sqrt(S0))
    nt = n_steps + 1
    dt = 300.0
    tgt = Q0 * (1.0 + 0.6 * np.sin(np.pi * np.arange(nt) / 8.0) ** 2 * float(rng.uniform(0.8, 1.2)))
    us = O.BC("flow_hydrograph", bed_level=float(geo["z_bed"][0]), target=tgt)
    ds = O.BC("normal_depth", bed_level=float(geo["z_bed"][-1]), bed_slope=S0)
    return O.Problem(geo=geo, h0=np.full(N, h0), Q0=np.full(N, Q0), us=us, ds=ds, theta=0.7, dt=dt, dx=dx, nt=nt, tol=1e-6)


def per_reach_polyline_batch(probs, history=True):
    """ONE batch, every reach its own polylines (and stage tables), node count and boundary parameters; rows of a shorter reach
    are padded by repeating its last node (include/flowsim_abi.h: fs_batch_set_geometry_irregular_per_reach)."""
    from fixture_batch import boundary_spec
    from flowsim_amd import PreissmannBatch
    from flowsim_amd import _abi as A
    B, N, L = len(probs), max(p.N for p in probs), max(p.nt for p in probs)
    P = max(p.geo["irr_x"].shape[1] for p in probs)
    p0 = probs[0]

    def pad_nodes(a):
        a = np.asarray(a)
        return np.concatenate([a, np.repeat(a[-1:], N - len(a), axis=0)], axis=0)

    def pad_pts(a, n_pts):          # [n, p] -> [n, P]: columns beyond a node's own count repeat its last station
        a = np.asarray(a, dtype=np.float64)
        return np.concatenate([a, np.repeat(a[:, -1:], P - a.shape[1], axis=1)], axis=1) if a.shape[1] < P else a
    geo = {k: np.stack([pad_nodes(p.geo[k]) for p in probs]) for k in A.GEO_ROWS}
    geo["irr_npts"] = np.stack([pad_nodes(p.geo["irr_npts"]) for p in probs]).astype(np.int32)
    geo["irr_x"] = np.stack([pad_nodes(pad_pts(p.geo["irr_x"], None)) for p in probs])
    geo["irr_z"] = np.stack([pad_nodes(pad_pts(p.geo["irr_z"], None)) for p in probs])
    geo["irr_limits"] = np.stack([pad_nodes(p.geo["irr_limits"]) for p in probs])
    b = PreissmannBatch(B, N, L, section_mode="irregular", history=history)
    b.set_scheme(p0.theta, p0.dt, p0.dx, p0.tol, p0.max_iter)
    b.set_geometry_irregular(geo)
    b.set_reach_nodes([p.N for p in probs])
    b.set_boundary_per_reach(A.UPSTREAM, [boundary_spec(p.us, p.nt) for p in probs])
    b.set_boundary_per_reach(A.DOWNSTREAM, [boundary_spec(p.ds, p.nt) for p in probs])
    b.set_state(np.stack([pad_nodes(p.h0) for p in probs]), np.stack([pad_nodes(p.Q0) for p in probs]))
    return b


def test_long_polyline_reaches_against_the_oracle():
    from flowsim_amd import _abi as A
    n_steps = 2
    probs = [long_polyline_problem(3000, False, 11, n_steps), long_polyline_problem(9000, True, 12, n_steps)]
    # the levee reach really has two wetted runs somewhere (the path that walks temporary sub-sections)
    from oracle import irregular_oracle as IO
    q = probs[1]
    assert len(IO.subchannels(q.geo["irr_x"][100], q.geo["irr_z"][100], q.h0[100] + q.geo["z_bed"][100])) == 2
    with per_reach_polyline_batch(probs) as b:
        assert b.poly_tables() == 1
        b.step(n_steps)
        e = A.kernel_table()[b.kernel_index()]
        assert (e["long_reach"], e["section_mode"], e["cells_per_thread"], e["waves_per_reach"]) == (1, A.SEC_IRREGULAR, 4, 4)
        assert np.all(b.status() == 0), b.status()
        h, Q = b.history_arrays(0, n_steps + 1)
        its = b.iterations(0, n_steps + 1)
        hyd = b.hydrographs(0, n_steps + 1)
    with per_reach_polyline_batch(probs) as c:            # chunked stepping: level by level, the same bits
        for _ in range(n_steps):
            c.step(1)
        hc, Qc = c.history_arrays(0, n_steps + 1)
        assert np.array_equal(c.iterations(0, n_steps + 1), its)
    for r, p in enumerate(probs):
        assert np.array_equal(hc[:, r, :p.N], h[:, r, :p.N]) and np.array_equal(Qc[:, r, :p.N], Q[:, r, :p.N]), p.N
        ref = O.newton_run(p)
        assert ref["status"] == 0
        assert rel_err(h[:, r, :p.N], ref["depth"], 1e-3) <= TOL, p.N
        assert rel_err(Q[:, r, :p.N], ref["flow"], 1e-3 * abs(p.Q0[0])) <= TOL, p.N
        assert np.array_equal(its[:, r], ref["iters"]), (p.N, its[:, r], ref["iters"])
        assert hyd[-1, 2, r] == h[-1, r, p.N - 1] and hyd[-1, 3, r] == Q[-1, r, p.N - 1]      # the downstream row is the reach's own last node
        assert int(its[1:, r].min()) >= 2                                                  # (a wave is passing: more than one iteration per level)
