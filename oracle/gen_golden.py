#!/usr/bin/env python3
"""Golden-vector generator: runs the *reference* (cve-mohd/flow-sim, mounted read-only at
/root/reference) in THIS container and dumps small .npz fixtures under tests/golden/.

TEST INFRASTRUCTURE ONLY.  Nothing here is imported by the product path.  The reference cannot
travel to the GPU box, so only the *data* it produces (inputs + expected outputs) is committed.

    python oracle/gen_golden.py [--only NAME] [--gerd-steps 48]

What is captured per case (all float64):
  geometry SoA at the computational nodes as the reference interpolated it
  (src/hydromodel/channel.py:213-241, cross_section.py:857-930), initial conditions
  (channel.py:296-390), pre-sampled boundary targets, the solution history depth/flow[nt,N]
  (solver.py:43-44, preissmann.py:166-177), Newton iteration counts and residual norms per time
  level (preissmann.py:122-156), and for the first Newton iteration of the first step the residual
  vector, the 8N-4 Jacobian entries in the reference's row-major order (preissmann.py:322-344)
  and the SuperLU update (preissmann.py:146).

Harness-only accommodations: cases/gerd_roseires hard-codes Windows path
separators and ends in a geopandas export, so the harness runs with cwd=/root/reference, rebinds
`read_csv` in the three case modules to a wrapper that maps '\\' to '/', and passes Q=<array> so
model.run returns before the plotting tail.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
os.environ.setdefault("MPLBACKEND", "Agg")


# --------------------------------------------------------------------------------------------
# spying on the reference's Newton loop without restating it
# --------------------------------------------------------------------------------------------
class Spy:
    """Wraps scipy.sparse.linalg.spsolve as seen by src.hydromodel.preissmann (preissmann.py:146)."""

    def __init__(self, solver):
        import scipy.sparse.linalg as spla
        self.spla = spla
        self.orig = spla.spsolve
        self.solver = solver
        self.norms = []      # (time_level, ||R||) per Newton iteration
        self.first = None    # (R, Jdata, delta) of the very first iteration
        self.x_first = None

    def __enter__(self):
        def wrapped(J, b, *a, **k):
            d = self.orig(J, b, *a, **k)
            self.norms.append((int(self.solver.time_level), float(np.sum(np.square(b)) ** 0.5)))
            if self.first is None:
                self.first = (-np.array(b, dtype=np.float64), np.array(J.data, dtype=np.float64),
                              np.array(d, dtype=np.float64))
            return d
        self.spla.spsolve = wrapped
        return self

    def __exit__(self, *exc):
        self.spla.spsolve = self.orig


def section_soa(channel):
    """Flatten the per-node section objects: TrapezoidalSection attributes (cross_section.py:569-613)
    and, for IrregularSection nodes (cross_section.py:207-243), the polyline padded with NaN."""
    xs = channel.xs_at_node
    irr = [hasattr(s, "x") for s in xs]
    def arr(f, g=lambda s: 0.0):
        return np.array([g(s) if hasattr(s, "x") else f(s) for s in xs], dtype=np.float64)
    out = dict(
        z_bed=arr(lambda s: s.z_bed, lambda s: s.z_min), b_main=arr(lambda s: s.b_main), m_main=arr(lambda s: s.m_main),
        n_main=arr(lambda s: s.n_main, lambda s: s.n_main), n_left=arr(lambda s: s.n_left, lambda s: s.n_left),
        n_right=arr(lambda s: s.n_right, lambda s: s.n_right),
        is_compound=arr(lambda s: 1.0 if s._is_compound else 0.0),
        is_rect=arr(lambda s: 1.0 if s._is_rect else 0.0),
        h_bf=arr(lambda s: s.bankfull_depth if s._is_compound else 0.0),
        b_fp_l=arr(lambda s: s.b_fp_left), b_fp_r=arr(lambda s: s.b_fp_right), m_fp=arr(lambda s: s.m_fp),
        curvature=arr(lambda s: s.curvature, lambda s: s.curvature),
        bed_slope=arr(lambda s: np.nan if s.bed_slope is None else s.bed_slope,
                      lambda s: np.nan if s.bed_slope is None else s.bed_slope),
        chainage=np.asarray(channel.ch_at_node, dtype=np.float64),
    )
    if any(irr):
        P = max(s.x.size for s in xs if hasattr(s, "x"))
        X = np.full((len(xs), P), np.nan); Z = np.full((len(xs), P), np.nan)
        cnt = np.zeros(len(xs), dtype=np.int32)
        lim = np.zeros((len(xs), 2))
        for i, s in enumerate(xs):
            if hasattr(s, "x"):
                cnt[i] = s.x.size; X[i, :cnt[i]] = s.x; Z[i, :cnt[i]] = s.z
                lim[i] = (s.left_fp_limit, s.right_fp_limit)
        out.update(irr_x=X, irr_z=Z, irr_npts=cnt, irr_limits=lim)
    return out


def run_and_capture(solver, tolerance, max_iter=100, slim=False):
    """Calls the reference's own PreissmannSolver.run (preissmann.py:101-163).  slim: benchmark-size fixtures keep
    geometry, initial conditions, the solution, the Newton counts / norms and the first residual vector only (no
    Jacobian entries, no derived fields)."""
    solver.prepare_results = lambda: None          # post-processing is not on the hot path
    ic = np.array(solver.channel.initial_conditions, dtype=np.float64)
    t0 = time.time()
    with Spy(solver) as spy:
        solver.run(tolerance=tolerance, verbose=0, max_iter=max_iter)
    wall = time.time() - t0
    nt = solver.number_of_time_levels
    iters = np.zeros(nt, dtype=np.int32)
    for k, _ in spy.norms:
        iters[k] += 1
    out = dict(
        initial_conditions=ic,
        depth=np.array(solver.depth, dtype=np.float64),
        flow=np.array(solver.flow, dtype=np.float64),
        iters=iters,
        norm_level=np.array([k for k, _ in spy.norms], dtype=np.int32),
        norm_value=np.array([v for _, v in spy.norms], dtype=np.float64),
        R0=spy.first[0], J0=spy.first[1], delta0=spy.first[2],
        final_unknowns=np.array(solver.unknowns, dtype=np.float64),
    )
    out.update({"geo_" + k: v for k, v in section_soa(solver.channel).items()})
    if slim:
        del out["J0"], out["delta0"]
        return out, wall
    # post-processing of the reference (Solver.prepare_results, solver.py:65-127) on the stored solution
    from src.hydromodel.solver import Solver
    Solver.prepare_results(solver)
    for name in ("level", "area", "top_width", "froude_number", "velocity", "wave_celerity", "amplitude", "peak_amplitude"):
        out["derived_" + name] = np.array(getattr(solver, name), dtype=np.float64)
    if hasattr(solver, "storage_outflow"):
        out["derived_storage_outflow"] = np.array(solver.storage_outflow, dtype=np.float64)
        out["derived_storage_stage"] = np.array(solver.storage_stage, dtype=np.float64)
    return out, wall


def save(name, arrays, meta):
    os.makedirs(OUT, exist_ok=True)
    meta = dict(meta)
    meta["generator"] = "oracle/gen_golden.py"
    meta["reference"] = "cve-mohd/flow-sim snapshot 2026-02-13, run in the build container"
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, meta=np.array(json.dumps(meta)), **arrays)
    print(f"  wrote {path}  ({os.path.getsize(path)/1024:.1f} KiB)  N={meta.get('N')} nt={meta.get('nt')}")


def sample_targets(hyd, nt, dt):
    return np.array([hyd.get_at(k * dt) for k in range(nt)], dtype=np.float64)


def base_meta(solver, tol, wall, **kw):
    m = dict(N=int(solver.number_of_nodes), nt=int(solver.number_of_time_levels),
             theta=float(solver.theta), dt=float(solver.time_step), dx=float(solver.spatial_step),
             tolerance=float(tol), max_iter=100, ref_wall_s=wall,
             us_condition=solver.channel.upstream_boundary.condition,
             ds_condition=solver.channel.downstream_boundary.condition,
             us_bed_level=solver.channel.upstream_boundary.bed_level,
             ds_bed_level=solver.channel.downstream_boundary.bed_level,
             us_chainage=float(solver.channel.upstream_boundary.chainage),
             ds_chainage=float(solver.channel.downstream_boundary.chainage),
             initial_flow=float(solver.channel.initial_flow_rate),
             interpolation_method=solver.channel.interpolation_method)
    m.update(kw)
    return m


# --------------------------------------------------------------------------------------------
# cases
# --------------------------------------------------------------------------------------------
def akbari_hydrograph(Q_b, Q_p, t_p, t_b):
    """Shape of cases/akbari_firoozi/settings.py:22-34 with free parameters."""
    from math import sin, cos, pi
    def f(t):
        if t <= t_p:
            return Q_p / 2 * sin(pi * t / t_p - pi / 2) + Q_p / 2 + Q_b
        elif t <= t_b:
            return Q_p / 2 * cos(pi * (t - t_p) / (t_b - t_p)) + Q_p / 2 + Q_b
        return Q_b
    return f


def case_akbari():
    """BASELINE.json configs[1]: cases/akbari_firoozi/main_preissmann.py:7-31 + settings.py."""
    from src.hydromodel.channel import Channel
    from src.hydromodel.boundary import Boundary
    from src.hydromodel.preissmann import PreissmannSolver
    from src.hydromodel.hydrograph import Hydrograph
    from cases.akbari_firoozi import settings as S
    hyd = Hydrograph(S.hydrograph)
    us = Boundary(condition='flow_hydrograph', bed_level=S.S_0 * S.length, chainage=0, hydrograph=hyd)
    ds = Boundary(condition='normal_depth', bed_level=0, chainage=S.length)
    ch = Channel(width=S.width, initial_flow=S.initial_flow, roughness=S.roughness,
                 upstream_boundary=us, downstream_boundary=ds, interpolation_method='steady-state')
    sol = PreissmannSolver(channel=ch, theta=S.theta, time_step=S.preissmann_dt,
                           spatial_step=S.spatial_step, simulation_time=S.duration, regularization=False)
    out, wall = run_and_capture(sol, S.tolerance)
    out["us_target"] = sample_targets(hyd, sol.number_of_time_levels, sol.time_step)
    save("akbari", out, base_meta(sol, S.tolerance, wall, width=S.width, roughness=S.roughness,
                                  length=S.length, S0=S.S_0))


def case_example():
    """BASELINE.json configs[0]: cases/example/main.py:8-57 (fixed_depth + LumpedStorage)."""
    from src.hydromodel.channel import Channel
    from src.hydromodel.boundary import Boundary
    from src.hydromodel.hydrograph import Hydrograph
    from src.hydromodel.lumped_storage import LumpedStorage
    from src.hydromodel.preissmann import PreissmannSolver

    def inflow(t):   # cases/example/main.py:8-29 restated as data (values only)
        q0, qp = 1000.0, 10000.0
        rise, hold, fall = 3 * 3600, 6 * 3600, 4 * 3600
        if t <= 0:
            return q0
        if t < rise:
            return q0 + (qp - q0) * t / rise
        if t - rise < hold:
            return qp
        if t - rise - hold < fall:
            return qp - (qp - q0) * (t - rise - hold) / fall
        return q0

    hyd = Hydrograph(function=inflow)
    us = Boundary(condition='flow_hydrograph', bed_level=5, chainage=0, hydrograph=hyd)
    ds = Boundary(condition='fixed_depth', initial_depth=5, bed_level=0, chainage=20000)
    ss = LumpedStorage(surface_area=5000 * 250, min_stage=5, solution_boundaries=(0, 200))
    ds.set_lumped_storage(ss)
    ch = Channel(width=250, initial_flow=hyd.get_at(0), roughness=0.027,
                 upstream_boundary=us, downstream_boundary=ds)
    sol = PreissmannSolver(channel=ch, theta=0.8, time_step=3600, spatial_step=1000,
                           simulation_time=24 * 3600)
    out, wall = run_and_capture(sol, 1e-4)
    out["us_target"] = sample_targets(hyd, sol.number_of_time_levels, sol.time_step)
    out["storage_stage"] = np.array(ss.stage_hydrograph, dtype=np.float64)[1:]   # level k -> (t, stage); [0] is the t=0 row prepare_results inserted
    save("example", out, base_meta(sol, 1e-4, wall, width=250, roughness=0.027,
                                   storage_area=5000.0 * 250, storage_min_stage=5.0,
                                   storage_bounds=[0, 200], ds_initial_depth=5.0))


def synthetic_rect(name, B, N, n_steps, seed, theta=0.6, dt=600, dx=250.0, tol=1e-6, slim=False):
    """SURVEY 8(d) C3 generator at a small shape, built through the reference's public API
    (Channel(width=, roughness=) -> provisional rectangular sections, channel.py:282-294)."""
    from src.hydromodel.channel import Channel
    from src.hydromodel.boundary import Boundary
    from src.hydromodel.preissmann import PreissmannSolver
    from src.hydromodel.hydrograph import Hydrograph
    rng = np.random.default_rng(seed)
    keys = None
    stack = {}
    params = []
    for r in range(B):
        b = rng.uniform(50, 300); n = rng.uniform(0.02, 0.04); S0 = rng.uniform(2e-4, 1e-3)
        Qb = rng.uniform(50, 500) * (b / 100)
        L = (N - 1) * dx
        hyd = Hydrograph(akbari_hydrograph(Qb, 2 * Qb, 5 * 3600.0, 15 * 3600.0))
        us = Boundary(condition='flow_hydrograph', bed_level=S0 * L, chainage=0, hydrograph=hyd)
        ds = Boundary(condition='normal_depth', bed_level=0, chainage=L)
        ch = Channel(width=b, initial_flow=Qb, roughness=n, upstream_boundary=us,
                     downstream_boundary=ds, interpolation_method='steady-state')
        sol = PreissmannSolver(channel=ch, theta=theta, time_step=dt, spatial_step=dx,
                               simulation_time=n_steps * dt)
        out, wall = run_and_capture(sol, tol, slim=slim)
        out["us_target"] = sample_targets(hyd, sol.number_of_time_levels, sol.time_step)
        params.append([b, n, S0, Qb])
        for k, v in out.items():
            stack.setdefault(k, []).append(v)
        meta = base_meta(sol, tol, wall, seed=seed, B=B)
    arrays = {}
    for k, v in stack.items():
        if k in ("norm_level", "norm_value"):
            arrays[k] = np.concatenate(v)
            arrays[k + "_offsets"] = np.cumsum([0] + [len(x) for x in v]).astype(np.int32)
        else:
            arrays[k] = np.stack(v)
    arrays["params"] = np.array(params)          # [B, (b, n, S0, Q_base)]
    save(name, arrays, meta)


def synthetic_trap(name, B, N, n_steps, seed, theta=0.6, dt=1800, dx=500.0, tol=1e-6, slim=False):
    """SURVEY 8(d) C5 generator at a small shape in fp64: simple trapezoid + power rating curve."""
    from src.hydromodel.channel import Channel
    from src.hydromodel.boundary import Boundary
    from src.hydromodel.preissmann import PreissmannSolver
    from src.hydromodel.hydrograph import Hydrograph
    from src.hydromodel.rating_curve import RatingCurve
    from src.hydromodel.cross_section import TrapezoidalSection
    rng = np.random.default_rng(seed)
    stack = {}
    params = []
    for r in range(B):
        b = rng.uniform(20, 100); m = rng.uniform(1, 3); n = rng.uniform(0.025, 0.04)
        S0 = rng.uniform(2e-4, 1e-3); Qb = rng.uniform(50, 500) * (b / 100)
        L = (N - 1) * dx
        xs_u = TrapezoidalSection(z_bed=S0 * L, b_main=b, m_main=m, n_main=n, bed_slope=S0)
        xs_d = TrapezoidalSection(z_bed=0.0, b_main=b, m_main=m, n_main=n, bed_slope=S0)
        h_n = xs_d.normal_depth(Q_target=Qb)
        be = 1.6
        a = Qb / h_n ** be
        rc = RatingCurve(); rc.set(type='power', a=a, b=be)
        hyd = Hydrograph(akbari_hydrograph(Qb, 2 * Qb, 5 * 3600.0, 15 * 3600.0))
        us = Boundary(condition='flow_hydrograph', bed_level=S0 * L, chainage=0, hydrograph=hyd)
        ds = Boundary(condition='rating_curve', bed_level=0.0, chainage=L, initial_depth=h_n, rating_curve=rc)
        ch = Channel(initial_flow=Qb, upstream_boundary=us, downstream_boundary=ds,
                     interpolation_method='steady-state')
        ch.set_cross_sections([0.0, L], [xs_u, xs_d])
        sol = PreissmannSolver(channel=ch, theta=theta, time_step=dt, spatial_step=dx,
                               simulation_time=n_steps * dt)
        out, wall = run_and_capture(sol, tol, slim=slim)
        out["us_target"] = sample_targets(hyd, sol.number_of_time_levels, sol.time_step)
        params.append([b, m, n, S0, Qb, h_n, a, be])
        for k, v in out.items():
            stack.setdefault(k, []).append(v)
        meta = base_meta(sol, tol, wall, seed=seed, B=B)
    arrays = {}
    for k, v in stack.items():
        if k in ("norm_level", "norm_value"):
            arrays[k] = np.concatenate(v)
            arrays[k + "_offsets"] = np.cumsum([0] + [len(x) for x in v]).astype(np.int32)
        else:
            arrays[k] = np.stack(v)
    arrays["params"] = np.array(params)          # [B, (b, m, n, S0, Q_base, h_n, rc_a, rc_b)]
    save(name, arrays, meta)


def case_bc_matrix():
    """Edge cases of the boundary plugin surface (boundary.py:56-242): stage_hydrograph upstream,
    fixed_depth without storage, polynomial rating curve, linear IC, compound section without
    curvature, GVF IC on a simple trapezoid."""
    from src.hydromodel.channel import Channel
    from src.hydromodel.boundary import Boundary
    from src.hydromodel.preissmann import PreissmannSolver
    from src.hydromodel.hydrograph import Hydrograph
    from src.hydromodel.rating_curve import RatingCurve
    from src.hydromodel.cross_section import TrapezoidalSection

    # (a) stage hydrograph upstream (table, np.interp) + fixed depth downstream, linear IC, rectangle
    L = 12000.0
    tab = np.array([[0, 8.0 + 3.0], [3600 * 2, 8.0 + 3.6], [3600 * 5, 8.0 + 3.1], [3600 * 12, 8.0 + 3.0]])
    hyd = Hydrograph(table=tab)
    us = Boundary(condition='stage_hydrograph', bed_level=8.0, chainage=0, initial_depth=3.0, hydrograph=hyd)
    ds = Boundary(condition='fixed_depth', bed_level=0.0, chainage=L, initial_depth=4.0)
    ch = Channel(width=80, initial_flow=300.0, roughness=0.03, upstream_boundary=us,
                 downstream_boundary=ds, interpolation_method='linear')
    sol = PreissmannSolver(channel=ch, theta=0.7, time_step=900, spatial_step=500, simulation_time=8 * 3600)
    out, wall = run_and_capture(sol, 1e-6)
    out["us_target"] = sample_targets(hyd, sol.number_of_time_levels, sol.time_step)
    out["us_table"] = tab
    save("bc_stage_fixed", out, base_meta(sol, 1e-6, wall, width=80, roughness=0.03,
                                          us_initial_depth=3.0, ds_initial_depth=4.0))

    # (b) simple trapezoid, GVF IC, polynomial rating curve downstream with a stage shift attribute
    L = 15000.0; S0 = 4e-4
    xs_u = TrapezoidalSection(z_bed=100.0 + S0 * L, b_main=40.0, m_main=1.5, n_main=0.032, bed_slope=S0)
    xs_m = TrapezoidalSection(z_bed=100.0 + S0 * L * 0.4, b_main=55.0, m_main=2.0, n_main=0.028, bed_slope=S0)
    xs_d = TrapezoidalSection(z_bed=100.0, b_main=60.0, m_main=2.5, n_main=0.03, bed_slope=S0)
    rc = RatingCurve(); rc.set(type='polynomial', a=9.0, b=35.0, c=-20.0)
    rc.stage_shift = -100.0
    h_ds = 3.0
    Q0 = rc.discharge(100.0 + h_ds)
    hyd = Hydrograph(akbari_hydrograph(Q0, 1.5 * Q0, 3 * 3600.0, 9 * 3600.0))
    us = Boundary(condition='flow_hydrograph', bed_level=xs_u.z_bed, chainage=0, hydrograph=hyd)
    ds = Boundary(condition='rating_curve', bed_level=100.0, chainage=L, initial_depth=h_ds, rating_curve=rc)
    ch = Channel(initial_flow=Q0, upstream_boundary=us, downstream_boundary=ds)      # GVF default
    ch.set_cross_sections([0.0, 0.6 * L, L], [xs_u, xs_m, xs_d])
    sol = PreissmannSolver(channel=ch, theta=0.6, time_step=1200, spatial_step=600, simulation_time=10 * 3600)
    out, wall = run_and_capture(sol, 1e-6)
    out["us_target"] = sample_targets(hyd, sol.number_of_time_levels, sol.time_step)
    save("bc_trap_poly", out, base_meta(sol, 1e-6, wall, rc_type='polynomial', rc_a=9.0, rc_b=35.0, rc_c=-20.0,
                                        rc_shift=-100.0, ds_initial_depth=h_ds,
                                        xs_chainages=[0.0, 0.6 * L, L],
                                        xs_params=[[s.z_bed, s.b_main, s.m_main, s.n_main, S0] for s in (xs_u, xs_m, xs_d)]))

    # (c) compound trapezoid going over bank (exercises SURVEY F3 bug-compatibility), no curvature,
    #     normal-depth downstream, steady IC
    L = 20000.0; S0 = 3e-4
    def comp(z, b, m, hb, bl, br, mf, slope):
        return TrapezoidalSection(z_bed=z, b_main=b, m_main=m, n_main=0.03, z_bank=z + hb, b_fp_left=bl,
                                  b_fp_right=br, m_fp=mf, n_left=0.06, n_right=0.05, bed_slope=slope)
    xs_u = comp(S0 * L, 30.0, 2.0, 2.5, 60.0, 40.0, 4.0, S0)
    xs_d = comp(0.0, 36.0, 1.5, 2.2, 80.0, 50.0, 3.0, S0)
    Qb = 120.0
    hyd = Hydrograph(akbari_hydrograph(Qb, 500.0, 2 * 3600.0, 7 * 3600.0))
    us = Boundary(condition='flow_hydrograph', bed_level=S0 * L, chainage=0, hydrograph=hyd)
    ds = Boundary(condition='normal_depth', bed_level=0.0, chainage=L)
    ch = Channel(initial_flow=Qb, upstream_boundary=us, downstream_boundary=ds, interpolation_method='steady-state')
    ch.set_cross_sections([0.0, L], [xs_u, xs_d])
    sol = PreissmannSolver(channel=ch, theta=0.65, time_step=600, spatial_step=500, simulation_time=6 * 3600)
    out, wall = run_and_capture(sol, 1e-6)
    out["us_target"] = sample_targets(hyd, sol.number_of_time_levels, sol.time_step)
    over = (out["depth"] > out["geo_h_bf"][None, :]).mean()
    save("bc_compound_normal", out, base_meta(sol, 1e-6, wall, overbank_fraction=float(over)))


def case_irregular():
    """IrregularSection channels (cross_section.py:207-543) incl. the mixed
    interpolation path (:932-968).  Three runs: single thalweg; a secondary channel behind a levee
    with composite roughness (sub-channel conveyance path, :372-447); trapezoid -> polyline mix with
    centre-line curvature."""
    from src.hydromodel.channel import Channel
    from src.hydromodel.boundary import Boundary
    from src.hydromodel.preissmann import PreissmannSolver
    from src.hydromodel.hydrograph import Hydrograph
    from src.hydromodel.rating_curve import RatingCurve
    from src.hydromodel.cross_section import TrapezoidalSection, IrregularSection

    def probe(sol, out, levels):
        """Section functions of the reference at a few stages per node (pins the oracle's section
        restatement independently of the Newton loop)."""
        xs = sol.channel.xs_at_node
        rows = []
        for i, s in enumerate(xs):
            for hw_rel in levels:
                hw = s.z_min + hw_rel
                h = hw_rel
                Q = 75.0
                s._last_hw = None; s._last_hw_n = None
                rows.append([i, h, Q, s.area(hw), s.wetted_perimeter(hw), s.top_width(hw), s.dA_dh(hw),
                             s.get_equivalent_n(hw), s.conveyance(hw), s.dR_dA(hw), s.dK_dA(hw),
                             s.friction_slope(h, Q), s.dSf_dA(h, Q), s.dSf_dQ(h, Q),
                             s.curvature_slope(h, Q), s.dSc_dA(h, Q), s.dSc_dQ(h, Q)])
        out["probe"] = np.array(rows, dtype=np.float64)

    def run(name, ch, theta, dt, dx, T, hyd, tol=1e-6, levels=(0.3, 0.9, 1.7, 2.35, 3.1, 4.4), **kw):
        sol = PreissmannSolver(channel=ch, theta=theta, time_step=dt, spatial_step=dx, simulation_time=T)
        out, wall = run_and_capture(sol, tol)
        out["us_target"] = sample_targets(hyd, sol.number_of_time_levels, sol.time_step)
        probe(sol, out, levels)
        save(name, out, base_meta(sol, tol, wall, **kw))

    # (a) single thalweg, two different polylines (union of stations at the interpolated nodes)
    L = 6000.0; S0 = 2e-4
    xu = np.array([0, 10, 14, 30, 34, 60, 66, 80.0]); zu = np.array([8, 3.0, 0.4, 0.0, 0.6, 2.5, 2.8, 8.0])
    xd = np.array([0, 12, 18, 33, 41, 58, 70, 90.0]); zd = np.array([7.5, 2.6, 0.3, 0.0, 0.5, 2.0, 2.6, 7.5])
    xs_u = IrregularSection(x=xu, z=S0 * L + zu, n=0.03, bed_slope=S0)
    xs_d = IrregularSection(x=xd, z=zd, n=0.034, bed_slope=S0)
    Qb = 40.0
    hyd = Hydrograph(akbari_hydrograph(Qb, 260.0, 1 * 3600.0, 3 * 3600.0))
    us = Boundary(condition='flow_hydrograph', bed_level=S0 * L, chainage=0, hydrograph=hyd)
    ds = Boundary(condition='normal_depth', bed_level=0.0, chainage=L)
    ch = Channel(initial_flow=Qb, upstream_boundary=us, downstream_boundary=ds, interpolation_method='steady-state')
    ch.set_cross_sections([0.0, L], [xs_u, xs_d])
    run("irr_single", ch, 0.65, 300, 500, 3 * 3600, hyd)

    # (b) secondary channel behind a levee + composite roughness; stage rises through the crest
    L = 5000.0; S0 = 3e-4
    x = np.array([0, 8, 12, 28, 32, 44, 50, 62, 70, 84.0])
    z = np.array([7, 3.2, 0.5, 0.0, 0.7, 2.6, 1.1, 1.3, 2.9, 7.0])
    xs_u = IrregularSection(x=x, z=S0 * L + z, n=0.03, bed_slope=S0)
    xs_u.set_roughness_para((0.05, 0.03, 0.06, 12.0, 44.0))
    xs_d = IrregularSection(x=x * 1.1, z=z * 0.95, n=0.032, bed_slope=S0)
    xs_d.set_roughness_para((0.05, 0.032, 0.055, 13.2, 48.4))
    rc = RatingCurve(); rc.set(type='power', a=14.0, b=1.9)
    h_ds = 1.6
    Q0 = rc.discharge(h_ds)
    hyd = Hydrograph(akbari_hydrograph(Q0, 6.0 * Q0, 1.5 * 3600.0, 4 * 3600.0))
    us = Boundary(condition='flow_hydrograph', bed_level=S0 * L, chainage=0, hydrograph=hyd)
    ds = Boundary(condition='rating_curve', bed_level=0.0, chainage=L, initial_depth=h_ds, rating_curve=rc)
    ch = Channel(initial_flow=Q0, upstream_boundary=us, downstream_boundary=ds, interpolation_method='linear')
    us.initial_depth = 1.7
    ch.set_cross_sections([0.0, L], [xs_u, xs_d])
    run("irr_levee", ch, 0.7, 300, 500, 4 * 3600, hyd, rc_type='power', rc_a=14.0, rc_b=1.9,
        ds_initial_depth=h_ds, us_initial_depth=1.7, levels=(0.4, 1.0, 1.25, 1.8, 2.45, 2.75, 3.3, 4.5))

    # (c) trapezoid -> polyline -> polyline with centre-line coordinates (curvature at the middle section)
    L = 8000.0; S0 = 2.5e-4
    xs_u = TrapezoidalSection(z_bed=S0 * L, b_main=30.0, m_main=2.0, n_main=0.03, bed_slope=S0)
    xm = np.array([-40, -25, -16, -6, 5, 17, 26, 41.0]); zm = np.array([6, 2.2, 0.5, 0.0, 0.1, 0.6, 2.4, 6.0])
    xs_m = IrregularSection(x=xm, z=S0 * L * 0.5 + zm, n=0.031, bed_slope=S0)
    xd = np.array([-45, -22, -15, -4, 8, 19, 30, 46.0]); zd = np.array([6, 2.0, 0.4, 0.0, 0.2, 0.7, 2.1, 6.0])
    xs_d = IrregularSection(x=xd, z=zd, n=0.033, bed_slope=S0)
    Qb = 60.0
    hyd = Hydrograph(akbari_hydrograph(Qb, 200.0, 1 * 3600.0, 3 * 3600.0))
    us = Boundary(condition='flow_hydrograph', bed_level=S0 * L, chainage=0, hydrograph=hyd)
    ds = Boundary(condition='normal_depth', bed_level=0.0, chainage=L)
    ch = Channel(initial_flow=Qb, upstream_boundary=us, downstream_boundary=ds, interpolation_method='steady-state')
    ch.set_cross_sections([0.0, 0.5 * L, L], [xs_u, xs_m, xs_d])
    ch.set_coords([[0.0, 0.0], [2500.0, 300.0], [4000.0, 1500.0], [4200.0, 3800.0], [6500.0, 5200.0]],
                  [0.0, 2500.0, 4400.0, 6700.0, L])
    run("irr_mixed", ch, 0.65, 300, 500, 3 * 3600, hyd)


def case_storage_general():
    """fixed_depth behind a LumpedStorage with an area curve, a reservoir rating
    curve and entrance losses (lumped_storage.py:24-179, boundary.py:97-133, :152-164, :213-237):
    the mass-balance root needs brentq, the interface stage carries friction + empirical head loss."""
    from src.hydromodel.channel import Channel
    from src.hydromodel.boundary import Boundary
    from src.hydromodel.hydrograph import Hydrograph
    from src.hydromodel.lumped_storage import LumpedStorage
    from src.hydromodel.preissmann import PreissmannSolver
    from src.hydromodel.rating_curve import RatingCurve
    from src.hydromodel.cross_section import TrapezoidalSection

    def run(name, rc_type, losses, trap):
        L = 10000.0
        Qb = 200.0
        hyd = Hydrograph(akbari_hydrograph(Qb, 900.0, 3 * 3600.0, 9 * 3600.0))
        us = Boundary(condition='flow_hydrograph', bed_level=2.0, chainage=0, hydrograph=hyd)
        ds = Boundary(condition='fixed_depth', initial_depth=5, bed_level=0, chainage=L)
        stages = np.arange(0.0, 30.01, 0.5)
        curve = np.column_stack([stages, 2.0e5 + 4.0e4 * stages + 900.0 * stages ** 2])
        rc = None
        meta = dict(storage_curve=curve.tolist(), storage_alpha=1.1, storage_beta=0.2, storage_min_stage=4.0,
                    ds_initial_depth=5.0, storage_rc_type=rc_type)
        if rc_type == 'polynomial':
            rc = RatingCurve(); rc.set(type='polynomial', a=6.0, b=10.0, c=0.0)
            meta.update(storage_rc=dict(a=6.0, b=10.0, c=0.0, shift=0.0))
        elif rc_type == 'power':
            rc = RatingCurve(); rc.set(type='power', a=17.0, b=1.5)
            meta.update(storage_rc=dict(a=17.0, b=1.5, shift=0.0))
        ss = LumpedStorage(surface_area=None, min_stage=4.0, solution_boundaries=(0, 30), rating_curve=rc)
        ss.set_area_curve(curve, alpha=1.1, beta=0.2)
        if losses:
            ss.capture_losses = True
            ss.reservoir_length = 800.0
            ss.K_q = 0.3
            meta.update(storage_losses=dict(reservoir_length=800.0, K_q=0.3))
        meta["storage_bounds"] = [float(ss.Y_min), float(ss.Y_max)]
        ds.set_lumped_storage(ss)
        if trap:
            xs_u = TrapezoidalSection(z_bed=2.0, b_main=90.0, m_main=2.0, n_main=0.028, bed_slope=2e-4)
            xs_d = TrapezoidalSection(z_bed=0.0, b_main=110.0, m_main=2.5, n_main=0.03, bed_slope=2e-4)
            ch = Channel(initial_flow=Qb, upstream_boundary=us, downstream_boundary=ds)
            ch.set_cross_sections([0.0, L], [xs_u, xs_d])
        else:
            ch = Channel(width=100, initial_flow=Qb, roughness=0.03, upstream_boundary=us, downstream_boundary=ds)
        sol = PreissmannSolver(channel=ch, theta=0.7, time_step=1200, spatial_step=500, simulation_time=12 * 3600)
        out, wall = run_and_capture(sol, 1e-6)
        out["us_target"] = sample_targets(hyd, sol.number_of_time_levels, sol.time_step)
        out["storage_stage"] = np.array(ss.stage_hydrograph, dtype=np.float64)[1:]
        save(name, out, base_meta(sol, 1e-6, wall, **meta))

    run("storage_curve_poly_losses", 'polynomial', True, False)
    run("storage_curve_power_trap", 'power', True, True)
    run("storage_curve_closed", None, False, False)


def _gerd_imports():
    """cwd + read_csv shim for cases/gerd_roseires (harness only)."""
    os.chdir(REF)
    import pandas
    real = pandas.read_csv
    def shim(path, *a, **k):
        if isinstance(path, str):
            path = path.replace("\\", "/")
        return real(path, *a, **k)
    import cases.gerd_roseires.gerd_discharge as gd
    import cases.gerd_roseires.custom_functions as cf
    import cases.gerd_roseires.roseires_rating_curve as rr
    gd.read_csv = shim; cf.read_csv = shim; rr.read_csv = shim
    return gd, cf, rr


def rating_spec(rc, rr):
    """Flatten RoseiresRatingCurve (roseires_rating_curve.py:65-109, :202-257) to data: both gate
    states reduce to a quadratic in stage because the regressions are degree-2 in (stage, x)."""
    def quad_of_state(state):
        openings, n_sl = state
        sp = rc.spillway_model; sl = rc.sluice_model
        cs = sp.named_steps["linreg"].coef_; i_s = sp.named_steps["linreg"].intercept_
        cl = sl.named_steps["linreg"].coef_; i_l = sl.named_steps["linreg"].intercept_
        # features: [x0, x1, x0^2, x0 x1, x1^2] with x0 = stage
        c0 = c1 = c2 = 0.0
        for o in openings:
            if o > 0:
                c0 += i_s + cs[1] * o + cs[4] * o * o
                c1 += cs[0] + cs[3] * o
                c2 += cs[2]
        t = rc.tail_water_level
        c0 += n_sl * (i_l + cl[1] * t + cl[4] * t * t)
        c1 += n_sl * (cl[0] + cl[3] * t)
        c2 += n_sl * cl[2]
        c0 += rr.HYDROPOWER_Q
        return [float(c0), float(c1), float(c2)]
    return dict(initial_stage=float(rc.initial_stage), buffer=float(rc.buffer),
                low=quad_of_state(rc.closed_state), high=quad_of_state(rc.open_state),
                closed_state=[list(map(float, rc.closed_state[0])), int(rc.closed_state[1])],
                open_state=[list(map(float, rc.open_state[0])), int(rc.open_state[1])],
                spill_coef=[float(x) for x in rc.spillway_model.named_steps["linreg"].coef_],
                spill_icpt=float(rc.spillway_model.named_steps["linreg"].intercept_),
                sluice_coef=[float(x) for x in rc.sluice_model.named_steps["linreg"].coef_],
                sluice_icpt=float(rc.sluice_model.named_steps["linreg"].intercept_),
                tail_water_level=float(rc.tail_water_level), hydropower_q=float(rr.HYDROPOWER_Q), dY=1e-3)


def build_gerd(gd, cf, rr, n_main, sim_hours, inflow_csv, coords):
    """Restates cases/gerd_roseires/model.py:36-92 in the harness (accommodation iv)."""
    from src.hydromodel.channel import Channel
    from src.hydromodel.boundary import Boundary
    from src.hydromodel.hydrograph import Hydrograph
    from src.hydromodel.preissmann import PreissmannSolver
    from cases.gerd_roseires import settings as S
    inflow = Hydrograph(table=cf.import_hydrograph(inflow_csv))
    duration = int(inflow.table[-1, 0]) if sim_hours is None else int(sim_hours * 3600)
    rel = gd.GerdHydrograph()
    rel.build(inflow_hydrograph=inflow, time_step=S.time_step, duration=duration, initial_stage=S.initial_gerd_level)
    Q0 = rel.get_at(time=0)
    chs, secs = cf.load_trapzoid_xs(file_path=S.cross_sections_path, n_fp=None, n_main=n_main)
    bed = secs[-1].z_min
    us = Boundary(condition='flow_hydrograph', hydrograph=rel, chainage=chs[0])
    rc = rr.RoseiresRatingCurve(initial_stage=S.initial_roseires_level, initial_flow=Q0,
                                jammed_sluice_gates=0, jammed_spillways=0)
    ds = Boundary(initial_depth=S.initial_roseires_level - bed, bed_level=bed, condition='rating_curve',
                  rating_curve=rc, chainage=chs[-1])
    ch = Channel(initial_flow=Q0, upstream_boundary=us, downstream_boundary=ds)
    if coords:
        c = cf.import_table(S.coords_path, sort_by='chainage')
        ch.set_coords(coords=c[:, 1:], chainages=c[:, 0])
    ch.set_cross_sections(chainages=chs, sections=secs)
    sol = PreissmannSolver(channel=ch, theta=S.theta, time_step=S.time_step, spatial_step=S.spatial_step,
                           simulation_time=duration)
    return sol, rel, rc, chs, secs


def case_gerd(steps):
    """BASELINE.json configs[3] geometry: cases/gerd_roseires (compound sections + curvature +
    Roseires gate rating curve), first `steps` hours of the 384 h run of main.py:4."""
    gd, cf, rr = _gerd_imports()
    from cases.gerd_roseires import settings as S
    sol, rel, rc, chs, secs = build_gerd(gd, cf, rr, None, steps, "cases/gerd_roseires/data/inflow_hydrograph.csv", True)
    out, wall = run_and_capture(sol, S.tolerance)
    out["us_target"] = np.array(rel.table[:sol.number_of_time_levels, 1], dtype=np.float64)
    out["rating_probe_stage"] = np.linspace(486.0, 489.0, 61)
    out["rating_probe_Q"] = np.array([rc.discharge(s) for s in out["rating_probe_stage"]])
    out["rating_probe_dQ"] = np.array([rc.dQ_dz(s) for s in out["rating_probe_stage"]])
    out["input_chainages"] = np.array(chs, dtype=np.float64)
    out["input_curvature"] = np.array([s.curvature for s in secs], dtype=np.float64)
    save("gerd", out, base_meta(sol, S.tolerance, wall, rating=rating_spec(rc, rr),
                                ds_initial_depth=float(sol.channel.downstream_boundary.initial_depth)))

    if os.environ.get("GEN_SKIP_ENSEMBLE"):
        os.chdir(os.path.dirname(os.path.abspath(__file__)))
        return
    # Manning-n ensemble: the n_calibrate.py:5-17 setup (small inflow table, no curvature, 32 steps)
    members = np.linspace(0.020, 0.060, 8)
    stack = {}
    for n in members:
        sol, rel, rc, chs, secs = build_gerd(gd, cf, rr, float(n), None,
                                             "cases/gerd_roseires/data/inflow_hydrograph_small.csv", False)
        o, wall = run_and_capture(sol, S.tolerance)
        o["us_target"] = np.array(rel.table[:sol.number_of_time_levels, 1], dtype=np.float64)
        for k in ("depth", "flow", "iters", "initial_conditions", "us_target", "geo_n_main"):
            stack.setdefault(k, []).append(o[k])
        geo = {k: v for k, v in o.items() if k.startswith("geo_") and k != "geo_n_main"}
        spec = rating_spec(rc, rr)
        print(f"   member n_main={n:.4f}: {int(o['iters'].sum())} its, {wall:.1f} s")
    arrays = {k: np.stack(v) for k, v in stack.items()}
    arrays.update(geo)
    arrays["n_members"] = members
    save("gerd_ensemble", arrays, base_meta(sol, S.tolerance, wall, rating=spec, B=len(members),
                                            ds_initial_depth=float(sol.channel.downstream_boundary.initial_depth)))
    os.chdir(os.path.dirname(os.path.abspath(__file__)))


def case_upstream_kinds():
    """The three boundary kinds the other fixtures only ever have downstream, imposed upstream (boundary.py:56-242 serves
    both ends; preissmann.py:200-218, :346-400 call it for node 0), and a flow hydrograph imposed downstream."""
    from src.hydromodel.channel import Channel
    from src.hydromodel.boundary import Boundary
    from src.hydromodel.preissmann import PreissmannSolver
    from src.hydromodel.hydrograph import Hydrograph
    from src.hydromodel.rating_curve import RatingCurve
    from src.hydromodel.cross_section import TrapezoidalSection

    # (a) reservoir level upstream (fixed_depth), gate closing downstream (outflow hydrograph that dips)
    L = 10000.0; S0 = 2e-4; Q0 = 150.0
    xs_u = TrapezoidalSection(z_bed=S0 * L, b_main=40.0, m_main=1.5, n_main=0.03, bed_slope=S0)
    xs_d = TrapezoidalSection(z_bed=0.0, b_main=40.0, m_main=1.5, n_main=0.03, bed_slope=S0)
    hn = xs_d.normal_depth(Q_target=Q0)
    hyd = Hydrograph(akbari_hydrograph(Q0, -60.0, 2 * 3600.0, 6 * 3600.0))
    us = Boundary(condition='fixed_depth', bed_level=S0 * L, chainage=0, initial_depth=hn)
    ds = Boundary(condition='flow_hydrograph', bed_level=0.0, chainage=L, hydrograph=hyd, initial_depth=hn)
    ch = Channel(initial_flow=Q0, upstream_boundary=us, downstream_boundary=ds, interpolation_method='steady-state')
    ch.set_cross_sections([0.0, L], [xs_u, xs_d])
    sol = PreissmannSolver(channel=ch, theta=0.65, time_step=600, spatial_step=500, simulation_time=8 * 3600)
    out, wall = run_and_capture(sol, 1e-6)
    out["ds_target"] = sample_targets(hyd, sol.number_of_time_levels, sol.time_step)
    save("bc_us_fixed_ds_flow", out, base_meta(sol, 1e-6, wall, us_initial_depth=float(hn), ds_initial_depth=float(hn)))

    # (b) head-dependent inflow upstream (rating curve with negative slope: Q falls as the stage rises), tide-like stage
    #     hydrograph downstream, rectangular channel, linear IC
    L = 8000.0; S0 = 3e-4; Q0 = 200.0; h_us = 3.0
    rc = RatingCurve(); rc.set(type='polynomial', a=0.0, b=-60.0, c=Q0 + 60.0 * (S0 * L + h_us))
    tab = np.array([[0, 3.0], [3600 * 2, 3.8], [3600 * 5, 3.2], [3600 * 12, 3.0]])
    hyd = Hydrograph(table=tab)
    us = Boundary(condition='rating_curve', bed_level=S0 * L, chainage=0, initial_depth=h_us, rating_curve=rc)
    ds = Boundary(condition='stage_hydrograph', bed_level=0.0, chainage=L, initial_depth=3.0, hydrograph=hyd)
    ch = Channel(width=60, initial_flow=Q0, roughness=0.03, upstream_boundary=us, downstream_boundary=ds,
                 interpolation_method='linear')
    sol = PreissmannSolver(channel=ch, theta=0.7, time_step=900, spatial_step=500, simulation_time=8 * 3600)
    out, wall = run_and_capture(sol, 1e-6)
    out["ds_target"] = sample_targets(hyd, sol.number_of_time_levels, sol.time_step)
    save("bc_us_rating_ds_stage", out, base_meta(sol, 1e-6, wall, us_initial_depth=h_us, ds_initial_depth=3.0, width=60,
                                                 roughness=0.03, us_rc_type='polynomial', us_rc_a=0.0, us_rc_b=-60.0,
                                                 us_rc_c=float(rc.c), us_rc_shift=0.0))

    # (c) normal depth imposed upstream, stage hydrograph downstream, simple trapezoid
    L = 12000.0; S0 = 4e-4; Q0 = 180.0
    xs_u = TrapezoidalSection(z_bed=S0 * L, b_main=50.0, m_main=2.0, n_main=0.03, bed_slope=S0)
    xs_d = TrapezoidalSection(z_bed=0.0, b_main=50.0, m_main=2.0, n_main=0.03, bed_slope=S0)
    hn = xs_d.normal_depth(Q_target=Q0)
    tab = np.array([[0, hn], [3600 * 2, hn + 0.6], [3600 * 5, hn + 0.1], [3600 * 12, hn]])
    hyd = Hydrograph(table=tab)
    us = Boundary(condition='normal_depth', bed_level=S0 * L, chainage=0, initial_depth=hn)
    ds = Boundary(condition='stage_hydrograph', bed_level=0.0, chainage=L, initial_depth=hn, hydrograph=hyd)
    ch = Channel(initial_flow=Q0, upstream_boundary=us, downstream_boundary=ds, interpolation_method='steady-state')
    ch.set_cross_sections([0.0, L], [xs_u, xs_d])
    sol = PreissmannSolver(channel=ch, theta=0.7, time_step=600, spatial_step=500, simulation_time=6 * 3600)
    out, wall = run_and_capture(sol, 1e-6)
    out["ds_target"] = sample_targets(hyd, sol.number_of_time_levels, sol.time_step)
    save("bc_us_normal_ds_stage", out, base_meta(sol, 1e-6, wall, us_initial_depth=float(hn), ds_initial_depth=float(hn)))


def case_irr_storage():
    """Polyline sections in front of a general LumpedStorage (area curve, polynomial outflow curve, entrance losses): the
    boundary row then needs the polyline's A, R, n_eq, dR/dA, dA/dh at two stages (boundary.py:110-124, :152-164)."""
    from src.hydromodel.channel import Channel
    from src.hydromodel.boundary import Boundary
    from src.hydromodel.hydrograph import Hydrograph
    from src.hydromodel.lumped_storage import LumpedStorage
    from src.hydromodel.preissmann import PreissmannSolver
    from src.hydromodel.rating_curve import RatingCurve
    from src.hydromodel.cross_section import IrregularSection
    L = 6000.0; S0 = 2e-4
    xu = np.array([0, 10, 14, 30, 34, 60, 66, 80.0]); zu = np.array([8, 3.0, 0.4, 0.0, 0.6, 2.5, 2.8, 8.0])
    xd = np.array([0, 12, 18, 33, 41, 58, 70, 90.0]); zd = np.array([7.5, 2.6, 0.3, 0.0, 0.5, 2.0, 2.6, 7.5])
    xs_u = IrregularSection(x=xu, z=S0 * L + zu, n=0.03, bed_slope=S0)
    xs_d = IrregularSection(x=xd, z=zd, n=0.034, bed_slope=S0)
    Qb = 40.0
    hyd = Hydrograph(akbari_hydrograph(Qb, 260.0, 1 * 3600.0, 3 * 3600.0))
    us = Boundary(condition='flow_hydrograph', bed_level=S0 * L, chainage=0, hydrograph=hyd)
    ds = Boundary(condition='fixed_depth', initial_depth=2.2, bed_level=0.0, chainage=L)
    stages = np.arange(0.0, 12.01, 0.25)
    curve = np.column_stack([stages, 6.0e4 + 1.5e4 * stages + 500.0 * stages ** 2])
    rc = RatingCurve(); rc.set(type='polynomial', a=4.0, b=6.0, c=0.0)
    ss = LumpedStorage(surface_area=None, min_stage=1.0, solution_boundaries=(0, 12), rating_curve=rc)
    ss.set_area_curve(curve, alpha=1.0, beta=0.0)
    ss.capture_losses = True; ss.reservoir_length = 300.0; ss.K_q = 0.2
    ds.set_lumped_storage(ss)
    ch = Channel(initial_flow=Qb, upstream_boundary=us, downstream_boundary=ds)
    ch.set_cross_sections([0.0, L], [xs_u, xs_d])
    sol = PreissmannSolver(channel=ch, theta=0.7, time_step=300, spatial_step=500, simulation_time=3 * 3600)
    out, wall = run_and_capture(sol, 1e-6)
    out["us_target"] = sample_targets(hyd, sol.number_of_time_levels, sol.time_step)
    out["storage_stage"] = np.array(ss.stage_hydrograph, dtype=np.float64)[1:]
    save("irr_storage", out, base_meta(sol, 1e-6, wall, storage_curve=curve.tolist(), storage_alpha=1.0, storage_beta=0.0,
                                       storage_min_stage=1.0, ds_initial_depth=2.2, storage_rc_type='polynomial',
                                       storage_rc=dict(a=4.0, b=6.0, c=0.0, shift=0.0),
                                       storage_losses=dict(reservoir_length=300.0, K_q=0.2),
                                       storage_bounds=[float(ss.Y_min), float(ss.Y_max)]))


def case_bench_size():
    """The shapes bench.py launches, pinned to the reference itself: SURVEY 8(d) C3 draws at N = 4096 (two reaches, three
    levels: 0.64 s per Newton iteration in the reference) and C5 draws at N = 512 (two reaches, five levels)."""
    synthetic_rect("c3_4096", 2, 4096, 3, 20260213, slim=True)
    synthetic_trap("c5_512", 2, 512, 5, 20260214, slim=True)


def case_gerd_full():
    """cases/gerd_roseires over its whole simulation (384 levels, settings.py:3-8), solution + Newton counts only."""
    gd, cf, rr = _gerd_imports()
    from cases.gerd_roseires import settings as S
    sol, rel, rc, chs, secs = build_gerd(gd, cf, rr, None, S.sim_duration // 3600,
                                         "cases/gerd_roseires/data/inflow_hydrograph.csv", True)
    out, wall = run_and_capture(sol, S.tolerance, slim=True)
    out["us_target"] = np.array(rel.table[:sol.number_of_time_levels, 1], dtype=np.float64)
    save("gerd_full", out, base_meta(sol, S.tolerance, wall, rating=rating_spec(rc, rr),
                                     ds_initial_depth=float(sol.channel.downstream_boundary.initial_depth)))
    os.chdir(os.path.dirname(os.path.abspath(__file__)))


GATES_INFLOW_SCALE, GATES_STEPS = 8.0, 40


def case_gerd_gates(name="gerd_gates", steps=None, spatial_step=None):
    """A boundary plugin with NO device form: RoseiresRatingCurve(smooth=False) (roseires_rating_curve.py:65-78, :111-140) -
    the gates open / close on the stage of the previous evaluation with a cool-down in simulation time, so discharge() is
    stateful and time-dependent.  The flood wave is the case's inflow table scaled by 8 above its base flow so that the
    opening threshold (initial stage + 0.5 m) is crossed within 40 levels (gates open at level 27, close at level 36)."""
    gd, cf, rr = _gerd_imports()
    from cases.gerd_roseires import settings as S
    from src.hydromodel.channel import Channel
    from src.hydromodel.boundary import Boundary
    from src.hydromodel.hydrograph import Hydrograph
    from src.hydromodel.preissmann import PreissmannSolver
    tab = cf.import_hydrograph("cases/gerd_roseires/data/inflow_hydrograph.csv")
    tab[:, 1] = tab[0, 1] + GATES_INFLOW_SCALE * (tab[:, 1] - tab[0, 1])
    inflow = Hydrograph(table=tab)
    rel = gd.GerdHydrograph()
    steps = GATES_STEPS if steps is None else steps
    rel.build(inflow_hydrograph=inflow, time_step=S.time_step, duration=steps * 3600, initial_stage=S.initial_gerd_level)
    Q0 = rel.get_at(time=0)
    chs, secs = cf.load_trapzoid_xs(file_path=S.cross_sections_path, n_fp=None, n_main=None)
    bed = secs[-1].z_min
    us = Boundary(condition='flow_hydrograph', hydrograph=rel, chainage=chs[0])
    rc = rr.RoseiresRatingCurve(initial_stage=S.initial_roseires_level, initial_flow=Q0, smooth=False)
    ds = Boundary(initial_depth=S.initial_roseires_level - bed, bed_level=bed, condition='rating_curve', rating_curve=rc,
                  chainage=chs[-1])
    ch = Channel(initial_flow=Q0, upstream_boundary=us, downstream_boundary=ds)
    c = cf.import_table(S.coords_path, sort_by='chainage')
    ch.set_coords(coords=c[:, 1:], chainages=c[:, 0])
    ch.set_cross_sections(chainages=chs, sections=secs)
    sol = PreissmannSolver(channel=ch, theta=S.theta, time_step=S.time_step, spatial_step=S.spatial_step if spatial_step is None else spatial_step,
                           simulation_time=steps * 3600)
    out, wall = run_and_capture(sol, S.tolerance, slim=True)
    out["us_target"] = np.array(rel.table[:sol.number_of_time_levels, 1], dtype=np.float64)
    spec = rating_spec(rc, rr)
    spec.update(smooth=False, max_cooldown=float(rc.max_cooldown))
    save(name, out, base_meta(sol, S.tolerance, wall, rating=spec, inflow_scale=GATES_INFLOW_SCALE,
                              ds_initial_depth=float(ds.initial_depth), host_evaluated="downstream"))
    os.chdir(os.path.dirname(os.path.abspath(__file__)))


GATES_LONG_STEPS, GATES_LONG_DX = 30, 50.0


def case_gerd_gates_long():
    """The same plugin on a reach LONGER than a table kernel keeps on chip: cases/gerd_roseires at a spatial step of 50 m - 2 409
    nodes (2 048 fit one lane grid) - under the same flood wave, 30 levels: the gates open at level 27 as they do at 1 000 m.
    The reference has no node limit (solver.py:34-38, :53-55) and evaluates any RatingCurve subclass on any grid."""
    case_gerd_gates("gerd_gates_long", GATES_LONG_STEPS, GATES_LONG_DX)


def weir_outflow(stage):
    """reservoir outlet of case_storage_callable: nothing below the crest, a broad-crested weir above it - a Python
    callable handed to RatingCurve.function (rating_curve.py:50-52), which has no device form"""
    crest = 6.0
    return 0.0 if stage <= crest else 55.0 * (stage - crest) ** 1.5


def case_storage_callable():
    """LumpedStorage whose outflow rating curve is a Python callable (lumped_storage.py:24-35 calls
    rating_curve.discharge(Y, time) inside the brentq root function): the second plugin shape without a device form."""
    from src.hydromodel.channel import Channel
    from src.hydromodel.boundary import Boundary
    from src.hydromodel.hydrograph import Hydrograph
    from src.hydromodel.lumped_storage import LumpedStorage
    from src.hydromodel.preissmann import PreissmannSolver
    from src.hydromodel.rating_curve import RatingCurve
    L = 10000.0; Qb = 200.0
    hyd = Hydrograph(akbari_hydrograph(Qb, 900.0, 3 * 3600.0, 9 * 3600.0))
    us = Boundary(condition='flow_hydrograph', bed_level=2.0, chainage=0, hydrograph=hyd)
    ds = Boundary(condition='fixed_depth', initial_depth=5, bed_level=0, chainage=L)
    rc = RatingCurve(); rc.function = weir_outflow; rc.defined = True
    ss = LumpedStorage(surface_area=8.0e5, min_stage=4.0, solution_boundaries=(0, 30), rating_curve=rc)
    ds.set_lumped_storage(ss)
    ch = Channel(width=100, initial_flow=Qb, roughness=0.03, upstream_boundary=us, downstream_boundary=ds)
    sol = PreissmannSolver(channel=ch, theta=0.7, time_step=1200, spatial_step=500, simulation_time=12 * 3600)
    out, wall = run_and_capture(sol, 1e-6)
    out["us_target"] = sample_targets(hyd, sol.number_of_time_levels, sol.time_step)
    out["storage_stage"] = np.array(ss.stage_hydrograph, dtype=np.float64)[1:]
    save("storage_callable_rc", out, base_meta(sol, 1e-6, wall, storage_area=8.0e5, storage_min_stage=4.0,
                                               storage_bounds=[0, 30], ds_initial_depth=5.0, width=100, roughness=0.03,
                                               weir=dict(crest=6.0, coefficient=55.0, exponent=1.5),
                                               host_evaluated="downstream"))


def case_rmse_curve():
    """the calibration loop of cases/gerd_roseires/n_calibrate.py:5-17, :55-67 - ten Manning-n values,
    each a 32-level run of model.run on the small inflow table, GERD tail-water levels interpolated at six discharges,
    RMSE against the target levels.  n_calibrate.py runs at import time and writes a CSV into the working directory, so its
    two small functions (run_model, calc_rmse_curve) are called through model.run here with its module constants."""
    _gerd_imports()
    import cases.gerd_roseires.model as M
    H_target = np.array([497.5, 500, 502, 505, 507, 510])          # n_calibrate.py:26-28
    Q = np.array([1562.5, 3850, 6000, 10000, 14000, 21000])
    n_values = np.linspace(0.020, 0.060, 10)                      # n_calibrate.py:66
    Y, rmse = [], []
    for n in n_values:
        t0 = time.time()
        y = np.array(M.run(n_main=float(n), Q=Q, verbose=0, folder=None,
                           inflow_hyd_path="cases\\gerd_roseires\\data\\inflow_hydrograph_small.csv", coords_path=None,
                           inflow_hyd_func=None, sim_duration=None))
        Y.append(y)
        rmse.append(float(np.mean((y - H_target) ** 2) ** 0.5))    # n_calibrate.py:60
        print(f"   n_main={n:.5f}: RMSE {rmse[-1]:.6f}  ({time.time() - t0:.1f} s)")
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "rmse_curve.npz"),
                        meta=np.array(json.dumps(dict(generator="oracle/gen_golden.py", kind="rmse_curve",
                                                      reference="cve-mohd/flow-sim snapshot 2026-02-13, run in the build container"))),
                        n_values=n_values, rmse=np.array(rmse), levels=np.array(Y), H_target=H_target, Q=Q)
    os.chdir(os.path.dirname(os.path.abspath(__file__)))


def case_result_summaries():
    """Solver.save_results' text summary (solver.py:188-233) as the reference writes it, for cases/akbari_firoozi and
    cases/example.  openpyxl is absent here, so pandas.ExcelWriter / DataFrame.to_excel are no-ops for the duration of the
    call (harness side only): the workbook is not produced, the .txt next to it is the reference's own output."""
    import tempfile
    import pandas as pd
    from src.hydromodel.solver import Solver

    class NoWorkbook:
        def __init__(self, *a, **k): pass
        def __enter__(self): return self
        def __exit__(self, *exc): return False

    def summary(sol):
        Solver.prepare_results(sol)
        real_writer, real_to_excel = pd.ExcelWriter, pd.DataFrame.to_excel
        pd.ExcelWriter, pd.DataFrame.to_excel = NoWorkbook, lambda *a, **k: None
        try:
            with tempfile.TemporaryDirectory() as d:
                sol.save_results(folder_path=d, file_name="results.xlsx")
                return open(os.path.join(d, "results.txt")).read()
        finally:
            pd.ExcelWriter, pd.DataFrame.to_excel = real_writer, real_to_excel

    texts = {}
    for name in ("akbari", "example"):
        sol = SOLVERS[name]()
        sol.prepare_results = lambda: None
        sol.run(tolerance=1e-4, verbose=0)
        del sol.prepare_results
        texts[name] = summary(sol)
        print(f"--- {name}\n{texts[name]}")
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "result_summaries.npz"),
                        meta=np.array(json.dumps(dict(generator="oracle/gen_golden.py", kind="result_summaries",
                                                      reference="cve-mohd/flow-sim snapshot 2026-02-13, run in the build container"))),
                        **{k: np.array(v) for k, v in texts.items()})


def _akbari_solver():
    from src.hydromodel.channel import Channel
    from src.hydromodel.boundary import Boundary
    from src.hydromodel.preissmann import PreissmannSolver
    from src.hydromodel.hydrograph import Hydrograph
    from cases.akbari_firoozi import settings as S
    hyd = Hydrograph(S.hydrograph)
    us = Boundary(condition='flow_hydrograph', bed_level=S.S_0 * S.length, chainage=0, hydrograph=hyd)
    ds = Boundary(condition='normal_depth', bed_level=0, chainage=S.length)
    ch = Channel(width=S.width, initial_flow=S.initial_flow, roughness=S.roughness,
                 upstream_boundary=us, downstream_boundary=ds, interpolation_method='steady-state')
    return PreissmannSolver(channel=ch, theta=S.theta, time_step=S.preissmann_dt, spatial_step=S.spatial_step,
                            simulation_time=S.duration, regularization=False)


def example_inflow(t):
    """cases/example/main.py:8-29 restated as data (the module runs the whole case when imported)"""
    q0, qp = 1000.0, 10000.0
    rise, hold, fall = 3 * 3600, 6 * 3600, 4 * 3600
    if t <= 0:
        return q0
    if t < rise:
        return q0 + (qp - q0) * t / rise
    if t - rise < hold:
        return qp
    if t - rise - hold < fall:
        return qp - (qp - q0) * (t - rise - hold) / fall
    return q0


def _example_solver():
    from src.hydromodel.channel import Channel
    from src.hydromodel.boundary import Boundary
    from src.hydromodel.hydrograph import Hydrograph
    from src.hydromodel.lumped_storage import LumpedStorage
    from src.hydromodel.preissmann import PreissmannSolver
    hyd = Hydrograph(function=example_inflow)
    us = Boundary(condition='flow_hydrograph', bed_level=5, chainage=0, hydrograph=hyd)
    ds = Boundary(condition='fixed_depth', initial_depth=5, bed_level=0, chainage=20000)
    ds.set_lumped_storage(LumpedStorage(surface_area=5000 * 250, min_stage=5, solution_boundaries=(0, 200)))
    ch = Channel(width=250, initial_flow=hyd.get_at(0), roughness=0.027, upstream_boundary=us, downstream_boundary=ds)
    return PreissmannSolver(channel=ch, theta=0.8, time_step=3600, spatial_step=1000, simulation_time=24 * 3600)


SOLVERS = {"akbari": _akbari_solver, "example": _example_solver}


CASES = {
    "akbari": case_akbari,
    "example": case_example,
    "synthetic_rect_64": lambda: synthetic_rect("synthetic_rect_64", 4, 64, 5, 20260213),
    "synthetic_rect_512": lambda: synthetic_rect("synthetic_rect_512", 2, 512, 4, 20260213),
    "synthetic_trap_64": lambda: synthetic_trap("synthetic_trap_64", 4, 64, 5, 20260214),
    "bc_matrix": case_bc_matrix,
    "irregular": case_irregular,
    "storage_general": case_storage_general,
    "upstream_kinds": case_upstream_kinds,
    "irr_storage": case_irr_storage,
    "bench_size": case_bench_size,
    "storage_callable": case_storage_callable,
    "result_summaries": case_result_summaries,
}
# long-running cases (minutes each): run with --only NAME
SLOW_CASES = {"gerd_full": case_gerd_full, "gerd_gates": case_gerd_gates, "rmse_curve": case_rmse_curve, "gerd_gates_long": case_gerd_gates_long}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--gerd-steps", type=int, default=48)
    a = ap.parse_args()
    names = list(CASES) + ["gerd"] + list(SLOW_CASES)
    for name in names:
        if a.only and a.only != name:
            continue
        print(f"[{name}]")
        if name == "gerd":
            case_gerd(a.gerd_steps)
        elif name in SLOW_CASES:
            SLOW_CASES[name]()
        else:
            CASES[name]()
