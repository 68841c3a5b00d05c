"""The golden cases rebuilt through the host mirror (flowsim_amd.hydromodel) with the same calls
oracle/gen_golden.py made against the reference: used to check the mirror's geometry / initial
conditions on the CPU and the whole drop-in path (Channel -> PreissmannSolver.run) on the GPU."""
from math import cos, pi, sin

import numpy as np

from flowsim_amd.hydromodel import (Boundary, Channel, Hydrograph, IrregularSection, LumpedStorage, PreissmannSolver,
                                    RatingCurve, TrapezoidalSection)


def akbari_shape(Q_b, Q_p, t_p, t_b):
    def f(t):
        if t <= t_p:
            return Q_p / 2 * sin(pi * t / t_p - pi / 2) + Q_p / 2 + Q_b
        elif t <= t_b:
            return Q_p / 2 * cos(pi * (t - t_p) / (t_b - t_p)) + Q_p / 2 + Q_b
        return Q_b
    return f


def akbari():
    from cases.akbari_firoozi.main_preissmann import build
    from cases.akbari_firoozi import settings as S
    return build(), S.tolerance


def example():
    from cases.example.main import build
    return build(), 1e-4


def bc_stage_fixed():
    L = 12000.0
    tab = np.array([[0, 8.0 + 3.0], [3600 * 2, 8.0 + 3.6], [3600 * 5, 8.0 + 3.1], [3600 * 12, 8.0 + 3.0]])
    us = Boundary(condition='stage_hydrograph', bed_level=8.0, chainage=0, initial_depth=3.0, hydrograph=Hydrograph(table=tab))
    ds = Boundary(condition='fixed_depth', bed_level=0.0, chainage=L, initial_depth=4.0)
    ch = Channel(width=80, initial_flow=300.0, roughness=0.03, upstream_boundary=us, downstream_boundary=ds,
                 interpolation_method='linear')
    return PreissmannSolver(channel=ch, theta=0.7, time_step=900, spatial_step=500, simulation_time=8 * 3600), 1e-6


def bc_trap_poly():
    L = 15000.0; S0 = 4e-4
    xs_u = TrapezoidalSection(z_bed=100.0 + S0 * L, b_main=40.0, m_main=1.5, n_main=0.032, bed_slope=S0)
    xs_m = TrapezoidalSection(z_bed=100.0 + S0 * L * 0.4, b_main=55.0, m_main=2.0, n_main=0.028, bed_slope=S0)
    xs_d = TrapezoidalSection(z_bed=100.0, b_main=60.0, m_main=2.5, n_main=0.03, bed_slope=S0)
    rc = RatingCurve(); rc.set(type='polynomial', a=9.0, b=35.0, c=-20.0)
    rc.stage_shift = -100.0
    h_ds = 3.0
    Q0 = rc.discharge(100.0 + h_ds)
    hyd = Hydrograph(akbari_shape(Q0, 1.5 * Q0, 3 * 3600.0, 9 * 3600.0))
    us = Boundary(condition='flow_hydrograph', bed_level=xs_u.z_bed, chainage=0, hydrograph=hyd)
    ds = Boundary(condition='rating_curve', bed_level=100.0, chainage=L, initial_depth=h_ds, rating_curve=rc)
    ch = Channel(initial_flow=Q0, upstream_boundary=us, downstream_boundary=ds)
    ch.set_cross_sections([0.0, 0.6 * L, L], [xs_u, xs_m, xs_d])
    return PreissmannSolver(channel=ch, theta=0.6, time_step=1200, spatial_step=600, simulation_time=10 * 3600), 1e-6


def bc_compound_normal():
    L = 20000.0; S0 = 3e-4

    def comp(z, b, m, hb, bl, br, mf, slope):
        return TrapezoidalSection(z_bed=z, b_main=b, m_main=m, n_main=0.03, z_bank=z + hb, b_fp_left=bl,
                                  b_fp_right=br, m_fp=mf, n_left=0.06, n_right=0.05, bed_slope=slope)
    xs_u = comp(S0 * L, 30.0, 2.0, 2.5, 60.0, 40.0, 4.0, S0)
    xs_d = comp(0.0, 36.0, 1.5, 2.2, 80.0, 50.0, 3.0, S0)
    hyd = Hydrograph(akbari_shape(120.0, 500.0, 2 * 3600.0, 7 * 3600.0))
    us = Boundary(condition='flow_hydrograph', bed_level=S0 * L, chainage=0, hydrograph=hyd)
    ds = Boundary(condition='normal_depth', bed_level=0.0, chainage=L)
    ch = Channel(initial_flow=120.0, upstream_boundary=us, downstream_boundary=ds, interpolation_method='steady-state')
    ch.set_cross_sections([0.0, L], [xs_u, xs_d])
    return PreissmannSolver(channel=ch, theta=0.65, time_step=600, spatial_step=500, simulation_time=6 * 3600), 1e-6


def synthetic_trap(member, N=64, n_steps=5, seed=20260214, theta=0.6, dt=1800, dx=500.0, tol=1e-6):
    """SURVEY 8d C5 generator in fp64 (simple trapezoid + power rating curve), member-th draw."""
    rng = np.random.default_rng(seed)
    for _ in range(member + 1):
        b = rng.uniform(20, 100); m = rng.uniform(1, 3); n = rng.uniform(0.025, 0.04)
        S0 = rng.uniform(2e-4, 1e-3); Qb = rng.uniform(50, 500) * (b / 100)
    L = (N - 1) * dx
    xs_u = TrapezoidalSection(z_bed=S0 * L, b_main=b, m_main=m, n_main=n, bed_slope=S0)
    xs_d = TrapezoidalSection(z_bed=0.0, b_main=b, m_main=m, n_main=n, bed_slope=S0)
    h_n = xs_d.normal_depth(Q_target=Qb)
    rc = RatingCurve(); rc.set(type='power', a=Qb / h_n ** 1.6, b=1.6)
    hyd = Hydrograph(akbari_shape(Qb, 2 * Qb, 5 * 3600.0, 15 * 3600.0))
    us = Boundary(condition='flow_hydrograph', bed_level=S0 * L, chainage=0, hydrograph=hyd)
    ds = Boundary(condition='rating_curve', bed_level=0.0, chainage=L, initial_depth=h_n, rating_curve=rc)
    ch = Channel(initial_flow=Qb, upstream_boundary=us, downstream_boundary=ds, interpolation_method='steady-state')
    ch.set_cross_sections([0.0, L], [xs_u, xs_d])
    return PreissmannSolver(channel=ch, theta=theta, time_step=dt, spatial_step=dx, simulation_time=n_steps * dt), tol


def gerd():
    """cases/gerd_roseires, first 48 h of the regulated scenario with the tabulated inflow (tests/golden/gerd.npz)."""
    from cases.gerd_roseires.model import build
    from cases.gerd_roseires import settings as S
    solver, _ = build(inflow_hyd_func=None, sim_duration=48 * 3600)
    return solver, S.tolerance


def gerd_member(n_main):
    """one member of the Manning-n study (n_calibrate set-up: short inflow table, no curvature)."""
    from cases.gerd_roseires.n_calibrate import member_setup
    from cases.gerd_roseires import settings as S
    solver, _ = member_setup(n_main)
    return solver, S.tolerance


def irr_single():
    """two polylines with different stations, one thalweg (tests/golden/irr_single.npz)"""
    L = 6000.0; S0 = 2e-4
    xu = np.array([0, 10, 14, 30, 34, 60, 66, 80.0]); zu = np.array([8, 3.0, 0.4, 0.0, 0.6, 2.5, 2.8, 8.0])
    xd = np.array([0, 12, 18, 33, 41, 58, 70, 90.0]); zd = np.array([7.5, 2.6, 0.3, 0.0, 0.5, 2.0, 2.6, 7.5])
    xs_u = IrregularSection(x=xu, z=S0 * L + zu, n=0.03, bed_slope=S0)
    xs_d = IrregularSection(x=xd, z=zd, n=0.034, bed_slope=S0)
    hyd = Hydrograph(akbari_shape(40.0, 260.0, 1 * 3600.0, 3 * 3600.0))
    us = Boundary(condition='flow_hydrograph', bed_level=S0 * L, chainage=0, hydrograph=hyd)
    ds = Boundary(condition='normal_depth', bed_level=0.0, chainage=L)
    ch = Channel(initial_flow=40.0, upstream_boundary=us, downstream_boundary=ds, interpolation_method='steady-state')
    ch.set_cross_sections([0.0, L], [xs_u, xs_d])
    return PreissmannSolver(channel=ch, theta=0.65, time_step=300, spatial_step=500, simulation_time=3 * 3600), 1e-6


def irr_levee():
    """secondary channel behind a levee, composite roughness, power rating curve (tests/golden/irr_levee.npz)"""
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
    hyd = Hydrograph(akbari_shape(Q0, 6.0 * Q0, 1.5 * 3600.0, 4 * 3600.0))
    us = Boundary(condition='flow_hydrograph', bed_level=S0 * L, chainage=0, hydrograph=hyd)
    ds = Boundary(condition='rating_curve', bed_level=0.0, chainage=L, initial_depth=h_ds, rating_curve=rc)
    ch = Channel(initial_flow=Q0, upstream_boundary=us, downstream_boundary=ds, interpolation_method='linear')
    us.initial_depth = 1.7
    ch.set_cross_sections([0.0, L], [xs_u, xs_d])
    return PreissmannSolver(channel=ch, theta=0.7, time_step=300, spatial_step=500, simulation_time=4 * 3600), 1e-6


def irr_mixed():
    """trapezoid -> polyline -> polyline along a bending centre line (tests/golden/irr_mixed.npz)"""
    L = 8000.0; S0 = 2.5e-4
    xs_u = TrapezoidalSection(z_bed=S0 * L, b_main=30.0, m_main=2.0, n_main=0.03, bed_slope=S0)
    xm = np.array([-40, -25, -16, -6, 5, 17, 26, 41.0]); zm = np.array([6, 2.2, 0.5, 0.0, 0.1, 0.6, 2.4, 6.0])
    xs_m = IrregularSection(x=xm, z=S0 * L * 0.5 + zm, n=0.031, bed_slope=S0)
    xd = np.array([-45, -22, -15, -4, 8, 19, 30, 46.0]); zd = np.array([6, 2.0, 0.4, 0.0, 0.2, 0.7, 2.1, 6.0])
    xs_d = IrregularSection(x=xd, z=zd, n=0.033, bed_slope=S0)
    hyd = Hydrograph(akbari_shape(60.0, 200.0, 1 * 3600.0, 3 * 3600.0))
    us = Boundary(condition='flow_hydrograph', bed_level=S0 * L, chainage=0, hydrograph=hyd)
    ds = Boundary(condition='normal_depth', bed_level=0.0, chainage=L)
    ch = Channel(initial_flow=60.0, upstream_boundary=us, downstream_boundary=ds, interpolation_method='steady-state')
    ch.set_cross_sections([0.0, 0.5 * L, L], [xs_u, xs_m, xs_d])
    ch.set_coords([[0.0, 0.0], [2500.0, 300.0], [4000.0, 1500.0], [4200.0, 3800.0], [6500.0, 5200.0]],
                  [0.0, 2500.0, 4400.0, 6700.0, L])
    return PreissmannSolver(channel=ch, theta=0.65, time_step=300, spatial_step=500, simulation_time=3 * 3600), 1e-6


def _storage_general(rc_type, losses, trap):
    L = 10000.0
    hyd = Hydrograph(akbari_shape(200.0, 900.0, 3 * 3600.0, 9 * 3600.0))
    us = Boundary(condition='flow_hydrograph', bed_level=2.0, chainage=0, hydrograph=hyd)
    ds = Boundary(condition='fixed_depth', initial_depth=5, bed_level=0, chainage=L)
    stages = np.arange(0.0, 30.01, 0.5)
    curve = np.column_stack([stages, 2.0e5 + 4.0e4 * stages + 900.0 * stages ** 2])
    rc = None
    if rc_type == 'polynomial':
        rc = RatingCurve(); rc.set(type='polynomial', a=6.0, b=10.0, c=0.0)
    elif rc_type == 'power':
        rc = RatingCurve(); rc.set(type='power', a=17.0, b=1.5)
    ss = LumpedStorage(surface_area=None, min_stage=4.0, solution_boundaries=(0, 30), rating_curve=rc)
    ss.set_area_curve(curve, alpha=1.1, beta=0.2)
    if losses:
        ss.capture_losses = True
        ss.reservoir_length = 800.0
        ss.K_q = 0.3
    ds.set_lumped_storage(ss)
    if trap:
        xs_u = TrapezoidalSection(z_bed=2.0, b_main=90.0, m_main=2.0, n_main=0.028, bed_slope=2e-4)
        xs_d = TrapezoidalSection(z_bed=0.0, b_main=110.0, m_main=2.5, n_main=0.03, bed_slope=2e-4)
        ch = Channel(initial_flow=200.0, upstream_boundary=us, downstream_boundary=ds)
        ch.set_cross_sections([0.0, L], [xs_u, xs_d])
    else:
        ch = Channel(width=100, initial_flow=200.0, roughness=0.03, upstream_boundary=us, downstream_boundary=ds)
    return PreissmannSolver(channel=ch, theta=0.7, time_step=1200, spatial_step=500, simulation_time=12 * 3600), 1e-6


def bc_us_fixed_ds_flow():
    """reservoir level upstream, closing gate (dipping outflow hydrograph) downstream"""
    L = 10000.0; S0 = 2e-4; Q0 = 150.0
    xs_u = TrapezoidalSection(z_bed=S0 * L, b_main=40.0, m_main=1.5, n_main=0.03, bed_slope=S0)
    xs_d = TrapezoidalSection(z_bed=0.0, b_main=40.0, m_main=1.5, n_main=0.03, bed_slope=S0)
    hn = xs_d.normal_depth(Q_target=Q0)
    hyd = Hydrograph(akbari_shape(Q0, -60.0, 2 * 3600.0, 6 * 3600.0))
    us = Boundary(condition='fixed_depth', bed_level=S0 * L, chainage=0, initial_depth=hn)
    ds = Boundary(condition='flow_hydrograph', bed_level=0.0, chainage=L, hydrograph=hyd, initial_depth=hn)
    ch = Channel(initial_flow=Q0, upstream_boundary=us, downstream_boundary=ds, interpolation_method='steady-state')
    ch.set_cross_sections([0.0, L], [xs_u, xs_d])
    return PreissmannSolver(channel=ch, theta=0.65, time_step=600, spatial_step=500, simulation_time=8 * 3600), 1e-6


def bc_us_rating_ds_stage():
    """head-dependent inflow upstream (rating curve with negative slope), stage hydrograph downstream"""
    L = 8000.0; S0 = 3e-4; Q0 = 200.0; h_us = 3.0
    rc = RatingCurve(); rc.set(type='polynomial', a=0.0, b=-60.0, c=Q0 + 60.0 * (S0 * L + h_us))
    tab = np.array([[0, 3.0], [3600 * 2, 3.8], [3600 * 5, 3.2], [3600 * 12, 3.0]])
    us = Boundary(condition='rating_curve', bed_level=S0 * L, chainage=0, initial_depth=h_us, rating_curve=rc)
    ds = Boundary(condition='stage_hydrograph', bed_level=0.0, chainage=L, initial_depth=3.0, hydrograph=Hydrograph(table=tab))
    ch = Channel(width=60, initial_flow=Q0, roughness=0.03, upstream_boundary=us, downstream_boundary=ds,
                 interpolation_method='linear')
    return PreissmannSolver(channel=ch, theta=0.7, time_step=900, spatial_step=500, simulation_time=8 * 3600), 1e-6


def bc_us_normal_ds_stage():
    """normal depth imposed upstream, stage hydrograph downstream"""
    L = 12000.0; S0 = 4e-4; Q0 = 180.0
    xs_u = TrapezoidalSection(z_bed=S0 * L, b_main=50.0, m_main=2.0, n_main=0.03, bed_slope=S0)
    xs_d = TrapezoidalSection(z_bed=0.0, b_main=50.0, m_main=2.0, n_main=0.03, bed_slope=S0)
    hn = xs_d.normal_depth(Q_target=Q0)
    tab = np.array([[0, hn], [3600 * 2, hn + 0.6], [3600 * 5, hn + 0.1], [3600 * 12, hn]])
    us = Boundary(condition='normal_depth', bed_level=S0 * L, chainage=0, initial_depth=hn)
    ds = Boundary(condition='stage_hydrograph', bed_level=0.0, chainage=L, initial_depth=hn, hydrograph=Hydrograph(table=tab))
    ch = Channel(initial_flow=Q0, upstream_boundary=us, downstream_boundary=ds, interpolation_method='steady-state')
    ch.set_cross_sections([0.0, L], [xs_u, xs_d])
    return PreissmannSolver(channel=ch, theta=0.7, time_step=600, spatial_step=500, simulation_time=6 * 3600), 1e-6


def irr_storage():
    """polyline sections in front of a general LumpedStorage (tests/golden/irr_storage.npz)"""
    L = 6000.0; S0 = 2e-4
    xu = np.array([0, 10, 14, 30, 34, 60, 66, 80.0]); zu = np.array([8, 3.0, 0.4, 0.0, 0.6, 2.5, 2.8, 8.0])
    xd = np.array([0, 12, 18, 33, 41, 58, 70, 90.0]); zd = np.array([7.5, 2.6, 0.3, 0.0, 0.5, 2.0, 2.6, 7.5])
    xs_u = IrregularSection(x=xu, z=S0 * L + zu, n=0.03, bed_slope=S0)
    xs_d = IrregularSection(x=xd, z=zd, n=0.034, bed_slope=S0)
    hyd = Hydrograph(akbari_shape(40.0, 260.0, 1 * 3600.0, 3 * 3600.0))
    us = Boundary(condition='flow_hydrograph', bed_level=S0 * L, chainage=0, hydrograph=hyd)
    ds = Boundary(condition='fixed_depth', initial_depth=2.2, bed_level=0.0, chainage=L)
    stages = np.arange(0.0, 12.01, 0.25)
    curve = np.column_stack([stages, 6.0e4 + 1.5e4 * stages + 500.0 * stages ** 2])
    rc = RatingCurve(); rc.set(type='polynomial', a=4.0, b=6.0, c=0.0)
    ss = LumpedStorage(surface_area=None, min_stage=1.0, solution_boundaries=(0, 12), rating_curve=rc)
    ss.set_area_curve(curve, alpha=1.0, beta=0.0)
    ss.capture_losses = True; ss.reservoir_length = 300.0; ss.K_q = 0.2
    ds.set_lumped_storage(ss)
    ch = Channel(initial_flow=40.0, upstream_boundary=us, downstream_boundary=ds)
    ch.set_cross_sections([0.0, L], [xs_u, xs_d])
    return PreissmannSolver(channel=ch, theta=0.7, time_step=300, spatial_step=500, simulation_time=3 * 3600), 1e-6


# ---- boundary plugins without a device form: evaluated on the host every Newton iteration (FS_BC_HOST_ROW) ----
def weir_outflow(stage):
    """reservoir outlet: nothing below the crest, a broad-crested weir above it (a Python callable as rating curve)"""
    crest = 6.0
    return 0.0 if stage <= crest else 55.0 * (stage - crest) ** 1.5


def storage_callable_rc():
    """LumpedStorage whose outflow curve is a Python callable (tests/golden/storage_callable_rc.npz)"""
    L = 10000.0
    hyd = Hydrograph(akbari_shape(200.0, 900.0, 3 * 3600.0, 9 * 3600.0))
    us = Boundary(condition='flow_hydrograph', bed_level=2.0, chainage=0, hydrograph=hyd)
    ds = Boundary(condition='fixed_depth', initial_depth=5, bed_level=0, chainage=L)
    rc = RatingCurve(); rc.function = weir_outflow; rc.defined = True
    ds.set_lumped_storage(LumpedStorage(surface_area=8.0e5, min_stage=4.0, solution_boundaries=(0, 30), rating_curve=rc))
    ch = Channel(width=100, initial_flow=200.0, roughness=0.03, upstream_boundary=us, downstream_boundary=ds)
    return PreissmannSolver(channel=ch, theta=0.7, time_step=1200, spatial_step=500, simulation_time=12 * 3600), 1e-6


def gerd_gates():
    """cases/gerd_roseires with operated gates (RoseiresRatingCurve(smooth=False): opens at level 27, closes at 36) under
    a flood wave scaled by 8 (tests/golden/gerd_gates.npz)"""
    from cases.gerd_roseires.model import build
    from cases.gerd_roseires import settings as S
    solver, _ = build(inflow_hyd_func=None, sim_duration=40 * 3600, inflow_scale=8.0, smooth_gates=False)
    return solver, S.tolerance


def gerd_gates_long():
    """the same gates on a reach longer than a table kernel keeps on chip: dx = 50 m, 2 409 nodes, 30 levels (tests/golden/gerd_gates_long.npz)"""
    from cases.gerd_roseires.model import build
    from cases.gerd_roseires import settings as S
    solver, _ = build(inflow_hyd_func=None, sim_duration=30 * 3600, inflow_scale=8.0, smooth_gates=False, spatial_step=50.0)
    return solver, S.tolerance


HOST_ROW_BUILDERS = {"storage_callable_rc": storage_callable_rc, "gerd_gates": gerd_gates, "gerd_gates_long": gerd_gates_long}


def storage_curve_poly_losses():
    return _storage_general('polynomial', True, False)


def storage_curve_power_trap():
    return _storage_general('power', True, True)


def storage_curve_closed():
    return _storage_general(None, False, False)


BUILDERS = {"storage_curve_poly_losses": storage_curve_poly_losses, "storage_curve_power_trap": storage_curve_power_trap,
            "storage_curve_closed": storage_curve_closed, "irr_single": irr_single, "irr_levee": irr_levee, "irr_mixed": irr_mixed, "gerd": gerd, "akbari": akbari, "example": example, "bc_stage_fixed": bc_stage_fixed, "bc_trap_poly": bc_trap_poly,
            "bc_compound_normal": bc_compound_normal, "bc_us_fixed_ds_flow": bc_us_fixed_ds_flow,
            "bc_us_rating_ds_stage": bc_us_rating_ds_stage, "bc_us_normal_ds_stage": bc_us_normal_ds_stage,
            "irr_storage": irr_storage}
