"""Seeded synthetic reaches (SURVEY.md section 8d, config C3 generator) as oracle Problems."""
from math import cos, pi, sin

import numpy as np

from oracle import preissmann_oracle as O


def normal_depth_rect(b, n, S0, Q):
    lo, hi = 1e-9, 200.0
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        A = b * mid
        P = b + 2 * mid
        if A * (A / P) ** (2.0 / 3.0) / n * S0 ** 0.5 < Q:
            lo = mid
        else:
            hi = mid
    return 0.5 * (lo + hi)


def c3_params(rng):
    b = rng.uniform(50, 300); n = rng.uniform(0.02, 0.04); S0 = rng.uniform(2e-4, 1e-3)
    Qb = rng.uniform(50, 500) * (b / 100)
    return b, n, S0, Qb


def akbari_shape(Qb, Qp, tp, tb, t):
    if t <= tp:
        return Qp / 2 * sin(pi * t / tp - pi / 2) + Qp / 2 + Qb
    if t <= tb:
        return Qp / 2 * cos(pi * (t - tp) / (tb - tp)) + Qp / 2 + Qb
    return Qb


def rect_problem(N, seed, n_steps, dt=600.0, dx=250.0, theta=0.6, tol=1e-6, steady=False):
    rng = np.random.default_rng(20260213 + seed)
    b, n, S0, Qb = c3_params(rng)
    L = (N - 1) * dx
    geo = {k: np.zeros(N) for k in O.GEO_KEYS}
    geo["b_main"][:] = b; geo["n_main"][:] = n; geo["n_left"][:] = n; geo["n_right"][:] = n
    w2 = np.arange(N) / (N - 1)
    geo["z_bed"] = S0 * L * (1 - w2)
    hn = normal_depth_rect(b, n, S0, Qb)
    nt = n_steps + 1
    if steady:
        tgt = np.full(nt, Qb)
    else:
        tgt = np.array([akbari_shape(Qb, 2 * Qb, 5 * 3600.0, 15 * 3600.0, k * dt) for k in range(nt)])
    us = O.BC("flow_hydrograph", bed_level=S0 * L, target=tgt)
    ds = O.BC("normal_depth", bed_level=0.0, bed_slope=S0)
    return O.Problem(geo=geo, h0=np.full(N, hn), Q0=np.full(N, Qb), us=us, ds=ds, theta=theta, dt=dt, dx=dx,
                     nt=nt, tol=tol)


def normal_depth_trap(b, m, n, S0, Q):
    lo, hi = 1e-9, 200.0
    s = (1.0 + m * m) ** 0.5
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        A = (b + m * mid) * mid
        P = b + 2 * mid * s
        if A * (A / P) ** (2.0 / 3.0) / n * S0 ** 0.5 < Q:
            lo = mid
        else:
            hi = mid
    return 0.5 * (lo + hi)


def trap_problem(b, m, n, S0, Qb, N, n_steps, dt=1800.0, dx=500.0, theta=0.6, tol=1e-6):
    """SURVEY 8d C5 channel (simple trapezoid, power rating curve through the normal depth, akbari-shaped
    inflow) for given parameters, started from uniform flow."""
    L = (N - 1) * dx
    geo = {k: np.zeros(N) for k in O.GEO_KEYS}
    geo["b_main"][:] = b; geo["m_main"][:] = m
    geo["n_main"][:] = n; geo["n_left"][:] = n; geo["n_right"][:] = n
    geo["z_bed"] = S0 * L * (1 - np.arange(N) / (N - 1))
    hn = normal_depth_trap(b, m, n, S0, Qb)
    nt = n_steps + 1
    tgt = np.array([akbari_shape(Qb, 2 * Qb, 5 * 3600.0, 15 * 3600.0, k * dt) for k in range(nt)])
    us = O.BC("flow_hydrograph", bed_level=S0 * L, target=tgt)
    ds = O.BC("rating_curve", bed_level=0.0, initial_depth=hn, rc_type="power", rc=dict(a=Qb / hn ** 1.6, b=1.6))
    return O.Problem(geo=geo, h0=np.full(N, hn), Q0=np.full(N, Qb), us=us, ds=ds, theta=theta, dt=dt, dx=dx,
                     nt=nt, tol=tol)
