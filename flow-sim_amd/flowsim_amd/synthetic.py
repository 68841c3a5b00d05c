"""Synthetic batched channels of SURVEY.md section 8(d) (configuration C3): seeded per-reach
parameter draws, steady-state initial depth, akbari-shaped inflow table.  Host-side set-up only."""
import numpy as np

C3_SEED = 20260213


def c3_reach_parameters(first, count, seed=C3_SEED):
    """b ~ U(50,300) m, n ~ U(0.02,0.04), S0 ~ U(2e-4,1e-3), Q_base ~ U(50,500)*(b/100) for global
    reach indices [first, first+count): 4 draws per reach, in reach order, so any shard of the
    global batch sees exactly the values a single process would."""
    rng = np.random.default_rng(seed)
    u = rng.random((first + count, 4))[first:]
    b = 50.0 + 250.0 * u[:, 0]
    n = 0.02 + 0.02 * u[:, 1]
    S0 = 2e-4 + 8e-4 * u[:, 2]
    Qb = (50.0 + 450.0 * u[:, 3]) * (b / 100.0)
    return b, n, S0, Qb


def normal_depth_rect(b, n, S0, Q):
    """Normal depth of a rectangular channel by vectorised bisection (the 'steady-state' initial
    condition of channel.py:296-305 for a whole batch at once)."""
    lo = np.full_like(b, 1e-9)
    hi = np.full_like(b, 200.0)
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        A = b * mid
        P = b + 2 * mid
        below = A * (A / P) ** (2.0 / 3.0) / n * np.sqrt(S0) < Q
        lo = np.where(below, mid, lo)
        hi = np.where(below, hi, mid)
    return 0.5 * (lo + hi)


def inflow_table(Qb, levels, dt, t_peak=5 * 3600.0, t_base=15 * 3600.0):
    """Sinusoidal flood wave Q_base -> 3 Q_base -> Q_base sampled at k*dt: [levels, B]
    (the hydrograph form of cases/akbari_firoozi with Q_p = 2 Q_base)."""
    t = np.arange(levels)[:, None] * dt
    Qp = 2.0 * Qb[None, :]
    rise = Qp / 2 * np.sin(np.pi * t / t_peak - np.pi / 2) + Qp / 2 + Qb[None, :]
    fall = Qp / 2 * np.cos(np.pi * (t - t_peak) / (t_base - t_peak)) + Qp / 2 + Qb[None, :]
    return np.where(t <= t_peak, rise, np.where(t <= t_base, fall, Qb[None, :]))


C5_SEED = 20260214


def c5_reach_parameters(first, count, seed=C5_SEED):
    """simple trapezoids b ~ U(20,100), m ~ U(1,3), n ~ U(0.025,0.04), S0 ~ U(2e-4,1e-3),
    Q_base ~ U(50,500)*(b/100); 5 draws per reach in reach order."""
    rng = np.random.default_rng(seed)
    u = rng.random((first + count, 5))[first:]
    b = 20.0 + 80.0 * u[:, 0]
    m = 1.0 + 2.0 * u[:, 1]
    n = 0.025 + 0.015 * u[:, 2]
    S0 = 2e-4 + 8e-4 * u[:, 3]
    Qb = (50.0 + 450.0 * u[:, 4]) * (b / 100.0)
    return b, m, n, S0, Qb


def normal_depth_trap(b, m, n, S0, Q):
    """Normal depth of simple trapezoids by vectorised bisection."""
    lo = np.full_like(b, 1e-9)
    hi = np.full_like(b, 200.0)
    s = np.sqrt(1.0 + m * m)
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        A = (b + m * mid) * mid
        P = b + 2 * mid * s
        below = A * (A / P) ** (2.0 / 3.0) / n * np.sqrt(S0) < Q
        lo = np.where(below, mid, lo)
        hi = np.where(below, hi, mid)
    return 0.5 * (lo + hi)
