#!/usr/bin/env python3
"""Random-sweep fixture: runs the *reference* (cve-mohd/flow-sim, /root/reference, read-only) in THIS container on
seeded random channels built through its public API and writes tests/golden/random_sweep.npz.

TEST INFRASTRUCTURE ONLY (like gen_golden.py, whose capture helpers it uses).  The bundled cases and the benchmark
generators pin the path at a handful of parameter points; this pins it across the parameter space: 2 or 3 input sections
(rectangles, trapezoids, compound trapezoids, all three initial-condition methods, so the reference's own interpolation
and GVF code produce the node geometry and the initial state), theta, time and space steps over two decades, five
downstream boundary kinds incl. a storage, flow or stage hydrograph upstream.

    python oracle/gen_random_sweep.py [--cases 48] [--polyline 12] [--storage 12] [--bends 8]

Arrays of case i are stored as c{i:02d}_<name>, the per-case metadata as the list meta["cases"].
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (the repository root stays OFF sys.path: its src/ is a regular package and would shadow the reference's namespace package)


def draw_recipe(rng):
    """one random channel as plain data (numbers and strings only: it is stored in the fixture's metadata, and
    tests/test_random_sweep.py rebuilds the same channel through the mirror package from it)"""
    N = int(rng.choice([2, 3, 6, 17, 33, 64, 65, 100, 129, 200, 257]))
    dx = float(int(np.exp(rng.uniform(np.log(50.0), np.log(1500.0)))))
    dt = int(np.exp(rng.uniform(np.log(60.0), np.log(3600.0))))
    n_steps = int(rng.integers(2, 6))
    theta = float(rng.uniform(0.55, 1.0))
    L = (N - 1) * dx
    S0 = float(np.exp(rng.uniform(np.log(1e-4), np.log(3e-3))))
    family = ("rect", "trap", "compound")[rng.integers(0, 3)]
    b0 = float(np.exp(rng.uniform(np.log(6.0), np.log(400.0))))
    n_main = float(rng.uniform(0.018, 0.05))
    q = float(np.exp(rng.uniform(np.log(0.2), np.log(6.0))))
    Qb = q * b0
    n_sections = 2 if N < 6 else int(rng.integers(2, 4))
    chain = [0.0, L] if n_sections == 2 else [0.0, float(rng.uniform(0.3, 0.7)) * L, L]
    sections = []
    for c in chain:
        z = S0 * (L - c)
        b = b0 * float(rng.uniform(0.85, 1.15)) if n_sections == 3 else b0
        kw = dict(z_bed=z, b_main=b, m_main=0.0, n_main=n_main, bed_slope=S0)
        if family != "rect":
            kw["m_main"] = float(rng.uniform(0.5, 2.5))
        if family == "compound":
            kw.update(z_bank=z + float(rng.uniform(0.8, 3.0)) * (q ** 0.6), b_fp_left=b * float(rng.uniform(0.5, 3)),
                      b_fp_right=b * float(rng.uniform(0.5, 3)), m_fp=float(rng.uniform(2, 6)),
                      n_left=n_main * float(rng.uniform(1.2, 2)), n_right=n_main * float(rng.uniform(1.2, 2)))
        sections.append(kw)
    amp = float(np.exp(rng.uniform(np.log(0.2), np.log(2.0))))
    us_kind = ("flow_hydrograph", "flow_hydrograph", "stage_hydrograph")[rng.integers(0, 3)]
    ic = ("steady-state", "GVF_equation", "linear")[rng.integers(0, 3)]
    if us_kind == "stage_hydrograph" and ic == "GVF_equation":
        ic = "linear"
    wave = dict(rise=float(rng.uniform(2, 5)) * dt, fall=float(rng.uniform(6, 12)) * dt)
    ds_kind = ("normal_depth", "power", "polynomial", "fixed_depth", "storage")[rng.integers(0, 5)]
    if ds_kind == "storage" and N > 65:
        ds_kind = "normal_depth"
    return dict(N=N, dx=dx, dt=dt, n_steps=n_steps, theta=theta, L=L, S0=S0, family=family, b0=b0, Qb=Qb, chain=chain,
                sections=sections, amp=amp, us_kind=us_kind, ds_kind=ds_kind, ic=ic, wave=wave,
                rc_exponent=float(rng.uniform(1.3, 2.0)))


def draw_polyline_recipe(rng):
    """a channel of polyline sections (IrregularSection, cross_section.py:207-543): a valley of 7 ... 12 stations, in half
    of the draws with a levee that splits the section at low stages, composite roughness over three strips; the two input
    sections differ (union of stations at the interpolated nodes), one of them may be a trapezoid (mixed interpolation)"""
    N = int(rng.choice([3, 9, 17, 33, 65]))
    dx = float(int(np.exp(rng.uniform(np.log(100.0), np.log(800.0)))))
    dt = int(np.exp(rng.uniform(np.log(120.0), np.log(1800.0))))
    n_steps = int(rng.integers(2, 5))
    theta = float(rng.uniform(0.55, 1.0))
    L = (N - 1) * dx
    S0 = float(np.exp(rng.uniform(np.log(1.5e-4), np.log(1.5e-3))))
    W = float(np.exp(rng.uniform(np.log(30.0), np.log(200.0))))
    n_main = float(rng.uniform(0.025, 0.04))

    def valley(scale):
        k = int(rng.integers(7, 13))
        x = np.sort(np.concatenate(([0.0, 1.0], rng.uniform(0.05, 0.95, k - 2))))
        x[1:-1] += np.linspace(-0.01, 0.01, k - 2)                       # no tied stations
        x = np.sort(x)
        c = float(rng.uniform(0.35, 0.65))
        z = 7.0 * np.abs(x - c) ** float(rng.uniform(1.0, 2.0)) / max(c, 1 - c) ** 1.5
        z += rng.uniform(0.0, 0.25, k)
        if rng.integers(0, 2):                                             # a levee with low ground behind it
            j = int(np.clip(np.searchsorted(x, c + 0.2), 2, k - 3))
            z[j] += float(rng.uniform(1.0, 2.0))
            z[j + 1] = max(z[j] - float(rng.uniform(0.8, 1.6)), 0.3)
        z[0] = z[-1] = 9.0
        z -= z.min()
        return (x * W * scale).tolist(), z.tolist()
    sections = []
    for pos, c in enumerate([0.0, L]):
        if pos == 0 and rng.integers(0, 4) == 0:
            sections.append(dict(z_bed=S0 * L, b_main=0.4 * W, m_main=float(rng.uniform(1.0, 2.5)), n_main=n_main, bed_slope=S0))
            continue
        x, z = valley(1.0 if pos == 0 else float(rng.uniform(0.9, 1.2)))
        lim = sorted(rng.uniform(0.15, 0.85, 2) * x[-1])
        sections.append(dict(x=x, z=(np.array(z) + S0 * (L - c)).tolist(), n=n_main * (1.0 if pos == 0 else float(rng.uniform(0.9, 1.15))),
                             bed_slope=S0, roughness=[n_main * 1.6, n_main, n_main * 1.8, float(lim[0]), float(lim[1])]))
    Qb = float(np.exp(rng.uniform(np.log(0.15), np.log(1.5)))) * W
    amp = float(np.exp(rng.uniform(np.log(0.5), np.log(4.0))))
    ds_kind = ("normal_depth", "power", "fixed_depth")[rng.integers(0, 3)]
    return dict(N=N, dx=dx, dt=dt, n_steps=n_steps, theta=theta, L=L, S0=S0, family="polyline", b0=W, Qb=Qb, chain=[0.0, L],
                sections=sections, amp=amp, us_kind="flow_hydrograph", ds_kind=ds_kind, ic=("steady-state", "linear")[rng.integers(0, 2)],
                wave=dict(rise=float(rng.uniform(2, 4)) * dt, fall=float(rng.uniform(5, 9)) * dt), rc_exponent=float(rng.uniform(1.4, 2.0)))


def draw_storage_recipe(rng):
    """a short trapezoid / rectangle reach that ends in a general LumpedStorage (lumped_storage.py:24-179): area curve with
    scale and shift, optionally an outflow rating curve (power / polynomial) and entrance losses - the mass-balance root is
    a brentq on the reference side, a Brent iteration inside the kernel on ours"""
    r = draw_recipe(rng)
    while r["family"] == "compound" or r["N"] > 65 or r["N"] < 3:
        r = draw_recipe(rng)
    r["us_kind"], r["ic"] = "flow_hydrograph", ("steady-state", "GVF_equation")[rng.integers(0, 2)]
    r["ds_kind"] = "storage_curve"
    width = r["b0"]
    a0 = float(np.exp(rng.uniform(np.log(20.0), np.log(400.0)))) * width * r["dx"] / 50.0
    r["storage"] = dict(a0=a0, a1=a0 * float(rng.uniform(0.02, 0.3)), a2=a0 * float(rng.uniform(0.0, 0.02)), alpha=float(rng.uniform(0.8, 1.3)),
                        beta=float(rng.uniform(-0.3, 0.3)), rc_type=(None, "power", "polynomial")[rng.integers(0, 3)],
                        rc_exponent=float(rng.uniform(1.3, 1.9)), losses=bool(rng.integers(0, 2)),
                        reservoir_length=float(rng.uniform(100.0, 2000.0)), K_q=float(rng.uniform(0.0, 0.6)))
    return r


def draw_bend_recipe(rng):
    """three input sections along a meandering centre line (Channel.set_coords): the reference derives a curvature for the
    middle section from the turning of the line (channel.py:243-277), interpolates it to the nodes, and the energy slope
    gains the transverse-circulation term (hydraulics.py:94-153)"""
    r = draw_recipe(rng)
    while len(r["chain"]) != 3 or r["N"] < 9 or r["L"] > 12000.0:        # (the three-point curvature is at most 4 / L)
        r = draw_recipe(rng)
    L = r["L"]
    k = int(rng.integers(4, 8))
    s = np.sort(np.concatenate(([0.0, L], rng.uniform(0.05, 0.95, k - 2) * L)))
    heading = np.cumsum(rng.uniform(-1.0, 1.0, k))
    step = np.diff(s, prepend=0.0)
    xy = np.column_stack([np.cumsum(step * np.cos(heading)), np.cumsum(step * np.sin(heading))])
    r["coords"] = dict(xy=xy.tolist(), chainages=s.tolist())
    return r


def draw_near_critical_recipe(rng):
    """a steep, smooth prismatic or compound reach whose base flow runs near or beyond critical depth (Froude number of the
    normal flow 0.9 ... 1.5), long enough (65 ... 513 nodes) for the supercritical regime to show in the conditioning of the
    Newton systems.  The scheme and the reference are built for subcritical rivers (the GVF set-up refuses such a profile,
    channel.py:327-332, so the initial condition is the steady state); the reference still runs these, and its answers pin
    what the kernel - whose elimination does not pivot - returns there, or that it raises FS_ILL_CONDITIONED."""
    N = int(rng.choice([65, 129, 200, 257, 400, 513]))
    dx = float(int(np.exp(rng.uniform(np.log(40.0), np.log(1500.0)))))
    dt = int(np.exp(rng.uniform(np.log(30.0), np.log(800.0))))
    n_steps = int(rng.integers(3, 6))
    theta = float(rng.uniform(0.55, 1.0))
    L = (N - 1) * dx
    family = ("rect", "trap", "compound")[rng.integers(0, 3)]
    b0 = float(np.exp(rng.uniform(np.log(5.0), np.log(450.0))))
    n_main = float(rng.uniform(0.012, 0.02))
    q = float(np.exp(rng.uniform(np.log(0.5), np.log(6.0))))
    Fr = float(rng.uniform(0.9, 1.5))
    h = (q * q / (9.80665 * Fr * Fr)) ** (1.0 / 3.0)                  # wide-channel estimates: Fr^2 = q^2 / (g h^3), q = h^(5/3) sqrt(S0) / n
    S0 = float((q * n_main / h ** (5.0 / 3.0)) ** 2)
    Qb = q * b0
    sections = []
    for c in (0.0, L):
        z = S0 * (L - c)
        kw = dict(z_bed=z, b_main=b0, m_main=0.0, n_main=n_main, bed_slope=S0)
        if family != "rect":
            kw["m_main"] = float(rng.uniform(0.5, 2.5)) if c == 0.0 else sections[0]["m_main"]
        if family == "compound":
            if c == 0.0:
                kw.update(z_bank=z + float(rng.uniform(1.2, 2.5)) * h, b_fp_left=b0 * float(rng.uniform(0.5, 3)),
                          b_fp_right=b0 * float(rng.uniform(0.5, 3)), m_fp=float(rng.uniform(2, 6)),
                          n_left=n_main * float(rng.uniform(1.5, 3)), n_right=n_main * float(rng.uniform(1.5, 3)))
            else:
                kw.update({k: v for k, v in sections[0].items() if k in ("b_fp_left", "b_fp_right", "m_fp", "n_left", "n_right")})
                kw["z_bank"] = z + (sections[0]["z_bank"] - sections[0]["z_bed"])
        sections.append(kw)
    amp = float(np.exp(rng.uniform(np.log(0.1), np.log(1.5))))
    us_kind = ("flow_hydrograph", "flow_hydrograph", "stage_hydrograph", "fixed_depth")[rng.integers(0, 4)]
    ds_kind = ("normal_depth", "normal_depth", "power", "polynomial", "fixed_depth")[rng.integers(0, 5)]
    wave = dict(rise=float(rng.uniform(2, 5)) * dt, fall=float(rng.uniform(6, 12)) * dt)
    return dict(N=N, dx=dx, dt=dt, n_steps=n_steps, theta=theta, L=L, S0=S0, family=family, b0=b0, Qb=Qb, chain=[0.0, L],
                sections=sections, amp=amp, us_kind=us_kind, ds_kind=ds_kind, ic="steady-state", wave=wave,
                rc_exponent=float(rng.uniform(1.3, 2.0)), froude_target=Fr)


def make_section(kw):
    from src.hydromodel.cross_section import IrregularSection, TrapezoidalSection
    if "x" not in kw:
        return TrapezoidalSection(**kw)
    s = IrregularSection(x=np.array(kw["x"]), z=np.array(kw["z"]), n=kw["n"], bed_slope=kw["bed_slope"])
    s.set_roughness_para(tuple(kw["roughness"]))
    return s


def flood_wave(Qb, Qp, rise, fall):
    """base flow Qb with a sinusoidal rise to Qb + Qp over `rise` seconds and a cosine recession until `fall`"""
    from math import cos, pi, sin

    def f(t):
        if t <= rise:
            return Qb + 0.5 * Qp * (1.0 + sin(pi * t / rise - 0.5 * pi))
        if t <= fall:
            return Qb + 0.5 * Qp * (1.0 + cos(pi * (t - rise) / (fall - rise)))
        return Qb
    return f


def build_from_recipe(r):
    """the channel and solver of a recipe, through whichever `src.hydromodel` is importable (the reference here, the
    mirror package in the tests).  Returns (solver, upstream hydrograph, metadata the oracle needs for the boundaries)."""
    from src.hydromodel.channel import Channel
    from src.hydromodel.boundary import Boundary
    from src.hydromodel.preissmann import PreissmannSolver
    from src.hydromodel.hydrograph import Hydrograph
    from src.hydromodel.rating_curve import RatingCurve
    from src.hydromodel.lumped_storage import LumpedStorage
    sections = [make_section(kw) for kw in r["sections"]]
    Qb, dt, L = r["Qb"], r["dt"], r["L"]
    h_n = sections[-1].normal_depth(Q_target=Qb)
    h_us = sections[0].normal_depth(Q_target=Qb)
    z_us = sections[0].z_min
    extra = {"us_initial_depth": float(h_us), "ds_initial_depth": float(h_n)}
    if r["us_kind"] == "flow_hydrograph":
        hyd = Hydrograph(flood_wave(Qb, r["amp"] * Qb, r["wave"]["rise"], r["wave"]["fall"]))
        us = Boundary(condition='flow_hydrograph', bed_level=z_us, chainage=0, initial_depth=h_us, hydrograph=hyd)   # (the linear IC reads it)
    elif r["us_kind"] == "fixed_depth":       # a reservoir level held upstream (nothing to sample: the stored target is constant)
        hyd = Hydrograph(table=np.array([[0.0, float(h_us)], [1e9, float(h_us)]]))
        us = Boundary(condition='fixed_depth', bed_level=z_us, chainage=0, initial_depth=h_us * (1.0 + 0.1 * min(r["amp"], 1.0)))   # the level steps up at t = 0
        extra["us_initial_depth"] = float(us.initial_depth)
    else:
        t_end = r["n_steps"] * dt
        tab = np.array([[0.0, z_us + h_us], [0.3 * t_end, z_us + h_us * (1 + 0.25 * min(r["amp"], 1.0))],
                        [0.8 * t_end, z_us + h_us * 1.05], [2.0 * t_end, z_us + h_us]])
        hyd = Hydrograph(table=tab)
        us = Boundary(condition='stage_hydrograph', bed_level=z_us, chainage=0, initial_depth=h_us, hydrograph=hyd)
    kind = r["ds_kind"]
    if kind == "normal_depth":
        ds = Boundary(condition='normal_depth', bed_level=0.0, chainage=L, initial_depth=h_n)
    elif kind in ("power", "polynomial"):
        rc = RatingCurve()
        if kind == "power":
            be = r["rc_exponent"]
            rc.set(type='power', a=Qb / h_n ** be, b=be)
            extra.update(ds_rc_type="power", ds_rc_a=float(rc.a), ds_rc_b=float(rc.b), ds_rc_shift=0.0)
        else:
            rc.set(type='polynomial', a=0.2 * Qb / h_n ** 2, b=0.8 * Qb / h_n, c=0.0)
            extra.update(ds_rc_type="polynomial", ds_rc_a=float(rc.a), ds_rc_b=float(rc.b), ds_rc_c=0.0, ds_rc_shift=0.0)
        ds = Boundary(condition='rating_curve', bed_level=0.0, chainage=L, initial_depth=h_n, rating_curve=rc)
    else:
        ds = Boundary(condition='fixed_depth', bed_level=0.0, chainage=L, initial_depth=h_n)
        if kind == "storage_curve":
            st = r["storage"]
            stages = np.arange(0.0, 40.0 * h_n + 1e-9, 0.25 * h_n)
            curve = np.column_stack([stages, st["a0"] + st["a1"] * stages + st["a2"] * stages ** 2])
            rc = None
            extra.update(storage_curve=curve.tolist(), storage_alpha=st["alpha"], storage_beta=st["beta"], storage_min_stage=0.5 * h_n,
                         storage_rc_type=st["rc_type"])
            if st["rc_type"] == "power":
                rc = RatingCurve(); rc.set(type='power', a=0.6 * Qb / h_n ** st["rc_exponent"], b=st["rc_exponent"])
                extra.update(storage_rc=dict(a=float(rc.a), b=float(rc.b), shift=0.0))
            elif st["rc_type"] == "polynomial":
                rc = RatingCurve(); rc.set(type='polynomial', a=0.1 * Qb / h_n ** 2, b=0.5 * Qb / h_n, c=0.0)
                extra.update(storage_rc=dict(a=float(rc.a), b=float(rc.b), c=0.0, shift=0.0))
            ss = LumpedStorage(surface_area=None, min_stage=0.5 * h_n, solution_boundaries=(0, 40.0 * h_n), rating_curve=rc)
            ss.set_area_curve(curve, alpha=st["alpha"], beta=st["beta"])
            if st["losses"]:
                ss.capture_losses, ss.reservoir_length, ss.K_q = True, st["reservoir_length"], st["K_q"]
                extra.update(storage_losses=dict(reservoir_length=st["reservoir_length"], K_q=st["K_q"]))
            extra["storage_bounds"] = [float(ss.Y_min), float(ss.Y_max)]
            ds.set_lumped_storage(ss)
        if kind == "storage":
            area = float(max(30.0 * r["b0"] * L / 40.0, 5e3))
            ds.set_lumped_storage(LumpedStorage(surface_area=area, min_stage=0.5 * h_n, solution_boundaries=(0.0, 60.0 * h_n)))
            extra.update(storage_area=area, storage_min_stage=0.5 * h_n, storage_bounds=[0.0, 60.0 * h_n])
    ch = Channel(initial_flow=Qb, upstream_boundary=us, downstream_boundary=ds, interpolation_method=r["ic"])
    if "coords" in r:
        ch.set_coords(coords=np.array(r["coords"]["xy"]), chainages=np.array(r["coords"]["chainages"]))
    ch.set_cross_sections(r["chain"], sections)
    sol = PreissmannSolver(channel=ch, theta=r["theta"], time_step=dt, spatial_step=r["dx"], simulation_time=r["n_steps"] * dt)
    extra["h_n"] = float(h_n)
    return sol, hyd, extra


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=48)
    ap.add_argument("--polyline", type=int, default=12)
    ap.add_argument("--storage", type=int, default=12)
    ap.add_argument("--bends", type=int, default=8)
    ap.add_argument("--near-critical", type=int, default=0,
                    help="this many near-critical / supercritical reaches INSTEAD of the sweep (tests/golden/near_critical.npz): "
                         "for each case the fixture also keeps the 1-norm condition number of the reference's Jacobian at the first "
                         "Newton iteration of every level; half of the cases are the worst-conditioned of the draws")
    ap.add_argument("--seed", type=int, default=20260301)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "random_sweep.npz"),
                    help="a soak run writes elsewhere (e.g. gpurun_out/sweep_soak.npz, FS_SWEEP_FIXTURE for the tests)")
    a = ap.parse_args()
    out_path = os.path.abspath(a.out)              # before the chdir below: nothing is ever written under /root/reference
    assert not out_path.startswith("/root/reference")
    sys.dont_write_bytecode = True
    os.environ.setdefault("MPLBACKEND", "Agg")
    os.chdir("/root/reference")
    sys.path.insert(0, "/root/reference")
    from gen_golden import base_meta, run_and_capture, sample_targets

    if a.near_critical:
        return near_critical(a, out_path, base_meta, run_and_capture, sample_targets)
    rng = np.random.default_rng(a.seed)
    rng_poly = np.random.default_rng(a.seed + 1)          # a stream of its own: the trapezoid-family cases stay what they were
    arrays, metas, tried, t0 = {}, [], 0, time.time()
    rng_store = np.random.default_rng(a.seed + 2)
    rng_bend = np.random.default_rng(a.seed + 3)
    width = 2 if a.cases + a.polyline + a.storage + a.bends <= 100 else 4
    total = a.cases + a.polyline + a.storage + a.bends
    while len(metas) < total and tried < 6 * total:
        tried += 1
        n = len(metas)
        recipe = (draw_recipe(rng) if n < a.cases else draw_polyline_recipe(rng_poly) if n < a.cases + a.polyline
                  else draw_storage_recipe(rng_store) if n < a.cases + a.polyline + a.storage else draw_bend_recipe(rng_bend))
        try:
            sol, hyd, extra = build_from_recipe(recipe)
            out, wall = run_and_capture(sol, 1e-6, slim=True)
        except (ValueError, RuntimeError, ZeroDivisionError, FloatingPointError) as e:      # the reference gives up on this draw
            print(f"  draw {tried}: reference raised {type(e).__name__}: {str(e)[:70]}")
            continue
        if not (np.all(np.isfinite(out["depth"])) and np.min(out["depth"]) > 0):
            continue
        i = len(metas)
        out["us_target"] = sample_targets(hyd, sol.number_of_time_levels, sol.time_step)
        for k in ("R0", "norm_level", "norm_value", "final_unknowns"):
            out.pop(k, None)
        for k, v in out.items():
            arrays[f"c{i:0{width}d}_{k}"] = v
        m = base_meta(sol, 1e-6, wall, **extra)
        m.update(family=recipe["family"], n_sections=len(recipe["chain"]), us_kind=recipe["us_kind"], ds_kind=recipe["ds_kind"],
                 ic=recipe["ic"], Qb=recipe["Qb"], amp=recipe["amp"], bends="coords" in recipe, recipe=recipe)
        metas.append(m)
        print(f"  case {i:02d}: N={m['N']:4d} nt={m['nt']} {m['family']:8s} x{m['n_sections']} {m['us_kind'][:5]} -> {m['ds_kind']:12s} "
              f"ic={m['ic']:12s} its={out['iters'][1:].tolist()}")
    meta = dict(generator="oracle/gen_random_sweep.py", reference="cve-mohd/flow-sim snapshot 2026-02-13, run in the build container",
                seed=a.seed, draws=tried, cases=metas)
    path = out_path
    np.savez_compressed(path, meta=np.array(json.dumps(meta)), **arrays)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB): {len(metas)} cases of {tried} draws in {time.time() - t0:.0f} s")


def near_critical(a, out_path, base_meta, run_and_capture, sample_targets):
    """draws 4 x the wanted number of near-critical reaches, runs the reference on each and keeps the worst-conditioned
    half of the wanted number plus an even spread of the rest"""
    import scipy.sparse.linalg as spla
    if out_path.endswith("random_sweep.npz"):
        out_path = os.path.join(os.path.dirname(out_path), "near_critical.npz")
    rng = np.random.default_rng(a.seed + 7)
    pool, tried, t0 = [], 0, time.time()
    while len(pool) < 4 * a.near_critical and tried < 12 * a.near_critical:
        tried += 1
        recipe = draw_near_critical_recipe(rng)
        try:
            sol, hyd, extra = build_from_recipe(recipe)
            conds, seen = [], set()
            orig = spla.spsolve

            def spy(J, b, *aa, _sol=sol, **kk):      # condition number of the reference's own matrix, first iteration of a level
                if _sol.time_level not in seen:
                    seen.add(_sol.time_level)
                    conds.append(float(np.linalg.cond(J.toarray(), 1)))
                return orig(J, b, *aa, **kk)
            spla.spsolve = spy
            try:
                out, wall = run_and_capture(sol, 1e-6, slim=True)
            finally:
                spla.spsolve = orig
        except (ValueError, RuntimeError, ZeroDivisionError, FloatingPointError) as e:
            print(f"  draw {tried}: reference raised {type(e).__name__}: {str(e)[:70]}")
            continue
        if not (np.all(np.isfinite(out["depth"])) and np.min(out["depth"]) > 0):
            continue
        h = out["depth"]; Q = out["flow"]
        xs = sol.channel.xs_at_node
        fr = max(abs(Q[k, i]) / xs[i].area(xs[i].z_min + h[k, i]) / (9.80665 * xs[i].area(xs[i].z_min + h[k, i]) / xs[i].top_width(xs[i].z_min + h[k, i])) ** 0.5
                 for k in range(h.shape[0]) for i in range(0, h.shape[1], max(1, h.shape[1] // 16)))
        out["us_target"] = sample_targets(hyd, sol.number_of_time_levels, sol.time_step)
        for k in ("R0", "norm_level", "norm_value", "final_unknowns"):
            out.pop(k, None)
        out["cond1"] = np.array(conds)
        m = base_meta(sol, 1e-6, wall, **extra)
        m.update(family=recipe["family"], n_sections=2, us_kind=recipe["us_kind"], ds_kind=recipe["ds_kind"], ic=recipe["ic"],
                 Qb=recipe["Qb"], amp=recipe["amp"], bends=False, recipe=recipe, froude_max=float(fr), cond1_max=float(max(conds)))
        pool.append((out, m))
        print(f"  draw {tried}: N={m['N']:4d} {m['family']:8s} {m['us_kind'][:5]} -> {m['ds_kind']:12s} Fr<={fr:.2f} cond1 {max(conds):.1e} "
              f"its={out['iters'][1:].tolist()}")
    pool.sort(key=lambda om: -om[1]["cond1_max"])
    half = a.near_critical // 2
    rest = pool[half:]
    keep = pool[:half] + [rest[int(i * len(rest) / (a.near_critical - half))] for i in range(a.near_critical - half)]
    arrays, metas = {}, []
    for i, (out, m) in enumerate(keep):
        for k, v in out.items():
            arrays[f"c{i:02d}_{k}"] = v
        metas.append(m)
    meta = dict(generator="oracle/gen_random_sweep.py --near-critical", reference="cve-mohd/flow-sim snapshot 2026-02-13, run in the build container",
                seed=a.seed + 7, draws=tried, cases=metas)
    np.savez_compressed(out_path, meta=np.array(json.dumps(meta)), **arrays)
    print(f"wrote {out_path} ({os.path.getsize(out_path) / 1024:.0f} KiB): {len(metas)} cases of {tried} draws in {time.time() - t0:.0f} s; "
          f"cond1 from {min(m['cond1_max'] for m in metas):.1e} to {max(m['cond1_max'] for m in metas):.1e}")


if __name__ == "__main__":
    main()
