"""Seeded random sweep: prismatic reaches drawn far outside the benchmark's parameter boxes - node counts 2 ... 700
(every lane / wave layout, ragged tails), theta 0.55 ... 1, time steps 30 s ... 1 h, spatial steps 25 m ... 2 km (Courant
numbers 0.3 ... 300), slopes 5e-5 ... 5e-3, widths 3 ... 600 m, Manning 0.012 ... 0.08, rectangles and trapezoids, every
pair of closed-form boundary kinds, flood waves of 10 % ... 300 % of the base flow - against the C oracle (pivoted banded
LU, the reference's algorithm).  fp64: 1e-8 relative on the whole history, identical Newton counts; the draw is part of
the test (seeded), a draw the ORACLE cannot solve is skipped and counted.

Supercritical draws are part of the sweep: there the Newton systems can be ill-conditioned (cond 1e13 ... 1e16: one boundary
condition per end is not what such a flow takes) and any two solvers part by 1e-6 ... 1e-2 - the reference and its own
restatements included (tests/test_near_critical.py pins that against the reference itself).  The rule for every draw is
therefore: 1e-8 with identical Newton counts, OR the kernel says so - status FS_ILL_CONDITIONED, raised by the conditioning
monitor of its elimination (include/flowsim_abi.h).  A last test keeps the monitor honest: it may flag only a small share
of the draws, and none of the clearly subcritical ones."""
import os

import numpy as np
import pytest

from oracle import preissmann_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-8
N_CASES = int(os.environ.get("FS_SWEEP_CASES", 384))          # a soak run: FS_SWEEP_CASES=8000 FS_SWEEP_TABLE=3000 pytest ...

US_KINDS = ("flow", "flow", "flow", "stage", "fixed", "rating_us")
DS_KINDS = ("normal", "normal", "power", "poly", "fixed", "stage_ds", "flow_ds", "storage", "blend")


def rel_err(got, want, floor):
    return float(np.max(np.abs(got - want) / np.maximum(np.abs(want), floor)))


def wave(nt, dt, rise, fall, amp):
    from synth import akbari_shape
    return np.array([akbari_shape(1.0, amp, rise, fall, k * dt) for k in range(nt)])


def random_problem(seed, nodes=None):
    from synth import normal_depth_rect, normal_depth_trap
    rng = np.random.default_rng(770000 + seed)
    N = int(rng.choice([2, 3, 5, 17, 63, 64, 65, 127, 128, 129, 200, 255, 256, 257, 400, 511, 512, 513, 700]))
    if nodes is not None:
        N = int(nodes)             # (the long-reach sweep below: the same draws on a reach of its own length)
    trapezoid = bool(rng.integers(0, 2))
    b = float(np.exp(rng.uniform(np.log(3.0), np.log(600.0))))
    m = float(rng.uniform(0.5, 3.0)) if trapezoid else 0.0
    n = float(rng.uniform(0.012, 0.08))
    S0 = float(np.exp(rng.uniform(np.log(5e-5), np.log(5e-3))))
    q = float(np.exp(rng.uniform(np.log(0.05), np.log(8.0))))          # base flow per metre of bed width
    Qb = q * b
    theta = float(rng.uniform(0.55, 1.0))
    dt = float(np.exp(rng.uniform(np.log(30.0), np.log(3600.0))))
    dx = float(np.exp(rng.uniform(np.log(25.0), np.log(2000.0))))
    n_steps = int(rng.integers(2, 7))
    amp = float(np.exp(rng.uniform(np.log(0.1), np.log(3.0))))
    L = (N - 1) * dx
    geo = {k: np.zeros(N) for k in O.GEO_KEYS}
    geo["b_main"][:] = b; geo["m_main"][:] = m
    geo["n_main"][:] = n; geo["n_left"][:] = n; geo["n_right"][:] = n
    geo["z_bed"] = S0 * L * (1 - np.arange(N) / (N - 1))
    hn = normal_depth_trap(b, m, n, S0, Qb) if trapezoid else normal_depth_rect(b, n, S0, Qb)
    nt = n_steps + 1
    shape = wave(nt, dt, rng.uniform(2, 6) * dt, rng.uniform(7, 14) * dt, amp)        # 1 ... 1 + amp ... 1
    us_k, ds_k = US_KINDS[rng.integers(0, len(US_KINDS))], DS_KINDS[rng.integers(0, len(DS_KINDS))]
    if ds_k == "storage" and N > 130:
        ds_k = "normal"            # the reservoir row's level-1 quirk (boundary.py:104-108): short reaches only
    if us_k != "flow" and ds_k in ("flow_ds",):
        ds_k = "normal"            # a flow imposed at the outlet needs a flow imposed at the inlet to stay well posed here
    zu = S0 * L
    if us_k == "flow":
        us = O.BC("flow_hydrograph", bed_level=zu, target=Qb * shape)
    elif us_k == "stage":
        us = O.BC("stage_hydrograph", bed_level=zu, target=zu + hn * (1.0 + 0.3 * np.minimum(amp, 1.0) * (shape - 1.0) / amp))
    elif us_k == "fixed":
        us = O.BC("fixed_depth", bed_level=zu, initial_depth=hn)
    else:
        us = O.BC("rating_curve", bed_level=zu, rc_type="polynomial", rc=dict(a=0.0, b=-0.2 * Qb, c=Qb + 0.2 * Qb * (zu + hn), shift=0.0))
    if ds_k == "normal":
        ds = O.BC("normal_depth", bed_level=0.0, bed_slope=S0)
    elif ds_k == "fixed":
        ds = O.BC("fixed_depth", bed_level=0.0, initial_depth=hn)
    elif ds_k == "poly":
        ds = O.BC("rating_curve", bed_level=0.0, rc_type="polynomial", rc=dict(a=0.15 * Qb / hn ** 2, b=0.85 * Qb / hn, c=0.0, shift=0.0))
    elif ds_k == "power":
        be = float(rng.uniform(1.2, 2.2))
        ds = O.BC("rating_curve", bed_level=0.0, rc_type="power", rc=dict(a=Qb / hn ** be, b=be, shift=0.0))
    elif ds_k == "flow_ds":
        ds = O.BC("flow_hydrograph", bed_level=0.0, target=Qb * (1.0 - 0.25 * (shape - 1.0) / amp))
    elif ds_k == "stage_ds":
        ds = O.BC("stage_hydrograph", bed_level=0.0, target=hn * (1.0 + 0.2 * (shape - 1.0) / amp))
    elif ds_k == "storage":
        ds = O.BC("fixed_depth", bed_level=0.0, initial_depth=hn,
                  storage=dict(area=max(40.0 * b * L / 50.0, 1e3), min_stage=0.5 * hn, Y_min=0.0, Y_max=50.0 * hn))
    else:
        lo = [0.0, 0.9 * Qb / hn, 0.1 * Qb / hn ** 2]
        ds = O.BC("rating_curve", bed_level=0.0, rc_type="blend",
                  rc=dict(initial_stage=hn, buffer=0.5, low=lo, high=[2 * v for v in lo], dY=1e-3))
    p = O.Problem(geo=geo, h0=np.full(N, hn), Q0=np.full(N, Qb), us=us, ds=ds, theta=theta, dt=dt, dx=dx, nt=nt, tol=1e-6)
    info = dict(N=N, trapezoid=trapezoid, b=b, m=m, n=n, S0=S0, Qb=Qb, theta=theta, dt=dt, dx=dx, amp=amp, us=us_k, ds=ds_k, hn=hn,
                courant=(Qb / (b * hn) + (9.80665 * hn) ** 0.5) * dt / dx, froude=Qb / (b * hn) / (9.80665 * hn) ** 0.5)
    return p, info


def random_table_problem(seed):
    """non-prismatic channel with compound sections (flood plains the wave reaches in some draws) and bends: width, side
    slope, bed slope, roughness, bankfull depth and curvature vary along the reach"""
    from synth import normal_depth_trap
    rng = np.random.default_rng(880000 + seed)
    N = int(rng.choice([3, 9, 33, 64, 65, 121, 128, 129, 200, 256, 257, 400, 513]))
    x = np.linspace(0.0, 1.0, N)
    smooth = lambda lo, hi: lo + (hi - lo) * (0.5 + 0.5 * np.sin(2 * np.pi * (rng.uniform(0.3, 1.5) * x + rng.uniform())))
    b0 = float(np.exp(rng.uniform(np.log(8.0), np.log(300.0))))
    b = b0 * smooth(0.85, 1.15)
    m = smooth(*sorted(rng.uniform(0.5, 2.5, 2)))
    nm = smooth(*sorted(rng.uniform(0.02, 0.045, 2)))
    S0 = float(np.exp(rng.uniform(np.log(1e-4), np.log(2e-3))))
    dx = float(np.exp(rng.uniform(np.log(100.0), np.log(1500.0))))
    dt = float(np.exp(rng.uniform(np.log(120.0), np.log(3600.0))))
    theta = float(rng.uniform(0.55, 1.0))
    q = float(np.exp(rng.uniform(np.log(0.3), np.log(5.0))))
    Qb = q * b0
    hn = normal_depth_trap(b0, float(m.mean()), float(nm.mean()), S0, Qb)
    geo = {k: np.zeros(N) for k in O.GEO_KEYS}
    geo["b_main"], geo["m_main"], geo["n_main"] = b, m, nm
    geo["n_left"] = nm * rng.uniform(1.2, 2.0); geo["n_right"] = nm * rng.uniform(1.2, 2.0)
    geo["z_bed"] = S0 * (N - 1) * dx * (1 - x) + 0.05 * hn * np.sin(6 * np.pi * x)
    geo["is_compound"][:] = 1.0
    geo["h_bf"] = hn * smooth(*sorted(rng.uniform(0.9, 2.2, 2)))          # some nodes over bank from the start, some never
    geo["b_fp_l"] = b * rng.uniform(0.5, 4.0); geo["b_fp_r"] = b * rng.uniform(0.5, 4.0)
    geo["m_fp"][:] = rng.uniform(2.0, 6.0)
    geo["curvature"] = rng.uniform(0.0, 2e-3) * np.sin(2 * np.pi * rng.uniform(0.5, 2.0) * x)
    n_steps = int(rng.integers(2, 6))
    nt = n_steps + 1
    amp = float(np.exp(rng.uniform(np.log(0.2), np.log(2.5))))
    shape = wave(nt, dt, rng.uniform(2, 5) * dt, rng.uniform(6, 12) * dt, amp)
    ds_k = ("normal", "power", "poly", "fixed", "stage_ds", "blend")[rng.integers(0, 6)]
    us = O.BC("flow_hydrograph", bed_level=float(geo["z_bed"][0]), target=Qb * shape)
    zb = float(geo["z_bed"][-1])
    if ds_k == "normal":
        ds = O.BC("normal_depth", bed_level=zb, bed_slope=S0)
    elif ds_k == "fixed":
        ds = O.BC("fixed_depth", bed_level=zb, initial_depth=hn)
    elif ds_k == "poly":
        ds = O.BC("rating_curve", bed_level=zb, rc_type="polynomial", rc=dict(a=0.15 * Qb / hn ** 2, b=0.85 * Qb / hn, c=0.0, shift=-zb))
    elif ds_k == "power":
        ds = O.BC("rating_curve", bed_level=zb, rc_type="power", rc=dict(a=Qb / hn ** 1.5, b=1.5, shift=-zb))
    elif ds_k == "stage_ds":
        ds = O.BC("stage_hydrograph", bed_level=zb, target=zb + hn * (1.0 + 0.2 * (shape - 1.0) / amp))
    else:
        lo = [0.0, 0.9 * Qb / hn, 0.1 * Qb / hn ** 2]
        ds = O.BC("rating_curve", bed_level=zb, rc_type="blend",
                  rc=dict(initial_stage=zb + hn, buffer=0.5, low=[lo[0] - lo[1] * zb + lo[2] * zb * zb, lo[1] - 2 * lo[2] * zb, lo[2]],
                          high=[2 * (lo[0] - lo[1] * zb + lo[2] * zb * zb), 2 * (lo[1] - 2 * lo[2] * zb), 2 * lo[2]], dY=1e-3))
    p = O.Problem(geo=geo, h0=np.full(N, hn), Q0=np.full(N, Qb), us=us, ds=ds, theta=theta, dt=dt, dx=dx, nt=nt, tol=1e-6)
    return p, dict(N=N, b0=b0, S0=S0, Qb=Qb, hn=hn, theta=theta, dt=dt, dx=dx, amp=amp, ds=ds_k, seed=seed)


_solved = []
_solved_table = []
_flagged = []          # (sweep, seed, Froude number or None, error) of every draw the conditioning monitor flagged
_unflagged_froude = []
ILL = 4                # FS_ILL_CONDITIONED
FLAGGED_TOL = 1e-2     # a flagged draw may miss 1e-8, not the flood wave (tests/test_near_critical.py: the same bound against the reference)
N_TABLE = int(os.environ.get("FS_SWEEP_TABLE", 128))


@pytest.mark.parametrize("seed", range(N_TABLE))
def test_random_compound_channel_against_the_oracle(seed):
    from fixture_batch import batch_from_problems
    from oracle import c_oracle as CO
    p, info = random_table_problem(seed)
    ref = CO.run(p)
    if ref["status"] != 0 or not np.all(np.isfinite(ref["depth"])) or np.min(ref["depth"]) <= 1e-3 * info["hn"]:
        _solved_table.append(False)
        pytest.skip(f"the oracle does not get through this draw (status {ref['status']}): {info}")
    _solved_table.append(True)
    with batch_from_problems([p], mode="table", history=True) as b:
        b.step(p.nt - 1)
        st = int(b.status()[0])
        assert st in (0, ILL), (st, info)
        h, Q = b.history_arrays(0, p.nt)
        its = b.iterations(0, p.nt)[:, 0]
    d, f = ref["depth"], ref["flow"]
    eh, eq = rel_err(h[:, 0], d, 1e-3 * info["hn"]), rel_err(Q[:, 0], f, 1e-3 * info["Qb"])
    if st == ILL:
        # flagged is not a free pass: the kernel says 1e-8 is not promised, the flood wave must still be the same one (the bound
        # tests/test_near_critical.py holds flagged reaches to against the reference itself)
        _flagged.append(("table", seed, None, max(eh, eq)))
        assert eh <= FLAGGED_TOL and eq <= FLAGGED_TOL, ("flagged draw off by more than 1e-2", eh, eq, info)
    else:
        assert eh <= TOL and eq <= TOL, (eh, eq, info)
        assert np.array_equal(its, ref["iters"]), (its, ref["iters"], info)
    over = np.any(d > p.geo["h_bf"][None, :]) and np.any(d < p.geo["h_bf"][None, :])
    info["both_sides_of_bankfull"] = bool(over)


@pytest.mark.parametrize("seed", range(N_CASES))
def test_random_reach_against_the_oracle(seed):
    from fixture_batch import batch_from_problems
    from oracle import c_oracle as CO
    p, info = random_problem(seed)
    ref = CO.run(p)
    if ref["status"] != 0 or not np.all(np.isfinite(ref["depth"])) or np.min(ref["depth"]) <= 1e-3 * info["hn"]:
        _solved.append(False)
        pytest.skip(f"the oracle does not get through this draw (status {ref['status']}): {info}")
    _solved.append(True)
    mode = "rect_uniform" if (not info["trapezoid"] and info["ds"] != "blend") else ("trap_uniform" if info["trapezoid"] and seed % 2 else "table")
    with batch_from_problems([p], mode=mode, history=True) as b:
        b.step(p.nt - 1)
        st = int(b.status()[0])
        assert st in (0, ILL), (st, info)
        h, Q = b.history_arrays(0, p.nt)
        its = b.iterations(0, p.nt)[:, 0]
    d, f = ref["depth"], ref["flow"]
    eh, eq = rel_err(h[:, 0], d, 1e-3 * info["hn"]), rel_err(Q[:, 0], f, 1e-3 * info["Qb"])
    if st == ILL:
        # the kernel itself says that this reach is beyond what its unpivoted elimination (or any solver, at 1e-8) stands for
        _flagged.append(("prismatic", seed, info["froude"], max(eh, eq)))
        assert eh <= FLAGGED_TOL and eq <= FLAGGED_TOL, ("flagged draw off by more than 1e-2", eh, eq, info)
        return
    _unflagged_froude.append(info["froude"])
    assert eh <= TOL and eq <= TOL, (eh, eq, info)
    assert np.array_equal(its, ref["iters"]), (its, ref["iters"], info)


N_LONG = int(os.environ.get("FS_SWEEP_LONG", 20))
_long_kernels = []


@pytest.mark.parametrize("seed", range(N_LONG))
def test_random_long_reach_against_the_oracle(seed, monkeypatch):
    """The same draws on reaches LONGER than one lane grid (4 097 ... 32 768 nodes: Courant numbers, slopes, boundary pairs and flood
    waves as above), on both kernels that take them: the team of workgroups (uniform sections; fs_kernel.hpp, TEAM) and the
    multi-pass kernel (FS_NO_TEAM=1, and the table mode; fs_long.hpp).  Node counts on and off the lane-grid multiples - a last
    member with one row, a full team, a ragged tail."""
    from fixture_batch import batch_from_problems
    from flowsim_amd import _abi as A
    from oracle import c_oracle as CO
    rng = np.random.default_rng(660000 + seed)
    special = (4097, 8192, 8193, 12288, 16384, 16385, 32768, 32767)
    N = special[seed] if seed < len(special) else int(rng.integers(4098, 24000))
    p, info = random_problem(5000 + seed, nodes=N)
    ref = CO.run(p)
    if ref["status"] != 0 or not np.all(np.isfinite(ref["depth"])) or np.min(ref["depth"]) <= 1e-3 * info["hn"]:
        pytest.skip(f"the oracle does not get through this draw (status {ref['status']}): {info}")
    d, f = ref["depth"], ref["flow"]
    uniform = "trap_uniform" if info["trapezoid"] else "rect_uniform"
    runs = [(uniform, False), (uniform, True)] if info["ds"] != "blend" else []
    if N <= 16384:
        runs.append(("table", False))
    for mode, no_team in runs:
        if no_team:
            monkeypatch.setenv("FS_NO_TEAM", "1")
        else:
            monkeypatch.delenv("FS_NO_TEAM", raising=False)
        with batch_from_problems([p], mode=mode, history=True) as b:
            b.step(p.nt - 1)
            st = int(b.status()[0])
            e = A.kernel_table()[b.kernel_index()]
            assert st in (0, ILL), (st, mode, no_team, info)
            assert (e["team"], e["long_reach"]) == ((1, 0) if (mode != "table" and not no_team) else (0, 1)), (e, mode, no_team)
            _long_kernels.append((e["team"], e["long_reach"]))
            h, Q = b.history_arrays(0, p.nt)
            its = b.iterations(0, p.nt)[:, 0]
        eh, eq = rel_err(h[:, 0], d, 1e-3 * info["hn"]), rel_err(Q[:, 0], f, 1e-3 * info["Qb"])
        if st == ILL:
            assert eh <= FLAGGED_TOL and eq <= FLAGGED_TOL, ("flagged draw off by more than 1e-2", eh, eq, mode, no_team, info)
            continue
        assert eh <= TOL and eq <= TOL, (eh, eq, mode, no_team, info)
        assert np.array_equal(its, ref["iters"]), (its, ref["iters"], mode, no_team, info)


def test_the_monitor_flags_few_draws_and_no_subcritical_prismatic_one():
    if len(_solved) < N_CASES or len(_solved_table) < N_TABLE:
        pytest.skip("runs after the sweeps")
    prism = [f for f in _flagged if f[0] == "prismatic"]
    assert all(fr >= 0.85 for _, _, fr, _ in prism), prism            # a flag on a prismatic reach means supercritical (or nearly) flow
    assert len(prism) <= 0.02 * N_CASES + 2 and len(_flagged) - len(prism) <= 0.03 * N_TABLE + 2, _flagged
    assert sum(fr >= 0.9 for fr in _unflagged_froude) >= 1 or N_CASES < 300        # supercritical draws that pass at 1e-8 unflagged exist too
    # what the flagged draws deviate by (each was held to FLAGGED_TOL above): printed with -s / in the captured output of a failure
    for sweep, seed, fr, err in sorted(_flagged, key=lambda f: -f[3]):
        print(f"flagged: {sweep} seed {seed} Froude {fr if fr is None else round(fr, 3)} deviation from the pivoted oracle {err:.3e}")
    assert all(err <= FLAGGED_TOL for *_, err in _flagged)


def tidal_problem(seed):
    """a mild reach whose downstream stage rises fast enough to push water back up the channel: the flow changes sign
    along the reach and in time (Q|Q| in the friction slope, the sign of the advective term, negative boundary flows)"""
    from synth import normal_depth_rect, normal_depth_trap
    rng = np.random.default_rng(990000 + seed)
    N = int(rng.choice([9, 33, 64, 65, 129, 256, 400]))
    trapezoid = bool(rng.integers(0, 2))
    b = float(rng.uniform(20.0, 200.0)); m = float(rng.uniform(1.0, 2.5)) if trapezoid else 0.0
    n = float(rng.uniform(0.02, 0.04)); S0 = float(np.exp(rng.uniform(np.log(3e-5), np.log(2e-4))))
    Qb = float(rng.uniform(0.05, 0.3)) * b
    dx = float(rng.uniform(200.0, 800.0)); dt = float(rng.uniform(120.0, 600.0)); theta = float(rng.uniform(0.6, 1.0))
    hn = normal_depth_trap(b, m, n, S0, Qb) if trapezoid else normal_depth_rect(b, n, S0, Qb)
    L = (N - 1) * dx
    geo = {k: np.zeros(N) for k in O.GEO_KEYS}
    geo["b_main"][:] = b; geo["m_main"][:] = m
    geo["n_main"][:] = n; geo["n_left"][:] = n; geo["n_right"][:] = n
    geo["z_bed"] = S0 * L * (1 - np.arange(N) / (N - 1))
    nt = 9
    rise = float(rng.uniform(0.5, 1.5))
    tide = hn * (1.0 + rise * np.sin(np.pi * np.arange(nt) / (nt - 1)) ** 2)
    us = O.BC("flow_hydrograph", bed_level=S0 * L, target=np.full(nt, Qb))
    ds = O.BC("stage_hydrograph", bed_level=0.0, target=tide)
    p = O.Problem(geo=geo, h0=np.full(N, hn), Q0=np.full(N, Qb), us=us, ds=ds, theta=theta, dt=dt, dx=dx, nt=nt, tol=1e-6)
    return p, dict(N=N, trapezoid=trapezoid, b=b, Qb=Qb, hn=hn, rise=rise, dt=dt, dx=dx, seed=seed)


_reversed = []


@pytest.mark.parametrize("seed", range(24))
def test_flow_reversal_against_the_oracle(seed):
    from fixture_batch import batch_from_problems
    from oracle import c_oracle as CO
    p, info = tidal_problem(seed)
    ref = CO.run(p)
    if ref["status"] != 0:
        pytest.skip(f"the oracle does not get through this draw: {info}")
    _reversed.append(bool(np.min(ref["flow"]) < -0.05 * info["Qb"]))
    for mode in (("trap_uniform",) if info["trapezoid"] else ("rect_uniform", "table")):
        with batch_from_problems([p], mode=mode, history=True) as b:
            b.step(p.nt - 1)
            assert np.all(b.status() == 0), (b.status(), info)
            h, Q = b.history_arrays(0, p.nt)
            its = b.iterations(0, p.nt)[:, 0]
        assert rel_err(h[:, 0], ref["depth"], 1e-3 * info["hn"]) <= TOL and rel_err(Q[:, 0], ref["flow"], 1e-3 * info["Qb"]) <= TOL, info
        assert np.array_equal(its, ref["iters"]), info


def test_the_tide_does_reverse_the_flow():
    if len(_reversed) < 12:
        pytest.skip("runs after the tidal cases")
    assert sum(_reversed) >= len(_reversed) // 2, _reversed


def test_most_draws_are_solvable():
    """the sweep means something only if the reference's algorithm itself gets through most of it"""
    if len(_solved) < N_CASES or len(_solved_table) < N_TABLE:
        pytest.skip("runs after the sweeps")
    assert sum(_solved) >= 0.8 * N_CASES, f"{sum(_solved)} of {N_CASES}"
    assert sum(_solved_table) >= 0.7 * N_TABLE, f"{sum(_solved_table)} of {N_TABLE}"
