"""Every instantiation of the step kernel (csrc/fs_entries.hpp, exported as the dispatch table of the C ABI) is
launched once and checked against the CPU oracle: fp64 to 1e-8 with identical Newton counts, fp32 to 5e-4 of the
fp64 answer.

The case for an entry follows from its attributes: section mode -> channel family, (cells per lane x waves per reach,
full / ragged) -> node count (full chunks: exactly the capacity; ragged: a few cells short of it, so the padding and
the lane that owns the last node move around), boundary class -> the boundary pair it was compiled for (class -1: the
general reservoir row it exists for; class 0 / 1: a different pair per shape so that every closed-form row and every
upstream kind runs somewhere).  FS_KERNEL_INDEX makes fs_batch_step use exactly that entry or fail; the test also
fails for an entry it has no recipe for - adding an instantiation without coverage is an error."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import preissmann_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-8
TOL_F32 = 5e-4


def rel_err(got, want, floor):
    return float(np.max(np.abs(got - want) / np.maximum(np.abs(want), floor)))


def _table():
    try:
        from flowsim_amd import _abi as A
        return A.kernel_table()
    except Exception:            # library not built: collection must not fail (the gpu run builds first)
        return []


TABLE = _table()


def _id(e):
    sec = ("rect", "trap", "table", "irr")[e["section_mode"]]
    return (f"{e['index']:03d}-{'f64' if e['dtype'] == 0 else 'f32'}-{sec}-{e['cells_per_thread']}x{e['waves_per_reach']}"
            f"{'-full' if e['full'] else ''}-bc{e['boundary_class']}{'' if e['diag'] else '-nodiag'}{'-long' if e.get('long_reach') else ''}"
            f"{'-tail%d' % e['tail'] if e.get('tail', -1) >= 0 else ''}{'-team' if e.get('team') else ''}")


def _nodes(e):
    cap = 64 * e["cells_per_thread"] * e["waves_per_reach"]      # rows of the scalar system: N - 1 cells + the boundary row
    if e.get("team"):
        return 3 * cap if e["full"] else 2 * cap + cap // 5 + e["index"] % 9            # a team of three workgroups (ragged: the last one mostly padding)
    if e.get("long_reach"):
        return 2 * cap + cap // 3 + e["index"] % 7            # the multi-pass kernel (fs_long.hpp): three passes, the last one ragged
    if e["full"]:
        return cap
    n = max(2, cap - 1 - e["index"] % 5)             # ragged: 1..5 rows short of the capacity
    while e.get("tail", -1) >= 0 and (n - 1) % e["cells_per_thread"] != e["tail"]:
        n -= 1                                       # tail-only form: the boundary row sits in local row `tail` of its lane
    return n


# boundary pairs for the general classes, rotated over the entries: (upstream, downstream)
LIGHT_PAIRS = [("flow", "normal"), ("stage", "fixed"), ("flow", "poly"), ("fixed", "flow_ds"), ("normal_us", "stage_ds"),
               ("rating_us", "stage_ds"), ("flow", "storage"), ("flow", "blend")]
GENERAL_PAIRS = LIGHT_PAIRS + [("flow", "power")]


def _hyd(Qb, nt, dt, amp=1.0):
    from synth import akbari_shape
    return np.array([akbari_shape(Qb, amp * Qb, 4 * dt, 12 * dt, k * dt) for k in range(nt)])


def prismatic_problem(e, pair, trapezoid, n_steps=4):
    """rectangular / simple trapezoidal prismatic reach with the boundary pair `pair`, started near uniform flow"""
    from synth import normal_depth_rect, normal_depth_trap
    N = _nodes(e)
    rng = np.random.default_rng(1000 + e["index"])
    b = rng.uniform(40, 120); n = rng.uniform(0.025, 0.035); S0 = rng.uniform(2e-4, 5e-4); m = rng.uniform(1, 2.5) if trapezoid else 0.0
    Qb = rng.uniform(1.0, 3.0) * b
    dx = 400.0 if N < 600 else 250.0
    dt = 900.0 if N < 600 else 600.0
    L = (N - 1) * dx
    geo = {k: np.zeros(N) for k in O.GEO_KEYS}
    geo["b_main"][:] = b; geo["m_main"][:] = m
    geo["n_main"][:] = n; geo["n_left"][:] = n; geo["n_right"][:] = n
    geo["z_bed"] = S0 * L * (1 - np.arange(N) / (N - 1))
    hn = normal_depth_trap(b, m, n, S0, Qb) if trapezoid else normal_depth_rect(b, n, S0, Qb)
    nt = n_steps + 1
    us_k, ds_k = pair
    if ds_k == "storage" and N > 130:
        # the reservoir row starts with the reference's level-1 quirk (vol_in forced to 0: the outflow reverses,
        # boundary.py:104-108), a transient Newton only gets through on short reaches
        ds_k = "normal"
    zu = S0 * L
    if us_k == "flow":
        us = O.BC("flow_hydrograph", bed_level=zu, target=_hyd(Qb, nt, dt))
    elif us_k == "stage":
        us = O.BC("stage_hydrograph", bed_level=zu, target=zu + hn + 0.25 * hn * (_hyd(1.0, nt, dt) - 1.0))
    elif us_k == "fixed":
        us = O.BC("fixed_depth", bed_level=zu, initial_depth=hn)
    elif us_k == "normal_us":
        us = O.BC("normal_depth", bed_level=zu, bed_slope=S0)
    elif us_k == "rating_us":        # head-dependent inflow: falls as the upstream stage rises
        us = O.BC("rating_curve", bed_level=zu, rc_type="polynomial", rc=dict(a=0.0, b=-0.2 * Qb, c=Qb + 0.2 * Qb * (zu + hn), shift=0.0))
    else:
        raise KeyError(us_k)
    if ds_k == "normal":
        ds = O.BC("normal_depth", bed_level=0.0, bed_slope=S0)
    elif ds_k == "fixed":
        ds = O.BC("fixed_depth", bed_level=0.0, initial_depth=hn)
    elif ds_k == "poly":
        ds = O.BC("rating_curve", bed_level=0.0, rc_type="polynomial", rc=dict(a=0.15 * Qb / hn ** 2, b=0.85 * Qb / hn, c=0.0, shift=0.0))
    elif ds_k == "power":
        ds = O.BC("rating_curve", bed_level=0.0, rc_type="power", rc=dict(a=Qb / hn ** 1.6, b=1.6, shift=0.0))
    elif ds_k == "flow_ds":          # gate closing: the outflow dips
        ds = O.BC("flow_hydrograph", bed_level=0.0, target=Qb * (1.0 - 0.3 * (_hyd(1.0, nt, dt) - 1.0)))
    elif ds_k == "stage_ds":
        ds = O.BC("stage_hydrograph", bed_level=0.0, target=hn + 0.2 * hn * (_hyd(1.0, nt, dt) - 1.0))
    elif ds_k == "storage":
        ds = O.BC("fixed_depth", bed_level=0.0, initial_depth=hn,
                  storage=dict(area=40.0 * b * L / 50.0, min_stage=0.5 * hn, Y_min=0.0, Y_max=50.0 * hn))
    elif ds_k == "blend":            # two quadratics blended over half a metre above the initial stage
        lo = [0.0, 0.9 * Qb / hn, 0.1 * Qb / hn ** 2]
        ds = O.BC("rating_curve", bed_level=0.0, rc_type="blend",
                  rc=dict(initial_stage=hn, buffer=0.5, low=lo, high=[2 * v for v in lo], dY=1e-3))
    else:
        raise KeyError(ds_k)
    return O.Problem(geo=geo, h0=np.full(N, hn), Q0=np.full(N, Qb), us=us, ds=ds, theta=0.65, dt=dt, dx=dx, nt=nt, tol=1e-6)


def fixture_problem(name, n_steps=None, member=None):
    fx, meta = O.load_fixture(os.path.join(GOLDEN, name + ".npz"))
    p = O.problem_from_fixture(fx, meta, member)
    if n_steps is not None:
        p.nt = min(p.nt, n_steps + 1)
    return p


def case_for(e):
    """(problem, section mode of the batch, n_main override) for a dispatch-table entry"""
    from flowsim_amd import _abi as A
    sec, bck, cap = e["section_mode"], e["boundary_class"], 64 * e["cells_per_thread"] * e["waves_per_reach"]
    if sec in (A.SEC_RECT_UNIFORM, A.SEC_TRAP_UNIFORM):
        trap = sec == A.SEC_TRAP_UNIFORM
        if bck >= 2:
            pair = ("flow", {A.BC_NORMAL_DEPTH: "normal", A.BC_RATING_POWER: "power", A.BC_RATING_BLEND: "blend"}[bck - 2])
        elif bck == 1:
            pair = LIGHT_PAIRS[e["index"] % len(LIGHT_PAIRS)]
        elif bck == 0:
            pair = GENERAL_PAIRS[e["index"] % len(GENERAL_PAIRS)]
        else:
            raise KeyError("no recipe: uniform-geometry kernels of class -1 do not exist")
        return prismatic_problem(e, pair, trap), ("trap_uniform" if trap else "rect_uniform"), None
    if sec == A.SEC_TABLE and e.get("long_reach"):
        # a long prismatic reach described as a table (the trapezoid-family code path of the multi-pass kernel, three passes)
        return prismatic_problem(e, GENERAL_PAIRS[e["index"] % len(GENERAL_PAIRS)], e["index"] % 2 == 0), "table", None
    if sec == A.SEC_TABLE:
        if bck == -1:      # the class exists for the general reservoir row: three reference fixtures, by capacity
            name = ("storage_curve_poly_losses", "storage_curve_power_trap", "storage_curve_closed")[e["index"] % 3]
            return fixture_problem(name, 10), "table", None
        if bck >= 2:
            assert bck - 2 == A.BC_RATING_BLEND and cap >= 120
            if e.get("tail", -1) == 1:      # an even node count: a prismatic trapezoid described as a table, blended rating curve downstream
                return prismatic_problem(e, ("flow", "blend"), True), "table", None
            return fixture_problem("gerd", 6), "table", None
        # class 0: compound sections going over bank where the capacity allows, else the table of a plain trapezoid
        # (fp32 cannot take the central difference of the Roseires gate curve: 1 mm on a stage of 487 m is 30 ulp)
        if cap >= 120 and e["index"] % 2 and e["dtype"] == A.F64:
            return fixture_problem("gerd", 5), "table", None
        return fixture_problem("bc_compound_normal" if e["index"] % 3 else "bc_trap_poly", 8), "table", None
    if sec == A.SEC_IRREGULAR:
        if bck == -1:
            return fixture_problem("irr_storage", 10), "irregular", None
        if bck >= 2:
            assert bck - 2 == A.BC_NORMAL_DEPTH
            return fixture_problem(("irr_single", "irr_mixed")[e["index"] % 2], 8), "irregular", None
        return fixture_problem(("irr_levee", "irr_mixed", "irr_single")[e["index"] % 3], 8), "irregular", None
    raise KeyError(f"no recipe for section mode {sec}")


def oracle_run(p):
    """C oracle where it applies (trapezoid family, closed-form boundaries), the numpy one otherwise"""
    general_storage = p.ds.storage is not None and any(p.ds.storage.get(k) is not None for k in ("curve", "rc", "losses"))
    if "irr_npts" in p.geo or general_storage:
        return O.newton_run(p)
    from oracle import c_oracle as CO
    return CO.run(p)


@pytest.mark.parametrize("e", TABLE, ids=[_id(e) for e in TABLE])
def test_instantiation_against_the_oracle(e, monkeypatch):
    from fixture_batch import batch_from_problems
    from flowsim_amd import _abi as A
    p, mode, override = case_for(e)
    f32 = e["dtype"] == A.F32
    if f32:
        p.tol = 1e-3 if p.N <= 600 else 2e-2      # fp32 cannot resolve ||R|| below ~6e-8 |Q| sqrt(2N) (bench.py uses the same)
    cap = 64 * e["cells_per_thread"] * e["waves_per_reach"] * (64 // e["waves_per_reach"] if (e.get("long_reach") or e.get("team")) else 1)
    assert p.N <= cap, "recipe does not fit the entry"
    ref = oracle_run(p)
    assert ref["status"] == 0
    monkeypatch.setenv("FS_KERNEL_INDEX", str(e["index"]))
    history = bool(e["diag"])
    with batch_from_problems([p], mode=mode, dtype="f32" if f32 else "f64", history=history, n_main_override=override) as b:
        b.step(p.nt - 1)
        assert b.kernel_index() == e["index"]
        assert np.all(b.status() == 0), b.status()
        hyd = b.hydrographs(0, p.nt)[:, :, 0]
        h, Q = b.state()
        its = b.iterations(0, p.nt)[:, 0]
        hist = b.history_arrays(0, p.nt) if history else None
    d, f = ref["depth"], ref["flow"]
    tol = TOL_F32 if f32 else TOL
    assert rel_err(hyd[:, 0], d[:, 0], 1e-3) <= tol and rel_err(hyd[:, 2], d[:, -1], 1e-3) <= tol
    assert rel_err(hyd[:, 1], f[:, 0], 1.0) <= tol and rel_err(hyd[:, 3], f[:, -1], 1.0) <= tol
    assert rel_err(h[0], d[-1], 1e-3) <= tol and rel_err(Q[0], f[-1], 1.0) <= tol
    if hist is not None:
        assert rel_err(hist[0][:, 0], d, 1e-3) <= tol and rel_err(hist[1][:, 0], f, 1.0) <= tol
    if not f32:
        assert np.array_equal(its, ref["iters"])


# ---- a SECOND recipe for the entries whose node evaluation is the intricate one (tables, polylines) ----
# Round 3's (8, 1) polyline kernel was right or wrong depending on code the case never executed: one case per entry is a thin net
# for a miscompile.  The second case comes from the reference's own random sweep (tests/golden/random_sweep.npz: the reference
# built these channels and ran them; compared here with ITS histories, not an oracle's): other sections, other boundary kinds,
# other node counts than the first recipe's.
SWEEP = {i: (fx, m) for i, fx, m in O.sweep_cases(os.path.join(GOLDEN, "random_sweep.npz"))}
SECOND_POLY = {-1: [58, 50, 52, 55], 0: [50, 52, 53, 58, 55, 48], 2: [54, 57]}           # by boundary class (2: flow upstream, normal depth downstream)
SECOND_TABLE = {-1: [61, 62, 65, 70, 68, 60], 0: [78, 79, 8, 10, 24, 3, 75, 21, 12]}      # general storages; compound sections, bends


def second_case_for(e):
    """(problem, mode, reference arrays or None) - None: compare with the oracle"""
    from flowsim_amd import _abi as A
    sec, bck = e["section_mode"], e["boundary_class"]
    cap = 64 * e["cells_per_thread"] * e["waves_per_reach"] * (64 // e["waves_per_reach"] if e.get("long_reach") else 1)
    if sec == A.SEC_TABLE and e.get("long_reach"):
        pair = GENERAL_PAIRS[(e["index"] + 4) % len(GENERAL_PAIRS)]
        return prismatic_problem(e, pair, e["index"] % 2 == 1, n_steps=3), "table", None
    if sec == A.SEC_TABLE and bck >= 2:
        if e.get("tail", -1) == 1:
            return prismatic_problem(dict(e, index=e["index"] + 2), ("flow", "blend"), False, n_steps=5), "table", None
        return fixture_problem("gerd_ensemble", 5, member=e["index"] % 8), "table", None
    pool = (SECOND_TABLE if sec == A.SEC_TABLE else SECOND_POLY)[min(bck, 2)]
    fits = [i for i in pool if SWEEP[i][1]["N"] <= cap]
    if e["dtype"] == A.F32:       # fp32 follows a flood wave, not one that stalls and reverses (a stage-driven case of the sweep does): those stay with fp64
        fits = [i for i in fits if float(np.min(SWEEP[i][0]["flow"])) > 0.3 * SWEEP[i][1]["Qb"]] or fits
    i = fits[e["index"] % len(fits)]
    fx, m = SWEEP[i]
    return O.problem_from_fixture(fx, m), ("table" if sec == A.SEC_TABLE else "irregular"), (fx, m)


SECOND = [e for e in TABLE if e["section_mode"] in (2, 3)]


@pytest.mark.parametrize("e", SECOND, ids=[_id(e) for e in SECOND])
def test_table_and_polyline_instantiations_on_a_second_case(e, monkeypatch):
    from fixture_batch import batch_from_problems
    from flowsim_amd import _abi as A
    p, mode, ref_fx = second_case_for(e)
    f32 = e["dtype"] == A.F32
    if f32:
        p.tol = max(p.tol, 1e-3)
    if ref_fx is None:
        ref = oracle_run(p)
        assert ref["status"] == 0
        d, f, ref_its = ref["depth"], ref["flow"], ref["iters"]
        hfloor, qfloor = 1e-3, 1.0
    else:
        fx, m = ref_fx
        d, f, ref_its = fx["depth"], fx["flow"], fx["iters"]
        hfloor, qfloor = 1e-3 * m["h_n"], 1e-3 * m["Qb"]
        if f32:                    # fp32 on the reference's random channels: the deviation measured against the case's scales (base depth, base
            hfloor, qfloor = m["h_n"], m["Qb"]          # flow), 2e-2 of them - a miscompiled kernel misses by orders of magnitude or faults
    monkeypatch.setenv("FS_KERNEL_INDEX", str(e["index"]))
    history = bool(e["diag"])
    with batch_from_problems([p], mode=mode, dtype="f32" if f32 else "f64", history=history) as b:
        b.step(p.nt - 1)
        assert b.kernel_index() == e["index"]
        assert np.all(b.status() == 0), b.status()
        hyd = b.hydrographs(0, p.nt)[:, :, 0]
        its = b.iterations(0, p.nt)[:, 0]
        hist = b.history_arrays(0, p.nt) if history else None
    tol = (2e-2 if ref_fx is not None else TOL_F32) if f32 else TOL      # (fp32 on a reference sweep case: the same flood wave, see above)
    assert rel_err(hyd[:, 0], d[:, 0], hfloor) <= tol and rel_err(hyd[:, 2], d[:, -1], hfloor) <= tol
    assert rel_err(hyd[:, 1], f[:, 0], qfloor) <= tol and rel_err(hyd[:, 3], f[:, -1], qfloor) <= tol
    if hist is not None:
        assert rel_err(hist[0][:, 0], d, hfloor) <= tol and rel_err(hist[1][:, 0], f, qfloor) <= tol
    if not f32:
        assert np.array_equal(its, ref_its)


def test_the_second_recipes_differ_from_the_first():
    assert len(SECOND) >= 30
    for e in SECOND:
        p1, _, _ = case_for(e)
        p2, _, _ = second_case_for(e)
        assert (p1.N, p1.dt, p1.ds.kind, p1.us.kind) != (p2.N, p2.dt, p2.ds.kind, p2.us.kind) or not np.array_equal(p1.geo["n_main"], p2.geo["n_main"]), _id(e)


def test_the_table_is_what_this_file_expects():
    """every entry has a recipe, the recipes between them reach every boundary kind on both ends, and the forced index
    is refused when it does not fit"""
    from flowsim_amd import _abi as A
    from flowsim_amd import PreissmannBatch
    assert len(TABLE) == A.lib().fs_kernel_table_size() > 0
    seen = set()
    for e in TABLE:
        p, mode, _ = case_for(e)
        seen.add(("us", p.us.kind)); seen.add(("ds", p.ds.kind, p.ds.rc_type, p.ds.storage is not None))
    for kind in ("flow_hydrograph", "stage_hydrograph", "fixed_depth", "normal_depth", "rating_curve"):
        assert ("us", kind) in seen, kind
    for key in (("ds", "flow_hydrograph", None, False), ("ds", "stage_hydrograph", None, False), ("ds", "fixed_depth", None, False),
                ("ds", "fixed_depth", None, True), ("ds", "normal_depth", None, False), ("ds", "rating_curve", "power", False),
                ("ds", "rating_curve", "polynomial", False), ("ds", "rating_curve", "blend", False)):
        assert key in seen, key
    os.environ["FS_KERNEL_INDEX"] = "0"            # entry 0 is an fp64 rectangular (2,1) kernel: 128 nodes at most
    try:
        from synth import rect_problem
        from fixture_batch import batch_from_problems
        with batch_from_problems([rect_problem(200, seed=1, n_steps=2)], mode="rect_uniform") as b:
            with pytest.raises(A.FlowsimError, match="FS_KERNEL_INDEX"):
                b.step(1)
    finally:
        del os.environ["FS_KERNEL_INDEX"]
