"""The instantiations bench.py launches, at the sizes it launches them, against fixtures produced by the
reference itself (oracle/gen_golden.py: c3_4096, c5_512, gerd_full).

bench.py creates its batches without FS_FLAG_HISTORY / FS_FLAG_TRACE, so fs_batch_step picks the kernels compiled
without those stores (DIAG = false) and with the downstream kind fixed at compile time (boundary class 2 + kind).
These tests build the batches the same way - same parameter draws (flowsim_amd.synthetic), no history - assert that
exactly that dispatch-table entry ran, and compare what such a batch keeps (the boundary hydrograph rows of every
level, the accepted state of the last level, the Newton start vector of the next one, the iteration counts) with the
reference's depth / flow history.  Tolerance 1e-8 relative, identical Newton counts."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import preissmann_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-8


def rel_err(got, want, floor):
    return float(np.max(np.abs(got - want) / np.maximum(np.abs(want), floor)))


def check_rows(b, depth, flow, iters, j=0):
    """hydrograph rows [levels, 4, B], final state and counts of reach j against depth / flow [nt, N]"""
    nt = depth.shape[0]
    hyd = b.hydrographs(0, nt)
    assert rel_err(hyd[:, 0, j], depth[:, 0], 1e-3) <= TOL and rel_err(hyd[:, 2, j], depth[:, -1], 1e-3) <= TOL
    assert rel_err(hyd[:, 1, j], flow[:, 0], 1.0) <= TOL and rel_err(hyd[:, 3, j], flow[:, -1], 1.0) <= TOL
    h, Q = b.state()
    assert rel_err(h[j], depth[-1], 1e-3) <= TOL and rel_err(Q[j], flow[-1], 1.0) <= TOL
    assert np.array_equal(b.iterations(0, nt)[:, j], iters)


def entry_of(b):
    from flowsim_amd import _abi as A
    return A.kernel_table()[b.kernel_index()]


def test_c3_flagship_kernel_at_4096_nodes_against_the_reference():
    """BASELINE configs[2] shape: rectangular, 4096 nodes, flow hydrograph -> normal depth, fp64, no history: the
    <double, RECT_UNIFORM, 16, 4, full, normal depth, DIAG = false> instantiation of BENCH_r01."""
    from flowsim_amd import BoundarySpec, PreissmannBatch
    from flowsim_amd import _abi as A
    from flowsim_amd.synthetic import c3_reach_parameters, inflow_table, normal_depth_rect
    fx, meta = O.load_fixture(os.path.join(GOLDEN, "c3_4096.npz"))
    B, N, nt = meta["B"], meta["N"], meta["nt"]
    assert (N, meta["dt"], meta["dx"], meta["theta"], meta["tolerance"]) == (4096, 600.0, 250.0, 0.6, 1e-6)   # bench.py's C3
    b_, n_, S0, Qb = c3_reach_parameters(0, B)                      # bench.py's own draws ...
    assert np.allclose(np.column_stack([b_, n_, S0, Qb]), fx["params"], rtol=1e-15, atol=0)    # ... are the reference run's
    hn = normal_depth_rect(b_, n_, S0, Qb)
    assert rel_err(hn, fx["initial_conditions"][:, 0, 0], 1e-3) <= 1e-10
    L = (N - 1) * meta["dx"]
    for uniform_ic in (True, False):
        with PreissmannBatch(B, N, nt, section_mode="rect_uniform", monitor=False) as b:      # history=False, trace=False, no monitor: as bench.py
            b.set_scheme(meta["theta"], meta["dt"], meta["dx"], meta["tolerance"], 100)
            b.set_geometry_uniform(b_, n_, S0 * L, np.zeros(B))
            b.set_boundary(A.UPSTREAM, BoundarySpec(A.BC_FLOW_HYDROGRAPH, {}, inflow_table(Qb, nt, meta["dt"])))
            b.set_boundary(A.DOWNSTREAM, BoundarySpec(A.BC_NORMAL_DEPTH, dict(bed_slope=S0, bed_level=np.zeros(B))))
            if uniform_ic:
                b.set_state_uniform(hn, Qb)                         # bench.py's path
            else:
                b.set_state(fx["initial_conditions"][:, :, 0], fx["initial_conditions"][:, :, 1])
            b.step(nt - 1)
            assert np.all(b.status() == 0)
            e = entry_of(b)
            assert (e["cells_per_thread"], e["waves_per_reach"], e["full"], e["diag"]) == (16, 4, 1, 0)
            assert e["boundary_class"] == 2 + A.BC_NORMAL_DEPTH and e["dtype"] == A.F64
            for j in range(B):
                check_rows(b, fx["depth"][j], fx["flow"][j], fx["iters"][j], j)
            gh, gQ = b.guess()
            fin = fx["final_unknowns"]
            assert rel_err(gh, fin[:, 0::2], 1e-3) <= 1e-7 and rel_err(gQ, fin[:, 1::2], 1.0) <= 1e-7


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_c5_kernel_at_512_nodes_against_the_reference(dtype):
    """BASELINE configs[4] shape: simple trapezoid, 512 nodes, power rating curve downstream, no history: the
    <R, TRAP_UNIFORM, 8, 1, full, rating power, DIAG = false> instantiations.  fp32 is the throughput mode:
    tolerance 1e-3, hydrographs within 5e-4 of the fp64 reference."""
    from flowsim_amd import BoundarySpec, PreissmannBatch
    from flowsim_amd import _abi as A
    from flowsim_amd.synthetic import c5_reach_parameters, inflow_table, normal_depth_trap
    fx, meta = O.load_fixture(os.path.join(GOLDEN, "c5_512.npz"))
    B, N, nt = meta["B"], meta["N"], meta["nt"]
    assert (N, meta["dt"], meta["dx"], meta["theta"]) == (512, 1800.0, 500.0, 0.6)
    b_, m_, n_, S0, Qb = c5_reach_parameters(0, B)
    assert np.allclose(np.column_stack([b_, m_, n_, S0, Qb]), fx["params"][:, :5], rtol=1e-15, atol=0)
    hn = normal_depth_trap(b_, m_, n_, S0, Qb)
    assert rel_err(hn, fx["params"][:, 5], 1e-3) <= 1e-10
    L = (N - 1) * meta["dx"]
    with PreissmannBatch(B, N, nt, dtype=dtype, section_mode="trap_uniform", monitor=False) as b:
        b.set_scheme(meta["theta"], meta["dt"], meta["dx"], 1e-6 if dtype == "f64" else 1e-3, 100)
        b.set_geometry_uniform(b_, n_, S0 * L, np.zeros(B), side_slope=m_)
        b.set_boundary(A.UPSTREAM, BoundarySpec(A.BC_FLOW_HYDROGRAPH, {}, inflow_table(Qb, nt, meta["dt"])))
        b.set_boundary(A.DOWNSTREAM, BoundarySpec(A.BC_RATING_POWER, dict(a=Qb / hn ** 1.6, b=np.full(B, 1.6),
                                                                        stage_shift=np.zeros(B), bed_level=np.zeros(B))))
        b.set_state_uniform(hn, Qb)
        b.step(nt - 1)
        assert np.all(b.status() == 0)
        e = entry_of(b)
        assert (e["cells_per_thread"], e["waves_per_reach"], e["full"]) == (8, 1, 1)
        assert e["boundary_class"] == 2 + A.BC_RATING_POWER and e["diag"] == 0
        if dtype == "f64":
            for j in range(B):
                check_rows(b, fx["depth"][j], fx["flow"][j], fx["iters"][j], j)
        else:
            hyd = b.hydrographs(0, nt)
            h, Q = b.state()
            for j in range(B):
                assert rel_err(hyd[:, 2, j], fx["depth"][j][:, -1], 1e-3) <= 5e-4
                assert rel_err(hyd[:, 3, j], fx["flow"][j][:, -1], 1.0) <= 5e-4
                assert rel_err(h[j], fx["depth"][j][-1], 1e-3) <= 5e-4 and rel_err(Q[j], fx["flow"][j][-1], 1.0) <= 5e-4


@pytest.mark.parametrize("chunks", [1, 4])
def test_gerd_roseires_over_all_384_levels_against_the_reference(chunks):
    """BASELINE configs[3] member: cases/gerd_roseires over its whole simulation (settings.py:3-8) in the ensemble
    kernel <double, TABLE, 2, 1, gate curve, DIAG = false>; in one launch and in four (chunked stepping)."""
    from fixture_batch import batch_from_problems
    from flowsim_amd import _abi as A
    fx, meta = O.load_fixture(os.path.join(GOLDEN, "gerd_full.npz"))
    assert meta["nt"] == 385 and meta["N"] == 121
    p = O.problem_from_fixture(fx, meta)
    with batch_from_problems([p], mode="table", history=False) as b:
        left = p.nt - 1
        for c in range(chunks):
            n = left // (chunks - c)
            b.step(n)
            left -= n
        assert np.all(b.status() == 0)
        e = entry_of(b)
        assert (e["cells_per_thread"], e["waves_per_reach"], e["diag"]) == (2, 1, 0)
        assert e["boundary_class"] == 2 + A.BC_RATING_BLEND and e["tail"] == 0       # 121 nodes: the boundary row is local row 0 of lane 60
        check_rows(b, fx["depth"], fx["flow"], fx["iters"])


def test_the_tail_only_form_gives_the_bits_of_the_ragged_kernel(monkeypatch):
    """<double, TABLE, 2, 1, gate curve, DIAG = false, TAIL = 0> (round 4: every row a cell but the boundary row's lane) against the
    general ragged instantiation of the same shape: the rows behind the boundary row are phantom cells there and identity rows here,
    and neither can reach a real unknown (the boundary row's super-diagonal is zero) - hydrographs, state, start vector and counts are
    bit-identical over 48 levels of cases/gerd_roseires and over an 8-member Manning-n ensemble."""
    from fixture_batch import batch_from_problems
    from flowsim_amd import _abi as A
    table = A.kernel_table()
    plain = [e["index"] for e in table if e["section_mode"] == A.SEC_TABLE and e["dtype"] == A.F64 and (e["cells_per_thread"], e["waves_per_reach"]) == (2, 1)
             and e["boundary_class"] == 2 + A.BC_RATING_BLEND and e["diag"] == 0 and e["tail"] == -1]
    assert len(plain) == 1
    for name, B in (("gerd", 1), ("gerd_ensemble", 8)):
        fx, meta = O.load_fixture(os.path.join(GOLDEN, name + ".npz"))
        probs = [O.problem_from_fixture(fx, meta, m) for m in (range(B) if meta.get("B") else [None])]
        override = [float(p.geo["n_main"][0]) for p in probs] if B > 1 else None
        out = {}
        for forced in (None, plain[0]):
            if forced is not None:
                monkeypatch.setenv("FS_KERNEL_INDEX", str(forced))
            with batch_from_problems(probs, mode="table", history=False, n_main_override=override) as b:
                b.step(probs[0].nt - 1)
                assert np.all(b.status() == 0)
                e = table[b.kernel_index()]
                assert e["tail"] == (0 if forced is None else -1)
                out[forced] = (b.hydrographs(0, probs[0].nt), b.state(), b.guess(), b.iterations(0, probs[0].nt))
            monkeypatch.delenv("FS_KERNEL_INDEX", raising=False)
        a, c = out[None], out[plain[0]]
        assert np.array_equal(a[0], c[0]) and np.array_equal(a[3], c[3])
        assert all(np.array_equal(x, y) for x, y in zip(a[1] + a[2], c[1] + c[2]))


def test_dispatch_prefers_the_most_specific_instantiation():
    """pick_kernel's ordering (fs_abi.hip): smallest capacity, then fewest waves per reach, then the most specific
    variant (boundary pair fixed > closed-form rows > general; no-history build when the batch keeps none)."""
    from flowsim_amd import BoundarySpec, PreissmannBatch
    from flowsim_amd import _abi as A
    from synth import rect_problem
    from fixture_batch import batch_from_problems
    want = {                                   # N -> (M, W, full, class with / without history)
        4096: (16, 4, 1, 2 + A.BC_NORMAL_DEPTH), 4000: (16, 4, 0, 2 + A.BC_NORMAL_DEPTH), 2048: (16, 2, 1, 2 + A.BC_NORMAL_DEPTH),
        1024: (16, 1, 1, 2 + A.BC_NORMAL_DEPTH), 512: (8, 1, 1, 2 + A.BC_NORMAL_DEPTH), 300: (8, 1, 0, 2 + A.BC_NORMAL_DEPTH),
        513: (16, 1, 0, 1), 200: (4, 1, 0, 1), 100: (2, 1, 0, 1), 40: (2, 1, 0, 1), 2000: (16, 2, 0, 1),
    }
    for N, (M, W, full, bck) in want.items():
        p = rect_problem(N, seed=3, n_steps=2)
        for history in (False, True):
            with batch_from_problems([p], mode="rect_uniform", history=history) as b:
                b.step(1)
                e = entry_of(b)
                assert (e["cells_per_thread"], e["waves_per_reach"], e["full"], e["boundary_class"]) == (M, W, full, bck), (N, history, e)
                nodiag_exists = any(t["diag"] == 0 and all(t[k] == e[k] for k in ("dtype", "section_mode", "cells_per_thread",
                                    "waves_per_reach", "full", "boundary_class")) for t in A.kernel_table())
                assert e["diag"] == (1 if history or not nodiag_exists else 0), (N, history, e)


@pytest.mark.parametrize("spec", [("c4", "f64", 96, 121, 3, 1), ("c5", "f32", 128, 512, 3, 1), ("c5", "f64", 128, 512, 3, 1),
                                  ("irr", "f64", 64, 128, 3, 1), ("long", "f64", 4, 16384, 2, 1)], ids=lambda s: f"{s[0]}-{s[1]}")
def test_extra_workloads_of_the_bench_line(spec):
    """bench.py's "workloads" entries (C4, C5 in both precisions, the polyline ensemble, long reaches), at a handful of reaches:
    every entry launches the instantiation the full-size run launches (the dispatch does not depend on the reach count), converges
    and carries the contract's fields; the full-size entries are what the driver's BENCH line holds."""
    import importlib.util
    from conftest import ROOT
    sp = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(sp); sp.loader.exec_module(bench)
    assert [s[:2] for s in bench.EXTRA_WORKLOADS] == [("c4", "f64"), ("c5", "f32"), ("c5", "f64"), ("irr", "f64"), ("long", "f64")]
    full = {s[:2]: s for s in bench.EXTRA_WORKLOADS}[spec[:2]]
    assert full[3] == spec[3]                                   # same node count as the driver-observed entry
    w = bench.run_extra_workload(spec, 0)
    assert w["all_converged"] and w["value"] > 0 and w["kernel_ms"] > 0 and w["reaches"] == spec[2] and w["nodes"] == spec[3]
    r = w["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    real = 4 if spec[1] == "f32" else 8
    assert abs(r["achieved"] - spec[2] * spec[4] * (4 * spec[3] * real + 40) / (w["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    k = w["kernel"]
    want = {"c4": (2, 1, 0), "c5": (8, 1, 0), "irr": (2, 1, 0), "long": (16, 4, 1)}[spec[0]]          # long: a team of four workgroups per reach
    assert (k["cells_per_thread"], k["waves_per_reach"], k["team"]) == want and k["long_reach"] == 0 and (k["diag"] == 0 or k["team"])
