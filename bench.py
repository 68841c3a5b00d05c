#!/usr/bin/env python3
"""bench.py - reach-timesteps/sec of the batched Preissmann Newton step on MI355X.

    python bench.py --gpus 1 --steps 32 --warmup 4
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], SURVEY.md 8d "C3"): per GPU 65 536 synthetic rectangular
reaches x 4 096 nodes, fp64, constant Manning n per reach, upstream flow hydrograph (akbari
shape), downstream normal depth, steady-state initial condition, theta 0.6, dt 600 s, dx 250 m,
tolerance 1e-6.  One "step" = one time level (full Newton loop) of every reach in the batch.
Weak scaling (default): every rank owns its own block of --reaches reaches (global reach index seeds
the draws).  Strong scaling (--total-reaches T): T reaches in all, split into contiguous blocks over
the ranks (flowsim_amd.shard.split_reaches; north_star quotes 65 536 reaches in total at 8 GPUs).  The
only collective is one RCCL gather of the boundary hydrographs to rank 0, inside the timed region.  Inputs
are resident in HBM before the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     - algorithmic HBM bytes of the step kernel / its HIP-event time, vs 8 TB/s
  cpu_baseline - the numpy/scipy oracle port (oracle/preissmann_oracle.py) on 1 host core on a
                 bounded sample of the same workload (reported, not the optimisation target)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# multi-process GPU work on this pool needs dmabuf IPC (RCCL fails with "hipIpcGetMemHandle: invalid argument" otherwise); the
# boxes export it already - kept here so that a bare `torch.distributed.run bench.py` from a clean shell works as well
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "flow-sim_amd"))
sys.path.insert(0, ROOT)

FP64_VECTOR_PEAK_TFLOPS = 78.6    # vector fp64: 256 CUs x 4 SIMDs x 16 lanes x 2 flop (FMA) x 2.4 GHz
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
FP64_VALU_PEAK_TFLOPS = 78.6   # vector fp64, the unit this kernel actually runs on
PROFILE_ROUND = "round4"       # profiles/<round>/<workload>_<dtype>.json: the rocprofv3 counter passes `traffic` and `roofline_compute` come from

# The other BASELINE configurations and the two shapes SURVEY 8(f) / DESIGN section 4.5 add, run after the headline (C3) at --gpus 1
# and reported as "workloads": [...] in the same JSON line: (workload, dtype, reaches, nodes, timed levels, warm-up levels).
# C4 = BASELINE configs[3] per GPU (262 144 members / 8), C5 = configs[4] per GPU (1 048 576 / 8) in fp32 and fp64, irr = polyline
# ensemble, long = 8 192 rectangular reaches of 16 384 nodes on the multi-pass kernel.
EXTRA_WORKLOADS = [("c4", "f64", 32768, 121, 16, 2), ("c5", "f32", 131072, 512, 32, 4), ("c5", "f64", 131072, 512, 32, 4),
                   ("irr", "f64", 8192, 128, 16, 2), ("long", "f64", 8192, 16384, 8, 2)]


from flowsim_amd.synthetic import (c3_reach_parameters, c5_reach_parameters, inflow_table,  # noqa: E402
                                   normal_depth_rect, normal_depth_trap)
from flowsim_amd.shard import gather_hydrographs, gather_hydrographs_split, reach_block, split_reaches  # noqa: E402


def library_sha256():
    import hashlib
    from flowsim_amd import _abi
    return hashlib.sha256(open(os.path.normpath(os.environ.get("FS_LIB", _abi.LIB_PATH)), "rb").read()).hexdigest()


def load_profile(workload, dtype, N, entry):
    """HBM traffic / flop counts of this workload's step kernel from the committed rocprofv3 PMC passes
    (profiles/round4/<workload>_<dtype>.json, tools/refresh_profiles.py) - only if they were taken on the very library
    that is loaded now and on the same instantiation; a stale file gives None (traffic: null) rather than a wrong number."""
    path = os.path.join(ROOT, "profiles", PROFILE_ROUND, f"{workload}_{dtype}.json")
    if not os.path.exists(path):
        return None
    p = json.load(open(path))
    same_kernel = all(p["kernel"].get(k) == entry[k] for k in ("cells_per_thread", "waves_per_reach", "full", "boundary_class", "diag")) \
        and p["kernel"].get("table_index", entry["index"]) == entry["index"]
    if p.get("library_sha256") != library_sha256() or p.get("nodes") != N or not same_kernel:
        return None
    p["file"] = os.path.relpath(path, ROOT)
    return p


def traffic_of_launch(prof, B, K):
    """HBM bytes of ONE launch of K levels over B reaches.  The step kernel reads the state once per launch and writes it
    back once per launch, whatever K is; only the boundary targets, hydrograph rows and Newton counts
    scale with K.  The profile therefore keeps the two parts apart (two rocprofv3 runs at different K give both,
    tools/refresh_profiles.py --second): traffic = fixed * B + per_level * B * K.  A profile that only has a per-launch
    figure is used for exactly its own K and nothing else."""
    if prof is None:
        return None
    if "hbm_bytes_per_reach_fixed" in prof:
        return (prof["hbm_bytes_per_reach_fixed"] + prof["hbm_bytes_per_reach_per_level"] * K) * float(B)
    if prof.get("levels_in_launch") == K and "hbm_bytes_per_reach_timestep" in prof:
        return prof["hbm_bytes_per_reach_timestep"] * float(B) * K
    return None


def cpu_baseline(N, dt, dx, theta, tol, budget_s=9.0, compiled=False):
    """Oracle port on one host core, same workload definition, bounded sample.  compiled=False: the
    numpy + scipy.sparse.linalg.spsolve port (the reference's own solver call); compiled=True: the
    plain-C restatement with its banded LU (oracle/preissmann_oracle.c)."""
    try:
        import psutil
        psutil.Process().cpu_affinity([psutil.Process().cpu_affinity()[0]])
    except Exception:
        pass
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    from oracle import preissmann_oracle as O
    if compiled:
        from oracle import c_oracle as CO
    b, n, S0, Qb = c3_reach_parameters(0, 1024)
    hn = normal_depth_rect(b, n, S0, Qb)
    steps = 4
    tgt = inflow_table(Qb, steps + 1, dt)
    done = 0
    t0 = time.perf_counter()
    for r in range(1024):
        geo = {k: np.zeros(N) for k in O.GEO_KEYS}
        geo["b_main"][:] = b[r]; geo["n_main"][:] = n[r]; geo["n_left"][:] = n[r]; geo["n_right"][:] = n[r]
        L = (N - 1) * dx
        geo["z_bed"] = S0[r] * L * (1 - np.arange(N) / (N - 1))
        p = O.Problem(geo=geo, h0=np.full(N, hn[r]), Q0=np.full(N, Qb[r]),
                      us=O.BC("flow_hydrograph", bed_level=S0[r] * L, target=tgt[:, r].copy()),
                      ds=O.BC("normal_depth", bed_level=0.0, bed_slope=float(S0[r])),
                      theta=theta, dt=dt, dx=dx, nt=steps + 1, tol=tol)
        (CO.run if compiled else O.newton_run)(p)
        done += steps
        if time.perf_counter() - t0 > budget_s:
            break
    el = time.perf_counter() - t0
    return {"value": done / el, "unit": "reach-timesteps/s", "cores": 1, "kind": "port",
            "sample": f"{done // steps} reaches x {N} nodes x {steps} steps, "
                      + ("C oracle (gcc -O2, banded LU)" if compiled else "numpy+scipy.spsolve oracle") + f", {el:.1f} s"}


def cpu_baseline_all_cores(N, dt, dx, theta, tol, cores, budget_s=6.0):
    """The compiled C restatement over all host cores the process may use (threads over reaches; the
    ctypes call releases the GIL).  Same workload definition, bounded sample."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import preissmann_oracle as O
    from oracle import c_oracle as CO
    try:
        import psutil
        psutil.Process().cpu_affinity(sorted(cores))          # undo the single-core pin of the legs above
    except Exception:
        pass
    nthreads = min(len(cores), 64)      # bounded: the sample is set up in Python before the clock starts
    per_thread, steps = 48, 4
    R = nthreads * per_thread
    b, n, S0, Qb = c3_reach_parameters(0, R)
    hn = normal_depth_rect(b, n, S0, Qb)
    tgt = inflow_table(Qb, steps + 1, dt)
    L = (N - 1) * dx
    ramp = 1 - np.arange(N) / (N - 1)
    CO.lib()
    problems = []                      # set up outside the timed region (numpy allocations hold the GIL)
    for r in range(R):
        geo = {k: np.zeros(N) for k in O.GEO_KEYS}
        geo["b_main"][:] = b[r]; geo["n_main"][:] = n[r]; geo["n_left"][:] = n[r]; geo["n_right"][:] = n[r]
        geo["z_bed"] = S0[r] * L * ramp
        problems.append(O.Problem(geo=geo, h0=np.full(N, hn[r]), Q0=np.full(N, Qb[r]),
                                  us=O.BC("flow_hydrograph", bed_level=S0[r] * L, target=tgt[:, r].copy()),
                                  ds=O.BC("normal_depth", bed_level=0.0, bed_slope=float(S0[r])),
                                  theta=theta, dt=dt, dx=dx, nt=steps + 1, tol=tol))
    t0 = time.perf_counter()

    def work(tid):
        done = 0
        for r in range(tid, R, nthreads):
            CO.run(problems[r])
            done += steps
            if time.perf_counter() - t0 > budget_s:
                break
        return done
    with ThreadPoolExecutor(max_workers=nthreads) as ex:
        done = sum(ex.map(work, range(nthreads)))
    el = time.perf_counter() - t0
    return {"value": done / el, "unit": "reach-timesteps/s", "cores": nthreads, "kind": "port",
            "sample": f"{done // steps} reaches x {N} nodes x {steps} steps, C oracle on {nthreads} threads, {el:.1f} s"}


def build_batch(workload, dtype, B, N, levels, first, local, spatial_step=None, per_reach_geometry=False):
    """One workload resident in HBM, ready to step: (batch, N, theta, dt, dx, tol, description).  `first`: global index of this
    rank's first reach (seeds the parameter draws).  "long" is the C3 channel family at a node count beyond one lane grid."""
    import numpy as np
    from flowsim_amd import BoundarySpec, PreissmannBatch
    from flowsim_amd import _abi as A
    if workload == "long":
        workload = "c3"
    if workload == "c3":
        # fp32 cannot resolve ||R|| below ~6e-8 |Q| sqrt(2N) (1e-2 for the largest of these reaches): its tolerance follows
        theta, dt, dx, tol = 0.6, 600.0, 250.0, (1e-6 if dtype == "f64" else 2e-2)
        b_, n_, S0, Qb = c3_reach_parameters(first, B)
        hn = normal_depth_rect(b_, n_, S0, Qb)
        L = (N - 1) * dx
        batch = PreissmannBatch(B, N, levels, dtype=dtype, section_mode="rect_uniform", device=local, monitor=False)
        batch.set_scheme(theta, dt, dx, tol, 100)
        batch.set_geometry_uniform(b_, n_, S0 * L, np.zeros(B))
        batch.set_boundary(A.DOWNSTREAM, BoundarySpec(A.BC_NORMAL_DEPTH, dict(bed_slope=S0, bed_level=np.zeros(B))))
        desc = ("C3: %d synthetic rectangular reaches x %d nodes per GPU, constant Manning n, flow-hydrograph upstream, "
                "normal-depth downstream, theta 0.6, dt 600 s, dx 250 m, tol %g" % (B, N, tol))
    elif workload == "c4":
        # BASELINE configs[3] / SURVEY 8d C4: gerd_roseires geometry shared by all members, n_main ~ U(0.02, 0.06)
        from cases.gerd_roseires.model import build as build_gerd
        from flowsim_amd.ensemble import gvf_profiles
        from flowsim_amd.hydromodel.preissmann import boundary_to_spec
        extra = {} if spatial_step is None else {"spatial_step": spatial_step}
        solver, _ = build_gerd(inflow_hyd_func=None, sim_duration=(levels - 1) * 3600, **extra)
        ch = solver.channel
        N = solver.number_of_nodes
        rng = np.random.default_rng(20260215)
        n_members = (0.020 + 0.040 * rng.random(first + B))[first:]
        theta, dt, dx, tol = solver.theta, float(solver.time_step), solver.spatial_step, 1e-6
        batch = PreissmannBatch(B, N, levels, dtype=dtype, section_mode="table", device=local, monitor=False)
        batch.set_scheme(theta, dt, dx, tol, 100)
        if per_reach_geometry:
            batch.set_geometry_table({k: np.broadcast_to(np.asarray(ch.node_geometry[k], dtype=np.float64), (B, N)) for k in A.GEO_ROWS},
                                     n_main_override=n_members)
        else:
            batch.set_geometry_table(ch.node_geometry, n_main_override=n_members)
        batch.set_boundary(A.UPSTREAM, boundary_to_spec(ch.upstream_boundary, levels, dt))
        batch.set_boundary(A.DOWNSTREAM, boundary_to_spec(ch.downstream_boundary, levels, dt))
        ic = gvf_profiles(ch, n_members)
        batch.set_state(ic[:, :, 0], ic[:, :, 1])
        Qb = None
        desc = ("C4: cases/gerd_roseires (%d nodes, compound sections + curvature, Roseires gate curve), %d-member "
                "Manning-n ensemble per GPU, theta 0.6, dt 3600 s, tol 1e-6" % (N, B))
    elif workload == "irr":
        # polyline sections (8 -> 15 stations after interpolation, berm on the right bank),
        # composite roughness over three strips, one channel shared by a Manning-n ensemble
        from flowsim_amd.hydromodel import Boundary, Channel, Hydrograph, IrregularSection, PreissmannSolver
        from flowsim_amd.hydromodel.preissmann import boundary_to_spec
        Lc, S0c = 63500.0, 3e-4                # 127 cells of 500 m: 128 nodes, the capacity of the two-rows-per-lane kernel
        xa = np.array([0, 10, 14, 30, 34, 60, 66, 80.0]); za = np.array([8, 3.0, 0.4, 0.0, 0.6, 2.5, 2.8, 8.0])
        xb = np.array([0, 12, 18, 33, 41, 58, 70, 90.0]); zb = np.array([7.5, 2.6, 0.3, 0.0, 0.5, 2.0, 2.6, 7.5])
        secs = []
        for f, xx, zz in ((1.0, xa, za), (0.5, xb, zb), (0.0, xa * 1.1, za * 0.95)):
            s_ = IrregularSection(x=xx, z=S0c * Lc * f + zz, n=0.03, bed_slope=S0c)
            s_.set_roughness_para((0.05, 0.03, 0.06, xx[2], xx[5]))
            secs.append(s_)
        Q0 = 45.0
        hyd = Hydrograph(table=np.column_stack([np.arange(levels + 1) * 300.0,
                                                Q0 * (1.0 + 2.0 * np.sin(np.pi * np.arange(levels + 1) / max(levels, 2)) ** 2)]))
        us = Boundary(condition='flow_hydrograph', bed_level=S0c * Lc, chainage=0, hydrograph=hyd, initial_depth=1.9)
        ds = Boundary(condition='normal_depth', bed_level=0.0, chainage=Lc, initial_depth=1.9)
        ch = Channel(initial_flow=Q0, upstream_boundary=us, downstream_boundary=ds, interpolation_method='steady-state')
        ch.set_cross_sections([0.0, 0.5 * Lc, Lc], secs)
        solver = PreissmannSolver(channel=ch, theta=0.7, time_step=300, spatial_step=500, simulation_time=(levels - 1) * 300)
        N = solver.number_of_nodes
        rng = np.random.default_rng(20260216)
        n_members = (0.025 + 0.015 * rng.random(first + B))[first:]
        theta, dt, dx, tol = 0.7, 300.0, solver.spatial_step, 1e-6
        batch = PreissmannBatch(B, N, levels, dtype="f64", section_mode="irregular", device=local, monitor=False)
        batch.set_scheme(theta, dt, dx, tol, 100)
        batch.set_geometry_irregular(ch.node_geometry, n_main_override=n_members)
        batch.set_boundary(A.UPSTREAM, boundary_to_spec(us, levels, dt))
        batch.set_boundary(A.DOWNSTREAM, boundary_to_spec(ds, levels, dt))
        batch.set_state(ch.initial_conditions[:, 0], ch.initial_conditions[:, 1])
        Qb = None
        desc = ("IRR: polyline channel (%d nodes, 8-15 stations per section, composite roughness over three strips)"
                ", %d-member Manning-n ensemble per GPU, theta 0.7, dt 300 s, tol 1e-6" % (N, B))
    else:
        theta, dt, dx = 0.6, 1800.0, 500.0
        tol = 1e-3 if dtype == "f32" else 1e-6
        b_, m_, n_, S0, Qb = c5_reach_parameters(first, B)
        hn = normal_depth_trap(b_, m_, n_, S0, Qb)
        L = (N - 1) * dx
        batch = PreissmannBatch(B, N, levels, dtype=dtype, section_mode="trap_uniform", device=local, monitor=False)
        batch.set_scheme(theta, dt, dx, tol, 100)
        batch.set_geometry_uniform(b_, n_, S0 * L, np.zeros(B), side_slope=m_)
        batch.set_boundary(A.DOWNSTREAM, BoundarySpec(A.BC_RATING_POWER, dict(a=Qb / hn ** 1.6, b=np.full(B, 1.6),
                                                                            stage_shift=np.zeros(B), bed_level=np.zeros(B))))
        desc = ("C5: %d synthetic trapezoidal reaches x %d nodes per GPU, power rating-curve downstream, theta 0.6, "
                "dt 1800 s, dx 500 m, tol %g" % (B, N, tol))
    if workload not in ("c4", "irr"):
        batch.set_boundary(A.UPSTREAM, BoundarySpec(A.BC_FLOW_HYDROGRAPH, {}, inflow_table(Qb, levels, dt)))
        batch.set_state_uniform(hn, Qb)
    batch.sync()
    return batch, N, theta, dt, dx, tol, desc


def roofline_blocks(workload, dtype, N, B, K, kern_ms, its_sum, entry):
    """(roofline, roofline_compute or None) of one launch of K levels over B reaches: algorithmic bytes (4 N sizeof(real)
    + 8 B boundary target + 32 B hydrograph row per reach-timestep) over the kernel's HIP-event time against 8 TB/s, the counter
    traffic of the same launch from the committed profile (null when the profile is of another library or instantiation), and the
    flop rate from the profile's instruction counters against the vector peak of the dtype."""
    real = 8 if dtype == "f64" else 4
    alg = float(B) * K * (4 * N * real + 8 + 32)
    ach = alg / (kern_ms * 1e-3) / 1e9
    prof = load_profile(workload, dtype, N, entry)
    traffic = traffic_of_launch(prof, B, K)
    roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": None if traffic is None else prof["file"] + ": " + prof["traffic_note"],
            "kernel": "preissmann_long_kernel" if entry.get("long_reach") else "preissmann_step_kernel", "kernel_ms": kern_ms, "launches": 1,
            "algorithmic_bytes_per_reach_timestep": 4 * N * real + 40,
            "note": ("multi-pass kernel: the Newton vector passes through HBM every iteration, see DESIGN.md section 4.5" if entry.get("long_reach")
                     else ("a team of workgroups per reach, each holding 64 W M rows on chip, one exchange through device memory per Newton "
                           "iteration: see DESIGN.md section 4.5" if entry.get("team")
                           else "VALU-issue bound, not HBM bound: the levels between load and store never touch HBM, see DESIGN.md section 4"))}
    comp = None
    if prof is not None and prof.get("flops_per_reach_iteration"):
        tf = prof["flops_per_reach_iteration"] * float(its_sum) / (kern_ms * 1e-3) / 1e12
        peak = FP64_VECTOR_PEAK_TFLOPS * (1 if dtype == "f64" else 2)
        comp = {"bound": "valu_" + dtype, "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
                "source": prof["file"] + " (rocprofv3 SQ_INSTS_VALU_* per Newton iteration x the iterations of this launch)"}
    return roof, comp


def run_extra_workload(spec, local):
    """One entry of "workloads": the same measurement as the headline - inputs resident in HBM, W untimed warm-up levels, then K
    levels in ONE launch between two synchronisations - on one of EXTRA_WORKLOADS."""
    import numpy as np
    from flowsim_amd import _abi as A
    workload, dtype, B, N, K, Wm = spec
    batch, N, theta, dt, dx, tol, desc = build_batch(workload, dtype, B, N, K + Wm + 1, 0, local)
    try:
        batch.step(Wm, sync=True)
        t0 = time.perf_counter()
        batch.step(K, sync=False)
        batch.sync()
        el = time.perf_counter() - t0
        kern_ms = batch.last_step_ms()
        st = batch.status()
        its = batch.iterations(Wm + 1, K)
        info, kidx = batch.kernel_info(), batch.kernel_index()
        entry = A.kernel_table()[kidx]
        roof, comp = roofline_blocks(workload, dtype, N, B, K, kern_ms, its.sum(), entry)
        return {"workload": desc, "name": workload, "dtype": dtype, "value": B * K / el, "unit": "reach-timesteps/s", "steps": K, "warmup": Wm,
                "reaches": B, "nodes": N, "ms_per_step": el * 1e3 / K, "kernel_ms": kern_ms,
                "kernel": dict(info, table_index=kidx, boundary_class=entry["boundary_class"], full=entry["full"], diag=entry["diag"],
                               long_reach=entry.get("long_reach", 0), team=entry.get("team", 0)),
                "mean_newton_iterations_per_step": float(its.sum()) / (B * K), "all_converged": bool(np.all(st == 0)),
                "roofline": roof, "roofline_compute": comp}
    finally:
        batch.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--reaches", type=int, default=65536, help="reaches per GPU (weak scaling)")
    ap.add_argument("--total-reaches", type=int, default=None,
                    help="strong scaling: this many reaches in total, split over the ranks in contiguous blocks")
    ap.add_argument("--nodes", type=int, default=4096)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--workload", default="c3", choices=["c3", "c5", "c4", "irr", "long"],
                    help="c3: rectangular, normal-depth outflow (the headline config); c5: SURVEY 8d trapezoid + power "
                         "rating curve (use with --dtype f32 --nodes 512 --reaches 131072); c4: cases/gerd_roseires "
                         "geometry with a Manning-n Monte-Carlo ensemble (use with --reaches 32768; nodes fixed at 121); irr: "
                         "polyline (IrregularSection) channel with a levee, Manning-n ensemble (use with --reaches 8192; "
                         "128 nodes); long: the c3 channels at a node count beyond one lane grid (use with --nodes 16384 --reaches 8192)")
    ap.add_argument("--spatial-step", type=float, default=None,
                    help="c4 only: override the case's spatial step (m); the node count follows (default 1000 m: 121 nodes)")
    ap.add_argument("--per-reach-geometry", action="store_true",
                    help="c4 only: every member gets its own copy of the node table (fs_batch_set_geometry_table_per_reach), as a "
                         "geometry Monte-Carlo would; the shared-table run is the default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the headline workload: without it a default run (C3 on one GPU) goes on to time EXTRA_WORKLOADS - C4, C5 in "
                         "fp32 and fp64, the polyline ensemble, long reaches - and reports them under \"workloads\"")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL)")
    ap.add_argument("--dump-hydrographs", default=None,
                    help="rank 0 saves the gathered boundary hydrographs of the timed levels [K, 4, all reaches] to this .npy "
                         "(tests compare a multi-rank run with a one-process run of the same global reaches, bit for bit)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal only: all ranks use cuda:0 (multi-rank control flow on a one-GPU box, gloo)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")

    import torch
    import torch.distributed as dist
    from flowsim_amd import BoundarySpec, PreissmannBatch
    from flowsim_amd import _abi as A

    if args.share_device:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)

    N, K, Wm = args.nodes, args.steps, args.warmup
    levels = K + Wm + 1
    strong = args.total_reaches is not None
    if strong:
        first, B = split_reaches(args.total_reaches, rank, world)
        if B < 1:
            raise SystemExit("--total-reaches must give every rank at least one reach")
    else:
        B = args.reaches
        first, _ = reach_block(rank, world, B)
    batch, N, theta, dt, dx, tol, desc = build_batch(args.workload, args.dtype, B, N, levels, first, local, args.spatial_step, args.per_reach_geometry)
    batch.sync()

    # device view of the hydrograph block for the RCCL gather (zero copy)
    esz = 8 if args.dtype == "f64" else 4
    tdt = torch.float64 if args.dtype == "f64" else torch.float32

    class _View:
        pass
    v = _View()
    v.__cuda_array_interface__ = {"shape": (levels, 4, B), "typestr": "<f8" if esz == 8 else "<f4",
                                  "data": (batch.hydrograph_device_ptr(), False), "version": 2}
    hyd_dev = torch.as_tensor(v, device=f"cuda:{local}")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        batch.sync()

    if Wm > 0:
        batch.step(Wm, sync=True)
    if world > 1:
        # untimed rehearsal of the one collective of the path (same shape): communicator channels and
        # RCCL's staging buffers are set up on first use
        warm = hyd_dev[1:1 + K] if args.backend == "nccl" else hyd_dev[1:1 + K].cpu()
        gather_hydrographs_split(warm, args.total_reaches, world, 0) if strong else gather_hydrographs(warm, world, 0)
    barrier()
    t0 = time.perf_counter()
    batch.step(K, sync=False)
    batch.sync()
    # the only exchange of the path: boundary hydrographs of the timed levels, gathered as [K, 4, all reaches] on rank 0 (every other
    # rank sends its [K, 4, B] block straight to it: one xGMI hop, 1 / world of the bytes an all_gather would move through every rank)
    timed_rows = hyd_dev[Wm + 1:Wm + 1 + K]
    if args.backend != "nccl" and world > 1:
        timed_rows = timed_rows.cpu()
    if strong:
        gathered = gather_hydrographs_split(timed_rows, args.total_reaches, world, 0)
        assert rank != 0 or gathered.shape == (K, 4, args.total_reaches)
    else:
        gathered = gather_hydrographs(timed_rows, world, 0)
        assert rank != 0 or gathered.shape == (K, 4, world * B)
    barrier()
    el = time.perf_counter() - t0
    kern_ms = batch.last_step_ms()
    if args.dump_hydrographs and rank == 0:
        np.save(args.dump_hydrographs, gathered.cpu().numpy())

    st = batch.status()
    its = batch.iterations(Wm + 1, K)
    if os.environ.get("FS_STAMPS"):          # diagnostic builds (-DFS_STAMP, FS_LIB=...): cycle sums per phase, to stderr
        import ctypes as C
        out = np.zeros((B, 16, 12), dtype=np.uint64)
        lib = A.lib(); lib.fs_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
        assert lib.fs_debug_stamps(batch._h, out.ctypes.data_as(C.c_void_p)) == 0
        tot_it = batch.iterations(1, Wm + K).sum(axis=0).astype(np.float64)
        names = ["fold", "bc", "tree-up", "barrier", "cross-wave", "norm+fence", "down+back", "update+loop", "acc:stores",
                 "acc:hydro", "acc:level-pass", "-"]
        nw = batch.kernel_info()["waves_per_reach"]
        per = out[:, :nw, :].astype(np.float64) / np.maximum(tot_it, 1.0)[:, None, None]
        for w in range(nw):
            print("wave", w, " ".join("%s=%.0f" % (names[i], per[:, w, i].mean()) for i in range(11)),
                  "total=%.0f" % per[:, w].sum(axis=1).mean(), file=sys.stderr)
    ok = bool(np.all(st == 0))
    el_t = torch.tensor([el], dtype=torch.float64, device=f"cuda:{local}")
    kms_t = torch.tensor([kern_ms], dtype=torch.float64, device=f"cuda:{local}")
    it_t = torch.tensor([float(its.sum()), float(ok), float(B)], dtype=torch.float64, device=f"cuda:{local}")
    if world > 1:
        if args.backend != "nccl":
            el_t, kms_t, it_t = el_t.cpu(), kms_t.cpu(), it_t.cpu()
        dist.all_reduce(el_t, op=dist.ReduceOp.MAX)
        dist.all_reduce(kms_t, op=dist.ReduceOp.MAX)
        dist.all_reduce(it_t, op=dist.ReduceOp.SUM)
    el = float(el_t.item()); kern_ms_max = float(kms_t.item())
    # who took part: one row per rank as the collective saw it (rank, device ordinal, PCI bus id, its kernel time, its reaches)
    try:
        bus = int(str(torch.cuda.get_device_properties(local).pci_bus_id))
    except Exception:
        bus = -1
    mine = torch.tensor([float(rank), float(local), float(bus), float(kern_ms), float(B), float(first)], dtype=torch.float64,
                        device=f"cuda:{local}" if (world == 1 or args.backend == "nccl") else "cpu")
    rows = [torch.empty_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(rows, mine)
    else:
        rows = [mine]
    ranks = [dict(rank=int(r[0]), device=int(r[1]), pci_bus_id=int(r[2]), kernel_ms=float(r[3]), reaches=int(r[4]), first_reach=int(r[5]))
             for r in (x.cpu().tolist() for x in rows)]
    info = batch.kernel_info()
    kidx = batch.kernel_index()

    if rank == 0:
        total = float(it_t[2].item()) * K               # reaches of all ranks x timed levels
        mean_its = float(it_t[0].item()) / total
        status_counts = {int(k): int(v) for k, v in zip(*np.unique(st, return_counts=True))}
        entry = A.kernel_table()[kidx]
        # roofline of the step kernel on THIS rank's GPU: algorithmic bytes of its launch / its own HIP-event time
        roof, comp = roofline_blocks(args.workload, args.dtype, N, B, K, kern_ms, its.sum(), entry)
        out = {
            "metric": "reach-timesteps/sec (batched Preissmann Newton step)",
            "value": total / el, "unit": "reach-timesteps/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": el * 1e3 / K, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": desc,
                       "reaches_per_gpu": B, "total_reaches": int(it_t[2].item()), "nodes": N, "parallelism": f"reach-sharded x{world}",
                       "mean_newton_iterations_per_step": mean_its, "all_converged": bool(it_t[1].item() == world),
                       "status_counts_rank0": status_counts,
                       "kernel": dict(info, table_index=kidx, boundary_class=entry["boundary_class"], full=entry["full"], diag=entry["diag"],
                                      long_reach=entry.get("long_reach", 0), team=entry.get("team", 0), tail=entry.get("tail", -1)),
                       "kernel_ms_max_over_ranks": kern_ms_max,
                       "collective_world_size": dist.get_world_size() if world > 1 else 1, "ranks": ranks},
            "roofline": roof,
        }
        if comp is not None:           # informational second roofline: the kernel is bound by vector issue, not by HBM
            out["roofline_compute"] = comp
        if not args.no_cpu_baseline and args.workload == "c3" and world == 1:      # reported at N=1 only
            host_cores = set(os.sched_getaffinity(0))
            out["cpu_baseline"] = cpu_baseline(N, dt, dx, theta, tol)
            out["cpu_baseline_c"] = cpu_baseline(N, dt, dx, theta, tol, compiled=True)
            out["cpu_baseline_c_all_cores"] = cpu_baseline_all_cores(N, dt, dx, theta, tol, host_cores)
    headline = (args.workload == "c3" and args.dtype == "f64" and args.nodes == 4096 and args.reaches == 65536 and not strong and world == 1)
    batch.close()
    if rank == 0:
        if headline and not args.no_extras:
            # the other configurations, driver-observed in the same line (the C3 batch is closed: its 8 GiB are free again)
            out["workloads"] = []
            for spec in EXTRA_WORKLOADS:
                try:
                    out["workloads"].append(run_extra_workload(spec, local))
                except Exception as exc:          # the headline line is printed whatever happens to an extra
                    out["workloads"].append({"name": spec[0], "dtype": spec[1], "error": f"{type(exc).__name__}: {exc}"})
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
