"""Debug aid: run the C5 workload (trapezoid + power rating curve; --workload c3: rectangular + normal depth,
--dx / --dt to move the population across regimes) with the residual trace on, list the reaches
whose status is not FS_OK and replay each alone (B = 1, same parameters) - with the kernel shape and
library taken from FS_KERNEL_SHAPE / FS_LIB as usual.
usage: python tools/find_bad_reach.py [--dtype f32] [--nodes 512] [--reaches 131072] [--split 4,32] [--only IDX]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flow-sim_amd"))
from flowsim_amd import BoundarySpec, PreissmannBatch
from flowsim_amd import _abi as A
from flowsim_amd.synthetic import C5_SEED, c3_reach_parameters, c5_reach_parameters, inflow_table, normal_depth_rect, normal_depth_trap


def run(first, B, N, levels, dtype, trace, split, seed=C5_SEED):
    theta, dt, dx = 0.6, a.dt, a.dx
    tol = 1e-3 if dtype == "f32" else 1e-6
    if a.workload == "c5":
        b_, m_, n_, S0, Qb = c5_reach_parameters(first, B, seed)
        hn = normal_depth_trap(b_, m_, n_, S0, Qb)
    else:
        b_, n_, S0, Qb = c3_reach_parameters(first, B, seed)
        m_ = np.zeros(B)
        hn = normal_depth_rect(b_, n_, S0, Qb)
    batch = PreissmannBatch(B, N, levels, dtype=dtype, section_mode="trap_uniform" if a.workload == "c5" else "rect_uniform",
                            trace=trace)
    batch.set_scheme(theta, dt, dx, tol, 100)
    if a.workload == "c5":
        batch.set_geometry_uniform(b_, n_, S0 * (N - 1) * dx, np.zeros(B), side_slope=m_)
        batch.set_boundary(A.DOWNSTREAM, BoundarySpec(A.BC_RATING_POWER, dict(a=Qb / hn ** 1.6, b=np.full(B, 1.6),
                                                                            stage_shift=np.zeros(B), bed_level=np.zeros(B))))
    else:
        batch.set_geometry_uniform(b_, n_, S0 * (N - 1) * dx, np.zeros(B))
        batch.set_boundary(A.DOWNSTREAM, BoundarySpec(A.BC_NORMAL_DEPTH, dict(bed_slope=S0, bed_level=np.zeros(B))))
    batch.set_boundary(A.UPSTREAM, BoundarySpec(A.BC_FLOW_HYDROGRAPH, {}, inflow_table(Qb, levels, dt)))
    batch.set_state_uniform(hn, Qb)
    for k in split:
        batch.step(k)
    return batch, (b_, m_, n_, S0, Qb, hn)


ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c5", choices=["c5", "c3"])
ap.add_argument("--dx", type=float, default=500.0); ap.add_argument("--dt", type=float, default=1800.0)
ap.add_argument("--dtype", default="f32"); ap.add_argument("--nodes", type=int, default=512)
ap.add_argument("--reaches", type=int, default=131072)
ap.add_argument("--only", type=int, default=-1)
ap.add_argument("--seed", type=int, default=C5_SEED)
ap.add_argument("--split", default="4,32", help="levels per launch")
ap.add_argument("--show", default="", help="first,last level to print (default: around the failure)")
a = ap.parse_args()
split = [int(x) for x in a.split.split(',')]
a.steps = sum(split)
levels = a.steps + 1
if a.only < 0:
    batch, par = run(0, a.reaches, a.nodes, levels, a.dtype, False, split, a.seed)
    st = batch.status(); its = batch.iterations(1, a.steps)
    bad = np.nonzero(st)[0]
    print("kernel", batch.kernel_info(), "iterations mean %.4f max %d" % (its.mean(), its.max()), "bad reaches", bad[:16], "status", st[bad][:16])
    batch.close()
else:
    bad = np.array([a.only])
for r in bad[:4]:
    b1, par = run(int(r), 1, a.nodes, levels, a.dtype, True, split, a.seed)
    st = b1.status()[0]; its = b1.iterations(1, a.steps)[:, 0]
    print(f"reach {r}: b {par[0][0]:.4f} m {par[1][0]:.4f} n {par[2][0]:.5f} S0 {par[3][0]:.3e} Qb {par[4][0]:.3f} hn {par[5][0]:.4f}")
    print("  alone: status", st, "iterations per level", its.tolist())
    tr = b1.residual_trace(1, a.steps)[:, :, 0]
    lv = int(np.argmax(its == 0)) if st != 0 and (its == 0).any() else len(its) - 1
    lo, hi = (max(0, lv - 2), min(len(its), lv + 1)) if not a.show else tuple(int(x) for x in a.show.split(','))
    for k in range(lo, hi):
        print(f"  level {k + 1}: residuals", np.array2string(tr[k][:8], precision=4))
    hy = b1.hydrographs()
    print("  last rows of the hydrograph block (h0,Q0,hN,QN):", np.array2string(hy[max(0, lv - 1):lv + 2, :, 0], precision=5))
    if st != 0:
        hk, Qk = b1.state(); hg, Qg = b1.guess()
        for nm, arr in (("h stored", hk), ("Q stored", Qk), ("h iterate", hg), ("Q iterate", Qg)):
            v = np.asarray(arr)[0]
            bad_i = np.nonzero(~np.isfinite(v))[0]
            print(f"  {nm}: level {b1.level} non-finite at {bad_i[:12].tolist()} ({len(bad_i)} of {len(v)}), finite range "
                  f"{np.nanmin(v):.5g} .. {np.nanmax(v):.5g}")
        np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", f"bad_reach_{r}.npy"),
                np.stack([np.asarray(x)[0] for x in (hk, Qk, hg, Qg)]))
    b1.close()
