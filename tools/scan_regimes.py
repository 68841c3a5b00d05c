"""Robustness scan of the unpivoted elimination: trapezoid reaches over a grid of bed slope x spatial step x
base flow (from backwater-resolved grids to kinematic ones far beyond h* = (5/3) Se dx),
fp64 HIP path against the partially pivoted banded LU of the C oracle.  Prints one line per case.
usage: python tools/scan_regimes.py [--nodes 200] [--steps 6]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flow-sim_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from fixture_batch import batch_from_problems
from oracle import c_oracle
from synth import trap_problem

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=200); ap.add_argument("--steps", type=int, default=6)
a = ap.parse_args()
worst = 0.0
for S0 in (1e-4, 5e-4, 1e-3, 2e-3, 5e-3):
    for dx in (100.0, 500.0, 2000.0):
        for q in (0.3, 1.0, 4.0):                       # base flow per metre of bed width
            b, m, n = 40.0, 2.0, 0.03
            p = trap_problem(b, m, n, S0, q * b, a.nodes, a.steps, dx=dx)
            ref = c_oracle.run(p)
            with batch_from_problems([p], mode="trap_uniform") as bt:
                bt.step(a.steps)
                st = int(bt.status()[0]); h, Q = bt.history_arrays(); its = bt.iterations()[:, 0]
            hn = p.h0[0]
            v = q * b / ((b + m * hn) * hn)
            fr = v / np.sqrt(9.81 * (b + m * hn) * hn / (b + 2 * m * hn))
            eh = np.max(np.abs(h[:, 0] - ref["depth"]) / np.maximum(np.abs(ref["depth"]), 1e-3))
            eq = np.max(np.abs(Q[:, 0] - ref["flow"]) / np.maximum(np.abs(ref["flow"]), 1.0))
            same = np.array_equal(its, ref["iters"])
            worst = max(worst, eh, eq) if st == 0 and ref["status"] == 0 else worst
            print(f"S0 {S0:.0e} dx {dx:6.0f} q {q:3.1f}: hn {hn:.3f} h/h* {hn / (5 / 3 * S0 * dx):7.3f} Fr {fr:.2f} | status gpu {st} "
                  f"oracle {ref['status']} | err h {eh:.1e} Q {eq:.1e} | iterations {'same' if same else str(its.tolist()) + ' vs ' + str(list(ref['iters']))}")
print("worst relative error over the cases both sides solved:", f"{worst:.2e}")
