"""How the unpivoted on-chip elimination behaves outside its home ground: the supercritical draws of the random sweep
(tests/test_gpu_random_cases.py skips them) against the pivoted LU of the C oracle and SuperLU of the numpy oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-sim_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_random_cases as T
from fixture_batch import batch_from_problems
from oracle import c_oracle as CO, preissmann_oracle as O
rows = []
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8000):
    p, info = T.random_problem(seed)
    if info["froude"] < 0.9:
        continue
    rc = CO.run(p)
    if rc["status"] != 0 or not np.all(np.isfinite(rc["depth"])):
        continue
    rn = O.newton_run(p) if os.environ.get('SCAN_NUMPY') else dict(status=1, depth=rc['depth'])
    mode = "rect_uniform" if (not info["trapezoid"] and info["ds"] != "blend") else "table"
    with batch_from_problems([p], mode=mode, history=True) as b:
        b.step(p.nt - 1)
        st = int(b.status()[0])
        h, Q = b.history_arrays(0, p.nt)
        its = b.iterations(0, p.nt)[:, 0]
    err = lambda a, r: float(np.max(np.abs(a - r) / np.maximum(np.abs(r), 1e-3 * info["hn"])))
    n = min(len(rc["depth"]), len(rn["depth"]))
    rows.append((info["froude"], seed, info["N"], st, err(h[:len(rc["depth"]), 0], rc["depth"]) if st in (0, 4) else np.nan,
                 err(rc["depth"][:n], rn["depth"][:n]) if rn["status"] == 0 else np.nan, int(np.sum(its)), int(np.sum(rc["iters"]))))
rows.sort()
print("Froude  seed     N  status  gpu-vs-C   C-vs-numpy  its(gpu) its(C)")
for r in rows:
    print("%5.2f %6d %5d %6d   %9.2e   %9.2e   %6d %6d" % r)
a = np.array([r[4] for r in rows if np.isfinite(r[4])]); f = np.array([r[0] for r in rows if np.isfinite(r[4])])
for lo, hi in ((0.9, 1.0), (1.0, 1.2), (1.2, 1.5), (1.5, 9.0)):
    m = (f >= lo) & (f < hi)
    if m.any():
        print("Fr %.1f-%.1f: %3d draws, gpu-vs-C median %.1e max %.1e, above 1e-8: %d" % (lo, hi, m.sum(), np.median(a[m]), a[m].max(), (a[m] > 1e-8).sum()))
