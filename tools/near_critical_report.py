"""The near-critical fixture (tests/golden/near_critical.npz, run by the reference) against the two oracles and the kernel:
deviation from the reference's depth / flow histories, Newton counts, the kernel's status.  Kept as
profiles/round3/near_critical_report.txt; tests/test_near_critical.py asserts the same facts."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-sim_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import preissmann_oracle as O, c_oracle as CO
from fixture_batch import batch_from_problems, is_rect_uniform


def dev(d, f, fx, m):
    rel = lambda g, w, fl: float(np.max(np.abs(g - w) / np.maximum(np.abs(w), fl)))
    return max(rel(d, fx["depth"], 1e-3 * m["h_n"]), rel(f, fx["flow"], 1e-3 * m["Qb"])) if d.shape == fx["depth"].shape else np.inf


print("case  N    family    Fr_max  cond1(J)   numpy-vs-ref  C-vs-ref   kernel-vs-ref (table / rect)  status  counts equal (numpy C kernel)")
for i, fx, m in O.sweep_cases(os.path.join(ROOT, "tests", "golden", "near_critical.npz")):
    p = O.problem_from_fixture(fx, m)
    rn, rc = O.newton_run(p), CO.run(p)
    out = []
    for mode in ["table"] + (["rect_uniform"] if is_rect_uniform(p) else []):
        with batch_from_problems([p], mode=mode, history=True) as b:
            b.step(p.nt - 1)
            st = int(b.status()[0]); h, Q = b.history_arrays(0, p.nt); its = b.iterations(0, p.nt)[:, 0]
        out.append((dev(h[:, 0], Q[:, 0], fx, m), st, bool(np.array_equal(its, fx["iters"]))))
    eq = lambda r: "y" if np.array_equal(r["iters"], fx["iters"]) else "n"
    print(f"{i:3d} {m['N']:4d}  {m['family']:8s}  {m['froude_max']:5.2f}  {m['cond1_max']:9.1e}   {dev(rn['depth'], rn['flow'], fx, m):9.1e}   {dev(rc['depth'], rc['flow'], fx, m):9.1e}   "
          + " / ".join(f"{d:8.1e}" for d, _, _ in out) + f"   {'/'.join(str(s) for _, s, _ in out):5s}   {eq(rn)} {eq(rc)} {'/'.join('y' if e else 'n' for _, _, e in out)}")
