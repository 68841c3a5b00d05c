"""One polyline fixture on one kernel shape against the reference's own history: python tools/repro_poly_shape.py FIXTURE M,W [class]
(FS_LIB selects an experiment build).  Prints status, Newton counts and the largest deviation - the quickest way to see whether an
instantiation is sound."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "flow-sim_amd"))
name, shape = sys.argv[1], sys.argv[2]
os.environ["FS_KERNEL_SHAPE"] = shape
if len(sys.argv) > 3 and sys.argv[3] == "general":
    os.environ["FS_KERNEL_GENERAL"] = "1"
from oracle import preissmann_oracle as O
from fixture_batch import batch_from_problems
fx, meta = O.load_fixture(os.path.join(ROOT, "tests", "golden", name + ".npz"))
p = O.problem_from_fixture(fx, meta)
p.nt = min(p.nt, 9)
with batch_from_problems([p], mode="irregular", history=True) as b:
    b.step(p.nt - 1)
    k = b.kernel_info(); k["index"] = b.kernel_index()
    st = b.status()
    h, Q = b.history_arrays(0, p.nt)
    its = b.iterations(0, p.nt)[:, 0]
dev = np.max(np.abs(h[:, 0] - fx["depth"][:p.nt]) / np.maximum(np.abs(fx["depth"][:p.nt]), 1e-3))
print(name, shape, k, "status", st, "its", its.tolist(), "dev %.2e" % dev,
      "nan nodes", np.argwhere(~np.isfinite(h[:, 0])).tolist()[:6])
