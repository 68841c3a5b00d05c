"""sha256 of what the flagship kernel leaves behind (state, Newton start vector, hydrograph rows, iteration counts) on 96 reaches x
4 096 nodes x 6 levels: `FS_LIB=…/lib_x.so python tools/variant_digest.py` - experiment builds that only move instructions must
print the digest of the shipped library."""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-sim_amd"))
from flowsim_amd import BoundarySpec, PreissmannBatch, _abi as A
from flowsim_amd.synthetic import c3_reach_parameters, inflow_table, normal_depth_rect
B, N, K = 96, int(os.environ.get("FS_DIGEST_NODES", "4096")), 6
b_, n_, S0, Qb = c3_reach_parameters(0, B); hn = normal_depth_rect(b_, n_, S0, Qb); L = (N - 1) * 250.0
with PreissmannBatch(B, N, K + 1, section_mode="rect_uniform", monitor=False) as x:
    x.set_scheme(0.6, 600.0, 250.0, 1e-6, 100); x.set_geometry_uniform(b_, n_, S0 * L, np.zeros(B))
    x.set_boundary(A.UPSTREAM, BoundarySpec(A.BC_FLOW_HYDROGRAPH, {}, inflow_table(Qb, K + 1, 600.0)))
    x.set_boundary(A.DOWNSTREAM, BoundarySpec(A.BC_NORMAL_DEPTH, dict(bed_slope=S0, bed_level=np.zeros(B))))
    x.set_state_uniform(hn, Qb)
    x.step(K)
    h, Q = x.state(); hg, Qg = x.guess()
    m = hashlib.sha256()
    for a in (h, Q, hg, Qg, x.hydrographs(0, K + 1), x.iterations(0, K + 1), x.status()):
        m.update(np.ascontiguousarray(a).tobytes())
    e = A.kernel_table()[x.kernel_index()]
    print(m.hexdigest()[:16], "its", int(x.iterations(0, K + 1).sum()), "kernel", e["cells_per_thread"], e["waves_per_reach"], "diag", e["diag"])
