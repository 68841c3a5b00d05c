"""Diagnostic: per-phase cycle shares of the Newton iteration from a -DFS_STAMP build (FS_LIB=...)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-sim_amd")); sys.path.insert(0, ROOT)
from flowsim_amd import BoundarySpec, PreissmannBatch, _abi as A
from flowsim_amd.synthetic import c3_reach_parameters, inflow_table, normal_depth_rect
B, N, K = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, int(sys.argv[2]) if len(sys.argv) > 2 else 4096, 8
b_, n_, S0, Qb = c3_reach_parameters(0, B); hn = normal_depth_rect(b_, n_, S0, Qb); L = (N - 1) * 250.0
bt = PreissmannBatch(B, N, K + 2, section_mode="rect_uniform", monitor=False)
bt.set_scheme(0.6, 600.0, 250.0, 1e-6, 100); bt.set_geometry_uniform(b_, n_, S0 * L, np.zeros(B))
bt.set_boundary(A.UPSTREAM, BoundarySpec(A.BC_FLOW_HYDROGRAPH, {}, inflow_table(Qb, K + 2, 600.0)))
bt.set_boundary(A.DOWNSTREAM, BoundarySpec(A.BC_NORMAL_DEPTH, dict(bed_slope=S0, bed_level=np.zeros(B))))
bt.set_state_uniform(hn, Qb); bt.step(K)
its = bt.iterations(1, K).sum(axis=0)
out = np.zeros((B, 16, 12), dtype=np.uint64)
lib = A.lib(); lib.fs_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
assert lib.fs_debug_stamps(bt._h, out.ctypes.data_as(C.c_void_p)) == 0
W = bt.kernel_info()["waves_per_reach"]
if A.kernel_table()[bt.kernel_index()].get("team"):          # a team's rows: (member, wave), up to 16
    W = min(16, W * ((N + 64 * W * bt.kernel_info()["cells_per_thread"] - 1) // (64 * W * bt.kernel_info()["cells_per_thread"])))
names = ["fold", "bc", "tree-up", "barrier-wait", "cross-wave", "update(+accept tail)", "down+back", "loop-top", "acc:stores", "acc:hydro", "acc:level-pass", "team:post+wait"]
per_it = out[:, :W, :].astype(np.float64) / its[:, None, None]
print(f"kernel {bt.kernel_info()}  ms {bt.last_step_ms():.2f}  mean its/step {its.mean()/K:.2f}")
tot = per_it.sum(axis=2).mean()
for w in range(W):
    print("wave", w, " ".join(f"{names[i]}={per_it[:, w, i].mean():8.0f}" for i in range(12)), f" total={per_it[:, w].sum(axis=1).mean():.0f}")
print("mean cycles/iteration/wave", tot)
