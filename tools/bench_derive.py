"""HBM roofline of the post-processing kernel (fs_batch_derive): bytes moved / HIP-event time."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-sim_amd"))
from flowsim_amd import BoundarySpec, PreissmannBatch, _abi as A
from flowsim_amd.synthetic import c3_reach_parameters, inflow_table, normal_depth_rect
B, N, K = 2048, 4096, 6
b_, n_, S0, Qb = c3_reach_parameters(0, B); hn = normal_depth_rect(b_, n_, S0, Qb); L = (N - 1) * 250.0
bt = PreissmannBatch(B, N, K + 1, section_mode="rect_uniform", history=True)
bt.set_scheme(0.6, 600.0, 250.0, 1e-6, 100); bt.set_geometry_uniform(b_, n_, S0 * L, np.zeros(B))
bt.set_boundary(A.UPSTREAM, BoundarySpec(A.BC_FLOW_HYDROGRAPH, {}, inflow_table(Qb, K + 1, 600.0)))
bt.set_boundary(A.DOWNSTREAM, BoundarySpec(A.BC_NORMAL_DEPTH, dict(bed_slope=S0, bed_level=np.zeros(B))))
bt.set_state_uniform(hn, Qb); bt.step(K)
import json
n = K + 1
bytes_ = B * N * 8 * (n * 2 + n * 7 + 1 + 1)      # h,Q in; 7 fields out per level; depth[0] in; peak out
res = []
for rep in range(4):
    bt.derive_device(0, n)                        # results stay on the device: the kernel alone (fs_batch_derive_device)
    bt.sync()
    ms = bt.last_step_ms()
    res.append(ms)
    print(f"derive: {B}x{N} x {n} levels  {ms:.3f} ms  {bytes_/ms/1e6:.1f} GB/s  ({bytes_/ms/1e6/8000*100:.1f}% of 8 TB/s)")
ms = float(np.median(res[1:]))
print(json.dumps({"kernel": "derive_fields_kernel", "reaches": B, "nodes": N, "levels": n, "algorithmic_bytes": bytes_, "kernel_ms": ms,
                  "achieved_GBps": bytes_ / ms / 1e6, "frac_of_8TBps": bytes_ / ms / 1e6 / 8000}))
