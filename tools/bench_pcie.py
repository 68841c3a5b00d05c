"""PCIe-inclusive rate of the flagship workload: the state handed over and taken back as host buffers
(fs_batch_set_state / fs_batch_get_state) around the 32 timed levels, next to the device-resident rate bench.py reports."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flow-sim_amd"))
from flowsim_amd import BoundarySpec, PreissmannBatch, _abi as A
from flowsim_amd.synthetic import c3_reach_parameters, inflow_table, normal_depth_rect
B, N, K = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 4096, 32
b_, n_, S0, Qb = c3_reach_parameters(0, B); hn = normal_depth_rect(b_, n_, S0, Qb); L = (N - 1) * 250.0
bt = PreissmannBatch(B, N, K + 1, section_mode="rect_uniform")
bt.set_scheme(0.6, 600.0, 250.0, 1e-6, 100); bt.set_geometry_uniform(b_, n_, S0 * L, np.zeros(B))
bt.set_boundary(A.UPSTREAM, BoundarySpec(A.BC_FLOW_HYDROGRAPH, {}, inflow_table(Qb, K + 1, 600.0)))
bt.set_boundary(A.DOWNSTREAM, BoundarySpec(A.BC_NORMAL_DEPTH, dict(bed_slope=S0, bed_level=np.zeros(B))))
h = np.repeat(hn[:, None], N, axis=1); Q = np.repeat(Qb[:, None], N, axis=1)          # 2 x B*N*8 bytes of host memory
bt.sync()
t0 = time.perf_counter(); bt.set_state(h, Q); bt.sync(); t_up = time.perf_counter() - t0
t0 = time.perf_counter(); bt.step(K, sync=True); t_step = time.perf_counter() - t0
t0 = time.perf_counter(); h2, Q2 = bt.state(); t_down = time.perf_counter() - t0
assert np.all(bt.status() == 0)
gb = B * N * 8 * 2 / 1e9
print(json.dumps({"reaches": B, "nodes": N, "levels": K, "upload_s": t_up, "step_s": t_step, "download_s": t_down,
                  "upload_GBps": gb / t_up, "download_GBps": gb / t_down, "device_resident_rts_per_s": B * K / t_step,
                  "pcie_inclusive_rts_per_s": B * K / (t_up + t_step + t_down)}))
