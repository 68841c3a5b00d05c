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
bt = PreissmannBatch(B, N, K + 1, section_mode="rect_uniform", monitor=False)
bt.set_scheme(0.6, 600.0, 250.0, 1e-6, 100); bt.set_geometry_uniform(b_, n_, S0 * L, np.zeros(B))
bt.set_boundary(A.UPSTREAM, BoundarySpec(A.BC_FLOW_HYDROGRAPH, {}, inflow_table(Qb, K + 1, 600.0)))
bt.set_boundary(A.DOWNSTREAM, BoundarySpec(A.BC_NORMAL_DEPTH, dict(bed_slope=S0, bed_level=np.zeros(B))))
h = np.repeat(hn[:, None], N, axis=1); Q = np.repeat(Qb[:, None], N, axis=1)          # 2 x B*N*8 bytes of host memory
bt.sync()
t0 = time.perf_counter(); bt.set_state(h, Q); bt.sync(); t_up = time.perf_counter() - t0
t0 = time.perf_counter(); bt.step(K, sync=True); t_step = time.perf_counter() - t0
h2, Q2 = np.zeros_like(h), np.zeros_like(Q)              # the caller's result buffers exist (and are mapped) before the clock starts
t0 = time.perf_counter(); bt.state(out=(h2, Q2)); t_down = time.perf_counter() - t0
t0 = time.perf_counter(); h3, Q3 = bt.state(); t_down_fresh = time.perf_counter() - t0     # fresh np.empty buffers: page faults included
assert np.all(bt.status() == 0) and np.array_equal(h2, h3) and np.array_equal(Q2, Q3)
hx, Qx = bt.state(); assert np.all(np.isfinite(hx)) and abs(hx.mean() / h.mean() - 1) < 0.5
gb = B * N * 8 * 2 / 1e9
# the same work as stream-ordered blocks of reaches (flowsim_amd.pipeline): block i + 1 uploads while block i steps, block i - 1
# downloads meanwhile
from flowsim_amd.pipeline import step_pipelined
bt.close(); del h3, Q3, hx, Qx


def make(lo, nb):
    x = PreissmannBatch(nb, N, K + 1, section_mode="rect_uniform", monitor=False)
    x.set_scheme(0.6, 600.0, 250.0, 1e-6, 100); x.set_geometry_uniform(b_[lo:lo + nb], n_[lo:lo + nb], S0[lo:lo + nb] * L, np.zeros(nb))
    x.set_boundary(A.UPSTREAM, BoundarySpec(A.BC_FLOW_HYDROGRAPH, {}, inflow_table(Qb[lo:lo + nb], K + 1, 600.0)))
    x.set_boundary(A.DOWNSTREAM, BoundarySpec(A.BC_NORMAL_DEPTH, dict(bed_slope=S0[lo:lo + nb], bed_level=np.zeros(nb))))
    x.sync()
    return x


ref_h, ref_Q = h2.copy(), Q2.copy()


def pipelined(n_chunks):
    nb = B // n_chunks
    parts = [make(i * nb, nb) for i in range(n_chunks)]
    h2[:] = 0; Q2[:] = 0
    t0 = time.perf_counter()
    step_pipelined(parts, h, Q, K, out=(h2, Q2))
    t = time.perf_counter() - t0
    assert np.array_equal(h2, ref_h) and np.array_equal(Q2, ref_Q)          # same bits as the one-batch run
    for x in parts:
        x.close()
    return t


chunk_counts = [int(c) for c in os.environ.get("FS_PCIE_CHUNKS", "2,4,8,16").split(",") if B % int(c) == 0 and B // int(c) >= 1]
t_chunks = {c: pipelined(c) for c in chunk_counts}
t_pipe = t_chunks.get(2, min(t_chunks.values()))
print(json.dumps({"reaches": B, "nodes": N, "levels": K, "upload_s": t_up, "step_s": t_step, "download_s": t_down,
                  "upload_GBps": gb / t_up, "download_GBps": gb / t_down, "download_into_fresh_buffers_s": t_down_fresh,
                  "download_into_fresh_buffers_GBps": gb / t_down_fresh, "device_resident_rts_per_s": B * K / t_step,
                  "pcie_inclusive_rts_per_s": B * K / (t_up + t_step + t_down),
                  "two_halves_pipelined_s": t_pipe, "pcie_inclusive_two_halves_rts_per_s": B * K / t_pipe,
                  "pipelined_chunks_s": {str(c): t for c, t in t_chunks.items()},
                  "pcie_inclusive_pipelined_rts_per_s": {str(c): B * K / t for c, t in t_chunks.items()}}))
