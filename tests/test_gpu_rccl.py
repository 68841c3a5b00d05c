"""RCCL (torch.distributed backend "nccl") on the library's own device memory.  A one-GPU box cannot
host two RCCL ranks, so this runs a world of ONE in a child process: process-group creation on the
device, barrier, all_reduce and - bypassing the single-rank shortcut of gather_hydrographs - the
all_gather_into_tensor that bench.py issues on the zero-copy view of the batch's hydrograph block
(memory allocated by libflowsim_hip.so, not by torch).  The multi-rank control flow is covered on the
CPU by tests/test_sharding_gloo.py."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CHILD = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "flow-sim_amd")); sys.path.insert(0, ROOT)
    from flowsim_amd import BoundarySpec, PreissmannBatch, _abi as A
    from flowsim_amd.synthetic import c3_reach_parameters, inflow_table, normal_depth_rect

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    B, N, K = 64, 300, 4
    b_, n_, S0, Qb = c3_reach_parameters(0, B); hn = normal_depth_rect(b_, n_, S0, Qb); L = (N - 1) * 250.0
    bt = PreissmannBatch(B, N, K + 1, section_mode="rect_uniform")
    bt.set_scheme(0.6, 600.0, 250.0, 1e-6, 100); bt.set_geometry_uniform(b_, n_, S0 * L, np.zeros(B))
    bt.set_boundary(A.UPSTREAM, BoundarySpec(A.BC_FLOW_HYDROGRAPH, {}, inflow_table(Qb, K + 1, 600.0)))
    bt.set_boundary(A.DOWNSTREAM, BoundarySpec(A.BC_NORMAL_DEPTH, dict(bed_slope=S0, bed_level=np.zeros(B))))
    bt.set_state_uniform(hn, Qb); bt.step(K, sync=True)

    class V: pass
    v = V()
    v.__cuda_array_interface__ = {"shape": (K + 1, 4, B), "typestr": "<f8", "data": (bt.hydrograph_device_ptr(), False), "version": 2}
    view = torch.as_tensor(v, device="cuda:0")
    rows = view[1:1 + K].contiguous()
    out = torch.empty((1,) + tuple(rows.shape), dtype=rows.dtype, device="cuda:0")
    dist.all_gather_into_tensor(out.view(K, 4, B), rows)                 # the all-ranks form of gather_hydrographs
    out2 = torch.empty_like(out)
    dist.gather(rows, [out2[0]], dst=0)                                  # the collective of bench.py: a gather to rank 0
    assert torch.equal(out2, out)
    t = torch.tensor([float(K)], dtype=torch.float64, device="cuda:0")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    torch.cuda.synchronize()
    host = bt.hydrographs(1, K)
    assert np.array_equal(out[0].cpu().numpy(), host), "gathered rows differ from fs_batch_get_hydrographs"
    assert t.item() == K and np.all(bt.status() == 0)
    bt.close()
    dist.destroy_process_group()
    print("rccl ok")
""")


def test_rccl_collectives_on_library_memory():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "rccl ok" in r.stdout
