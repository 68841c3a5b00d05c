"""Two ranks around the real kernel.  The build box has ONE MI355X, so both ranks of `python -m torch.distributed.run
--nproc-per-node 2 bench.py --gpus 2 --share-device --backend gloo ...` drive libflowsim_hip.so on cuda:0 (two processes, two
HIP contexts, two batches) and exchange the boundary hydrographs over gloo; what is checked is everything of the multi-GPU
path that does not depend on the fabric: rank -> reach-block mapping (weak: rank r owns [r B, (r+1) B); strong:
split_reaches), the per-rank parameter draws, the padded gather of unequal blocks to rank 0, the timing reductions, the JSON line.
The gathered hydrographs of the two-rank run must equal those of a ONE-process run over the same global reaches bit for bit
(reaches are independent; a reach's result may not depend on which rank or batch position it ran at).
(The launcher starts before anything touches the GPU; the test process itself only counts as the third GPU user.)"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
COMMON = ["--nodes", "300", "--steps", "4", "--warmup", "1", "--no-cpu-baseline"]


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_bench(extra, dump, ranks, common=None):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    bench = os.path.join(ROOT, "bench.py")
    if ranks == 1:
        cmd = [sys.executable, bench, "--gpus", "1"]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), bench, "--gpus", str(ranks), "--share-device", "--backend", "gloo"]
    r = subprocess.run(cmd + (common or COMMON) + extra + ["--dump-hydrographs", dump], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # ONE JSON line, from rank 0
    return json.loads(lines[0]), np.load(dump)


@pytest.mark.parametrize("layout", ["weak", "strong"])
def test_two_ranks_equal_one_process_bit_for_bit(layout, tmp_path):
    two_args = ["--reaches", "64"] if layout == "weak" else ["--total-reaches", "127"]
    one_args = ["--reaches", "128"] if layout == "weak" else ["--total-reaches", "127"]
    total = 128 if layout == "weak" else 127
    line2, hyd2 = run_bench(two_args, str(tmp_path / "two.npy"), 2)
    line1, hyd1 = run_bench(one_args, str(tmp_path / "one.npy"), 1)
    assert hyd2.shape == (4, 4, total) and hyd1.shape == hyd2.shape
    assert np.array_equal(hyd2, hyd1)                                   # same bits whichever rank stepped the reach
    assert np.all(np.isfinite(hyd2)) and np.ptp(hyd2[:, 1, :], axis=0).max() > 0       # a flood wave is in it
    # the line: whole-job numbers and who took part
    assert line2["n_gpus"] == 2 and line2["scaling"] == layout and line2["config"]["collective_world_size"] == 2
    assert line2["config"]["total_reaches"] == total and line2["config"]["all_converged"]
    ranks = line2["config"]["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and all(r["kernel_ms"] > 0 for r in ranks)
    assert [r["reaches"] for r in ranks] == ([64, 64] if layout == "weak" else [64, 63])
    assert [r["first_reach"] for r in ranks] == [0, 64]
    assert abs(line2["value"] - total * 4 / (line2["ms_per_step"] * 4e-3)) <= 1e-6 * line2["value"]
    assert line1["n_gpus"] == 1 and len(line1["config"]["ranks"]) == 1 and line1["config"]["ranks"][0]["reaches"] == total
    assert line1["config"]["mean_newton_iterations_per_step"] == pytest.approx(line2["config"]["mean_newton_iterations_per_step"], rel=1e-12)


def test_two_processes_step_long_reaches_on_one_gpu_at_the_same_time(tmp_path):
    """Reaches of 12 000 nodes: every reach a TEAM of three workgroups that meet once per Newton iteration through device memory
    (fs_kernel.hpp, TEAM).  Two processes launch such kernels on the one GPU at the same time - 768 workgroups each, more than the chip
    holds at once, so the two grids interleave on the CUs.  Membership by ticket means a team never waits for a workgroup that cannot
    start, whatever the other process occupies: both runs complete, every reach converges, and the gathered hydrographs equal those of
    one process stepping all 512 reaches, bit for bit."""
    common = ["--nodes", "12000", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--workload", "long"]
    line2, hyd2 = run_bench(["--reaches", "256"], str(tmp_path / "two.npy"), 2, common)
    line1, hyd1 = run_bench(["--reaches", "512"], str(tmp_path / "one.npy"), 1, common)
    assert hyd2.shape == (4, 4, 512) and np.array_equal(hyd2, hyd1)
    assert line2["config"]["all_converged"] and line1["config"]["all_converged"]
    assert line2["config"]["kernel"]["team"] == 1 and line1["config"]["kernel"]["team"] == 1
