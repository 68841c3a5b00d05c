"""Build invariants of libflowsim_hip.so (CPU: hipcc cross-compiles, llvm-objdump reads the code objects).

No device function may be called out of line (tools/check_isa.py; why: DESIGN.md section 4.4 - round 3's wrong polyline results and
GPU fault came from interprocedural register allocation around one call the inliner had left).  The Makefile runs the check
between compiling and linking; here it is run on the objects the shipped library was linked from, and shown to trip on a kernel
that does call."""
import glob
import os
import subprocess
import sys

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "flow-sim_amd", "csrc")
CHECK = os.path.join(ROOT, "tools", "check_isa.py")
HIPCC = "/opt/rocm/bin/hipcc"


def test_no_device_function_of_the_shipped_library_is_called_out_of_line():
    objs = sorted(glob.glob(os.path.join(CSRC, "build", "*.o")))
    if not objs:
        pytest.skip("no build directory next to the library (the GPU box receives the built .so only)")
    so = os.path.join(CSRC, "libflowsim_hip.so")
    assert all(os.path.getmtime(o) <= os.path.getmtime(so) + 1 for o in objs), "objects newer than the library: run make"
    r = subprocess.run([sys.executable, CHECK] + objs, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("no call out of line") == len(objs)


def test_the_makefile_runs_the_check_and_disables_ipra():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    assert "check_isa.py $(OBJS)" in mk and "-mllvm -enable-ipra=0" in mk
    # the check sits between the objects and the link line of the library's rule
    rule = mk[mk.index("libflowsim_hip.so:"):]
    assert rule.index("objs") < rule.index("check_isa.py") < rule.index("-shared")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_the_check_trips_on_a_kernel_that_calls(tmp_path):
    src = tmp_path / "calls.hip"
    src.write_text("""
#include <hip/hip_runtime.h>
__device__ __noinline__ double helper(double x, int n) { for (int i = 0; i < n; ++i) x = x * x + 1.0; return x; }
__global__ void calls(double *p, int n) { p[threadIdx.x] = helper(p[threadIdx.x], n) + helper(p[threadIdx.x] + 1.0, n + 1); }
__global__ void clean(double *p) { p[threadIdx.x] += 1.0; }
""")
    obj = tmp_path / "calls.o"
    subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-c", "-o", str(obj), str(src)], check=True, capture_output=True)
    r = subprocess.run([sys.executable, CHECK, str(obj)], capture_output=True, text=True)
    assert r.returncode == 1, r.stdout
    assert "device function emitted out of line" in r.stdout and "helper" in r.stdout
    assert "call instruction(s)" in r.stdout and "clean" not in r.stdout.split("VIOLATION")[1]


def test_benchmark_kernels_have_the_recorded_machine_code():
    """The flagship kernel loses or gains a per cent through source changes it never executes (tools/isa_digest.py): the machine code
    of the shipped benchmark kernels is recorded with the round's measurements, and a build whose code moved fails here until it has
    been measured again and the record rewritten (python3 tools/isa_digest.py --write)."""
    import json
    objs = [os.path.join(CSRC, "build", f) for f in ("fs_part_nodiag.o", "fs_part_team.o")]
    if not all(os.path.exists(o) for o in objs):
        pytest.skip("no build directory next to the library")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_digest
    want = json.load(open(isa_digest.RECORD))
    got = {}
    for o in objs:
        got.update(isa_digest.digests(o))
    moved = sorted(k for k in want if got.get(k, {}).get("md5") != want[k]["md5"])
    assert not moved and set(got) == set(want), ("kernels whose machine code differs from profiles/round4/isa_digests.json", moved, sorted(set(got) ^ set(want)))
