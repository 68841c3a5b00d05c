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
