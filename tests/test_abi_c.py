"""include/flowsim_abi.h used from plain C: compiles as C99 (-Wall -Wextra -pedantic), links against
libflowsim_hip.so and, on a GPU box, steps a small reach.  On a CPU box the same program checks that
fs_batch_create refuses to run (exit code 2)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT
from flowsim_amd import _abi as A

SRC = os.path.join(ROOT, "tests", "c_abi", "smoke.c")


def build(tmp_path):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    A.lib()                                             # raises if the library was not built
    libdir = os.path.dirname(os.path.abspath(A.LIB_PATH))
    exe = str(tmp_path / "abi_smoke")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), SRC,
           "-L", libdir, "-lflowsim_hip", "-lm", "-Wl,-rpath," + libdir, "-o", exe]
    subprocess.run(cmd, check=True)
    return exe


def test_header_is_c99_and_library_links(tmp_path):
    exe = build(tmp_path)
    if A.device_count() > 0:
        pytest.skip("covered by the gpu test")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2, (r.returncode, r.stdout, r.stderr)
    assert "no HIP device" in r.stdout


@pytest.mark.gpu
def test_c_program_steps_a_reach(tmp_path):
    exe = build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert r.stdout.startswith("ok:")
