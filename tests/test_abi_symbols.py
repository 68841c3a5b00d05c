"""The C-ABI library loads on a machine without a GPU and exports every function that
include/flowsim_abi.h declares; without a device it refuses to compute instead of falling back."""
import os
import re

import pytest

from conftest import ROOT
from flowsim_amd import _abi as A


def declared_functions():
    text = open(os.path.join(ROOT, "include", "flowsim_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fs_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = A.lib()
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), n
        assert n in A.SIGNATURES, f"{n} missing from the ctypes signature table"
    assert sorted(A.SIGNATURES) == names
    assert lib.fs_abi_version() == A.ABI_VERSION


def test_no_cpu_fallback_without_device():
    if A.device_count() > 0:
        pytest.skip("a GPU is visible")
    from flowsim_amd import PreissmannBatch
    with pytest.raises(A.FlowsimError, match="no HIP device"):
        PreissmannBatch(1, 30, 21)


def test_product_path_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "flow-sim_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dirpath, f)
                assert "preissmann_oracle" not in src, os.path.join(dirpath, f)
