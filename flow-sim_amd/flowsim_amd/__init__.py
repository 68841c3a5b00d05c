"""flowsim_amd - MI355X-native batched Preissmann stepper: Python host side.

`flowsim_amd._abi`   ctypes binding of the C ABI (include/flowsim_abi.h, libflowsim_hip.so)
`flowsim_amd.batch`  PreissmannBatch: many independent reaches stepped on one GPU
`flowsim_amd.hydromodel`  the reference's Channel / Boundary / ... / PreissmannSolver interface
"""
from . import _abi  # noqa: F401
from .batch import BoundarySpec, PreissmannBatch  # noqa: F401

__all__ = ["PreissmannBatch", "BoundarySpec"]
