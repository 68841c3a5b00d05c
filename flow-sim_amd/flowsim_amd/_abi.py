"""ctypes binding of include/flowsim_abi.h.

The HIP library is the product; there is no Python or CPU fallback.  If libflowsim_hip.so has not
been built (`python -c "import __graft_entry__ as g; g.build()"`) or no MI355X is visible, the
calls below raise - loudly - instead of computing anything on the host.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "..", "csrc", "libflowsim_hip.so")

F64, F32 = 0, 1
SEC_RECT_UNIFORM, SEC_TRAP_UNIFORM, SEC_TABLE, SEC_IRREGULAR = 0, 1, 2, 3
RU_WIDTH, RU_MANNING, RU_Z_US, RU_Z_DS, RU_NPARAM = 0, 1, 2, 3, 4
TU_SIDE_SLOPE, TU_NPARAM = 4, 5
BC_MAX_PARAMS = 10
GEO_ROWS = ("z_bed", "b_main", "m_main", "n_main", "n_left", "n_right", "is_compound", "h_bf",
            "b_fp_l", "b_fp_r", "m_fp", "curvature")
GEO_NPARAM = len(GEO_ROWS)
(BC_FLOW_HYDROGRAPH, BC_STAGE_HYDROGRAPH, BC_FIXED_DEPTH, BC_NORMAL_DEPTH, BC_RATING_POWER,
 BC_RATING_POLY, BC_RATING_BLEND, BC_STORAGE, BC_STORAGE_CURVE, BC_HOST_ROW) = range(10)
# scalar slots of BC_STORAGE_CURVE (FS_SC_* of the header), followed by stage[n_curve], area[n_curve]
SC_NAMES = ("min_stage", "Y_min", "Y_max", "bed_level", "surface_area", "alpha", "beta", "n_curve", "rc_type", "rc_a",
            "rc_b", "rc_c", "rc_shift", "capture_losses", "reservoir_length", "K_q")
UPSTREAM, DOWNSTREAM = 0, 1
OK, MAX_ITER, NAN, STORAGE_RANGE, ILL_CONDITIONED, TEAM_STALL = 0, 1, 2, 3, 4, 5
FLAG_HISTORY, FLAG_TRACE, FLAG_MONITOR = 1, 2, 4
TRACE_CAP = 64
DERIVE_ALL = 255
ABI_VERSION = 3


class BatchDesc(C.Structure):
    _fields_ = [("n_reaches", C.c_int32), ("n_nodes", C.c_int32), ("dtype", C.c_int32),
                ("section_mode", C.c_int32), ("device", C.c_int32), ("max_levels", C.c_int32),
                ("flags", C.c_int32), ("reserved", C.c_int32)]


_P = C.c_void_p
_D = C.POINTER(C.c_double)
_I = C.POINTER(C.c_int32)

# name -> (restype, argtypes); must list every function declared in include/flowsim_abi.h
SIGNATURES = {
    "fs_abi_version": (C.c_int, []),
    "fs_device_count": (C.c_int, []),
    "fs_last_error": (C.c_char_p, []),
    "fs_batch_create": (_P, [C.POINTER(BatchDesc)]),
    "fs_batch_destroy": (None, [_P]),
    "fs_batch_set_scheme": (C.c_int, [_P, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32]),
    "fs_batch_set_geometry_uniform": (C.c_int, [_P, _D]),
    "fs_batch_set_geometry_table": (C.c_int, [_P, _D, _D]),
    "fs_batch_set_geometry_irregular": (C.c_int, [_P, _D, _I, C.c_int32, _D, _D, _D, _D]),
    "fs_batch_set_bc": (C.c_int, [_P, C.c_int32, C.c_int32, _D, C.c_int32, C.c_int32, _D]),
    "fs_batch_set_geometry_table_per_reach": (C.c_int, [_P, _D, _D]),
    "fs_batch_set_geometry_irregular_per_reach": (C.c_int, [_P, _D, _I, C.c_int32, _D, _D, _D, _D]),
    "fs_batch_set_reach_nodes": (C.c_int, [_P, _I]),
    "fs_batch_set_reach_scheme": (C.c_int, [_P, _D, _D, _D]),
    "fs_batch_set_reach_tolerance": (C.c_int, [_P, _D, _I]),
    "fs_batch_set_bc_per_reach": (C.c_int, [_P, C.c_int32, _I, _D, _D]),
    "fs_batch_set_bc_per_reach_wide": (C.c_int, [_P, C.c_int32, _I, _D, C.c_int32, _D]),
    "fs_batch_set_state": (C.c_int, [_P, _D, _D]),
    "fs_batch_set_state_uniform": (C.c_int, [_P, _D, _D]),
    "fs_batch_step": (C.c_int, [_P, C.c_int32]),
    "fs_batch_sync": (C.c_int, [_P]),
    "fs_batch_iterate": (C.c_int, [_P, _I]),
    "fs_batch_set_host_rows": (C.c_int, [_P, C.c_int32, _D]),
    "fs_batch_get_boundary_iterate": (C.c_int, [_P, _D]),
    "fs_batch_restart": (C.c_int, [_P, C.c_int32, _D, _D, _D, _D, _D]),
    "fs_batch_level": (C.c_int32, [_P]),
    "fs_batch_get_state": (C.c_int, [_P, _D, _D]),
    "fs_batch_get_guess": (C.c_int, [_P, _D, _D]),
    "fs_batch_get_hydrographs": (C.c_int, [_P, C.c_int32, C.c_int32, _D]),
    "fs_batch_get_iterations": (C.c_int, [_P, C.c_int32, C.c_int32, _I]),
    "fs_batch_get_status": (C.c_int, [_P, _I]),
    "fs_batch_get_history": (C.c_int, [_P, C.c_int32, C.c_int32, _D, _D]),
    "fs_batch_get_residual_trace": (C.c_int, [_P, C.c_int32, C.c_int32, _D]),
    "fs_batch_get_storage_stage": (C.c_int, [_P, _D]),
    "fs_batch_get_storage_stages": (C.c_int, [_P, C.c_int32, C.c_int32, _D]),
    "fs_batch_derive": (C.c_int, [_P, C.c_int32, C.c_int32, _D, _D, _D, _D, _D, _D, _D, _D]),
    "fs_batch_derive_device": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32]),
    "fs_batch_derived_device_ptr": (_P, [_P, C.c_int32]),
    "fs_batch_hydrograph_device_ptr": (_P, [_P]),
    "fs_batch_stream": (_P, [_P]),
    "fs_batch_last_step_ms": (C.c_double, [_P]),
    "fs_batch_last_launch_count": (C.c_int32, [_P]),
    "fs_batch_kernel_info": (C.c_int, [_P, _I, _I, _I, _I]),
    "fs_kernel_table_size": (C.c_int32, []),
    "fs_kernel_table_entry": (C.c_int, [C.c_int32, _I]),
    "fs_batch_kernel_index": (C.c_int32, [_P]),
    "fs_batch_poly_tables": (C.c_int32, [_P]),
    "fs_kernel_table_entry_tail": (C.c_int32, [C.c_int32]),
    "fs_kernel_table_entry_team": (C.c_int32, [C.c_int32]),
}

_lib = None


class FlowsimError(RuntimeError):
    pass


def lib():
    """Loads libflowsim_hip.so (once).  Raises FlowsimError if it has not been built."""
    global _lib
    if _lib is None:
        path = os.path.normpath(os.environ.get("FS_LIB", LIB_PATH))     # FS_LIB: experiment builds
        if not os.path.exists(path):
            raise FlowsimError(
                f"{path} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; "
                "g.build()' or make -C flow-sim_amd/csrc).  There is no CPU fallback.")
        l = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        if l.fs_abi_version() != ABI_VERSION:
            raise FlowsimError(f"ABI mismatch: library {l.fs_abi_version()}, binding {ABI_VERSION}")
        _lib = l
    return _lib


def last_error():
    return lib().fs_last_error().decode("utf-8", "replace")


def check(rc, what=""):
    if rc != 0:
        raise FlowsimError(f"{what}: {last_error()}" if what else last_error())


def device_count():
    return lib().fs_device_count()


KERNEL_FIELDS = ("dtype", "section_mode", "cells_per_thread", "waves_per_reach", "full", "boundary_class", "diag", "long_reach")


def kernel_table():
    """The dispatch table of the step kernel: one dict per instantiation (fs_kernel_table_entry)."""
    l = lib()
    out = []
    for i in range(l.fs_kernel_table_size()):
        v = (C.c_int32 * 8)()
        check(l.fs_kernel_table_entry(i, v), "kernel_table")
        out.append(dict(zip(KERNEL_FIELDS, (int(x) for x in v)), index=i, tail=int(l.fs_kernel_table_entry_tail(i)), team=int(l.fs_kernel_table_entry_team(i))))
    return out
