"""ctypes front end of oracle/preissmann_oracle.c (TEST INFRASTRUCTURE ONLY).

`run(problem)` takes the same `Problem` as oracle/preissmann_oracle.py and returns the same dict.
Used by tests for larger parity cases (it is ~100x faster than the numpy oracle at small N) and by
bench.py as the compiled single-core CPU baseline."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import preissmann_oracle as O

_HERE = os.path.dirname(os.path.abspath(__file__))
_SAN = os.environ.get("FS_ORACLE_SAN") == "1"          # the ASan + UBSan build (make SAN=1; tests/test_sanitizers.py)
_SO = os.path.join(_HERE, "_build", "liboracle_c_san.so" if _SAN else "liboracle_c.so")
_D = C.POINTER(C.c_double)
_KIND = {"flow_hydrograph": 0, "stage_hydrograph": 1, "normal_depth": 3}


class _BC(C.Structure):
    _fields_ = [("kind", C.c_int), ("p", C.c_double * 10), ("target", _D)]


class _Problem(C.Structure):
    _fields_ = [("N", C.c_int), ("nt", C.c_int), ("max_iter", C.c_int), ("theta", C.c_double), ("dt", C.c_double),
                ("dx", C.c_double), ("tol", C.c_double)] + [(k, _D) for k in O.GEO_KEYS] + \
               [("h0", _D), ("Q0", _D), ("us", _BC), ("ds", _BC)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.run(["make", "-C", _HERE] + (["SAN=1"] if _SAN else []), check=True, capture_output=True)
        _lib = C.CDLL(_SO)
        _lib.fso_run.restype = C.c_int
        _lib.fso_run.argtypes = [C.POINTER(_Problem), _D, _D, C.POINTER(C.c_int), _D]
    return _lib


def _bc(bc: O.BC, keep):
    out = _BC()
    p = []
    if bc.kind in _KIND:
        out.kind = _KIND[bc.kind]
        p = {"flow_hydrograph": [], "stage_hydrograph": [bc.bed_level], "normal_depth": [bc.bed_slope, bc.bed_level]}[bc.kind]
    elif bc.kind == "fixed_depth" and bc.storage is None:
        out.kind, p = 2, [bc.initial_depth]
    elif bc.kind == "fixed_depth":
        s = bc.storage
        out.kind, p = 7, [s["area"], s["min_stage"], s["Y_min"], s["Y_max"], bc.bed_level]
    elif bc.kind == "rating_curve":
        rc = bc.rc
        if bc.rc_type == "power":
            out.kind, p = 4, [rc["a"], rc["b"], rc.get("shift", 0.0), bc.bed_level]
        elif bc.rc_type == "polynomial":
            out.kind, p = 5, [rc["a"], rc["b"], rc["c"], rc.get("shift", 0.0), bc.bed_level]
        else:
            out.kind, p = 6, [rc["initial_stage"], rc["buffer"], *rc["low"], *rc["high"], rc.get("dY", 1e-3), bc.bed_level]
    else:
        raise ValueError(bc.kind)
    for i, v in enumerate(p):
        out.p[i] = float(v)
    if bc.target is not None:
        t = np.ascontiguousarray(bc.target, dtype=np.float64)
        keep.append(t)
        out.target = t.ctypes.data_as(_D)
    return out


def run(p: O.Problem):
    keep = []
    cp = _Problem()
    cp.N, cp.nt, cp.max_iter = p.N, p.nt, p.max_iter
    cp.theta, cp.dt, cp.dx, cp.tol = p.theta, p.dt, p.dx, p.tol
    for k in O.GEO_KEYS:
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(p.geo[k], dtype=np.float64), (p.N,)))
        keep.append(a)
        setattr(cp, k, a.ctypes.data_as(_D))
    h0 = np.ascontiguousarray(p.h0, dtype=np.float64); Q0 = np.ascontiguousarray(p.Q0, dtype=np.float64)
    cp.h0, cp.Q0 = h0.ctypes.data_as(_D), Q0.ctypes.data_as(_D)
    cp.us, cp.ds = _bc(p.us, keep), _bc(p.ds, keep)
    depth = np.zeros((p.nt, p.N)); flow = np.zeros((p.nt, p.N)); iters = np.zeros(p.nt, dtype=np.int32)
    stage = np.zeros(p.nt)
    st = lib().fso_run(C.byref(cp), depth.ctypes.data_as(_D), flow.ctypes.data_as(_D),
                       iters.ctypes.data_as(C.POINTER(C.c_int)), stage.ctypes.data_as(_D))
    return dict(depth=depth, flow=flow, iters=iters, status=st, storage_stage=stage[1:])
