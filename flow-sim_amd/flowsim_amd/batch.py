"""PreissmannBatch - B independent reaches advanced together on one MI355X through the C ABI.

Host-side mirror of the state the reference's PreissmannSolver carries (preissmann.py:23-59,
solver.py:11-51): scheme parameters, node geometry, two boundaries, initial conditions; `step(n)`
is the batched equivalent of n passes of the outer loop of PreissmannSolver.run
(preissmann.py:108-161).  All arithmetic happens in the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

from . import _abi as A

_KIND_PARAMS = {
    A.BC_FLOW_HYDROGRAPH: (), A.BC_STAGE_HYDROGRAPH: ("bed_level",), A.BC_FIXED_DEPTH: ("initial_depth",),
    A.BC_NORMAL_DEPTH: ("bed_slope", "bed_level"), A.BC_RATING_POWER: ("a", "b", "stage_shift", "bed_level"),
    A.BC_RATING_POLY: ("a", "b", "c", "stage_shift", "bed_level"),
    A.BC_RATING_BLEND: ("stage0", "buffer", "lo0", "lo1", "lo2", "hi0", "hi1", "hi2", "dY", "bed_level"),
    A.BC_STORAGE: ("surface_area", "min_stage", "Y_min", "Y_max", "bed_level"),
}


@dataclass
class BoundarySpec:
    """One boundary of every reach in the batch (Boundary, boundary.py:10-46, flattened).

    params: dict name -> scalar (shared by all reaches) or array [B] (per reach); the names per
    kind are in _KIND_PARAMS.  target: hydrograph pre-sampled at level*dt, [n_levels] (shared) or
    [n_levels, B]."""
    kind: int
    params: dict = field(default_factory=dict)
    target: Optional[np.ndarray] = None


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class PreissmannBatch:
    def __init__(self, n_reaches: int, n_nodes: int, max_levels: int, dtype: str = "f64",
                 section_mode: str = "rect_uniform", device: int = 0, history: bool = False, trace: bool = False,
                 monitor: bool = True):
        """monitor: the conditioning monitor of the on-chip elimination (status ILL_CONDITIONED, the stand-in for the reference's
        `diagnos` check, preissmann.py:133-144) runs on every batch unless the caller opts out - a batch without history or
        trace then runs the step kernels compiled without diagnostics (1 - 3 % faster on the benchmark shapes; bench.py does)."""
        self.B, self.N, self.L = int(n_reaches), int(n_nodes), int(max_levels)
        self.dtype = {"f64": A.F64, "f32": A.F32}[dtype]
        self.mode = {"rect_uniform": A.SEC_RECT_UNIFORM, "trap_uniform": A.SEC_TRAP_UNIFORM,
                     "table": A.SEC_TABLE, "irregular": A.SEC_IRREGULAR}[section_mode]
        self._lib = A.lib()
        desc = A.BatchDesc(self.B, self.N, self.dtype, self.mode, device, self.L,
                           (A.FLAG_HISTORY if history else 0) | (A.FLAG_TRACE if trace else 0) | (A.FLAG_MONITOR if monitor else 0), 0)
        self._h = self._lib.fs_batch_create(C.byref(desc))
        if not self._h:
            raise A.FlowsimError(A.last_error())
        self.history = history

    # -- lifetime -------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.fs_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- setup ----------------------------------------------------------------------------
    def set_scheme(self, theta, dt, dx, tolerance=1e-4, max_iter=100):
        A.check(self._lib.fs_batch_set_scheme(self._h, float(theta), float(dt), float(dx), float(tolerance),
                                              int(max_iter)), "set_scheme")

    def set_geometry_uniform(self, width, manning, z_us, z_ds, side_slope=None):
        trap = self.mode == A.SEC_TRAP_UNIFORM
        p = np.empty((A.TU_NPARAM if trap else A.RU_NPARAM, self.B), dtype=np.float64)
        p[A.RU_WIDTH], p[A.RU_MANNING], p[A.RU_Z_US], p[A.RU_Z_DS] = width, manning, z_us, z_ds
        if trap:
            p[A.TU_SIDE_SLOPE] = 0.0 if side_slope is None else side_slope
        A.check(self._lib.fs_batch_set_geometry_uniform(self._h, _dptr(p)), "set_geometry_uniform")

    def set_geometry_table(self, geo: dict, n_main_override: Optional[Sequence[float]] = None):
        """geo[k]: [N] (one channel shared by the batch) or [B, N] (one channel per reach: geometry ensembles, different
        rivers; rows of a shorter reach padded by repeating its last node, see set_reach_nodes)"""
        per_reach = any(np.ndim(geo[k]) == 2 for k in A.GEO_ROWS)
        tab = np.empty((self.B, A.GEO_NPARAM, self.N) if per_reach else (A.GEO_NPARAM, self.N), dtype=np.float64)
        for i, k in enumerate(A.GEO_ROWS):
            if per_reach:
                tab[:, i, :] = np.broadcast_to(np.asarray(geo[k], dtype=np.float64), (self.B, self.N))
            else:
                tab[i] = np.asarray(geo[k], dtype=np.float64)
        ov = None
        if n_main_override is not None:
            ov = np.ascontiguousarray(n_main_override, dtype=np.float64)
            assert ov.shape == (self.B,)
        fn = self._lib.fs_batch_set_geometry_table_per_reach if per_reach else self._lib.fs_batch_set_geometry_table
        A.check(fn(self._h, _dptr(tab), _dptr(ov) if ov is not None else None), "set_geometry_table")

    def set_reach_nodes(self, n_nodes):
        """per-reach node counts (<= N of the batch): each reach its own channel length / grid (solver.py:34-38, :53-55)"""
        n = np.ascontiguousarray(n_nodes, dtype=np.int32)
        assert n.shape == (self.B,)
        A.check(self._lib.fs_batch_set_reach_nodes(self._h, n.ctypes.data_as(A._I)), "set_reach_nodes")

    def set_reach_scheme(self, theta=None, dt=None, dx=None):
        """per-reach theta / time step / spatial step (arrays [B]; None: the batch-wide value of set_scheme)"""
        arrs = [None if v is None else np.ascontiguousarray(np.broadcast_to(np.asarray(v, dtype=np.float64), (self.B,))) for v in (theta, dt, dx)]
        A.check(self._lib.fs_batch_set_reach_scheme(self._h, *[None if a is None else _dptr(a) for a in arrs]), "set_reach_scheme")

    def set_reach_tolerance(self, tolerance=None, max_iter=None):
        """per-reach convergence tolerance and iteration cap (arrays [B]; None: the batch-wide value of set_scheme): what each
        run(tolerance=, max_iter=) of the reference has of its own (preissmann.py:101)"""
        tol = None if tolerance is None else np.ascontiguousarray(np.broadcast_to(np.asarray(tolerance, dtype=np.float64), (self.B,)))
        mit = None if max_iter is None else np.ascontiguousarray(np.broadcast_to(np.asarray(max_iter, dtype=np.int32), (self.B,)))
        A.check(self._lib.fs_batch_set_reach_tolerance(self._h, None if tol is None else _dptr(tol),
                                                       None if mit is None else mit.ctypes.data_as(A._I)), "set_reach_tolerance")

    def set_boundary_per_reach(self, side: int, specs):
        """one BoundarySpec per reach, kinds free to differ: the closed-form kinds BC_FLOW_HYDROGRAPH .. BC_STORAGE and, in table /
        polyline batches, BC_STORAGE_CURVE (a general reservoir behind some of the channels, each with its own scalars and an area
        curve of its own length) and BC_HOST_ROW on the reaches whose plugin has no device form (such a batch advances with
        iterate(), and set_host_rows() writes the rows of those reaches only); a spec's target is [n_levels] (sampled at that
        reach's own k * dt)"""
        assert len(specs) == self.B
        kinds = np.array([s.kind for s in specs], dtype=np.int32)
        curves = {r: np.asarray(s.params.get("curve", np.empty((0, 2))), dtype=np.float64).reshape(-1, 2)
                  for r, s in enumerate(specs) if s.kind == A.BC_STORAGE_CURVE}
        rows = max([A.BC_MAX_PARAMS] + [len(A.SC_NAMES) + 2 * len(c) for c in curves.values()])
        p = np.zeros((rows, self.B), dtype=np.float64)
        tgt = None
        for r, s in enumerate(specs):
            if s.kind == A.BC_STORAGE_CURVE:       # the FS_SC_* scalars (missing ones 0, alpha 1), then stages, then areas
                sc = {"alpha": 1.0}
                sc.update({k: v for k, v in s.params.items() if k != "curve"})
                sc["n_curve"] = float(len(curves[r]))
                nf, nc = len(A.SC_NAMES), len(curves[r])
                p[:nf, r] = [float(sc.get(k) or 0.0) for k in A.SC_NAMES]
                p[nf:nf + nc, r] = curves[r][:, 0]
                p[nf + nc:nf + 2 * nc, r] = curves[r][:, 1]
            else:
                for i, name in enumerate(_KIND_PARAMS.get(s.kind, ())):
                    p[i, r] = float(s.params[name])
            if s.target is not None:
                if tgt is None:
                    tgt = np.zeros((self.L, self.B), dtype=np.float64)
                t = np.asarray(s.target, dtype=np.float64).reshape(-1)
                n = min(self.L, t.shape[0])
                tgt[:n, r] = t[:n]
                tgt[n:, r] = t[n - 1]
        A.check(self._lib.fs_batch_set_bc_per_reach_wide(self._h, side, kinds.ctypes.data_as(A._I), _dptr(p), rows,
                                                         _dptr(tgt) if tgt is not None else None), "set_boundary_per_reach")

    def set_geometry_irregular(self, geo: dict, n_main_override: Optional[Sequence[float]] = None):
        """geo: the TABLE rows plus irr_x / irr_z [N, P] (rows padded beyond irr_npts), irr_npts [N]
        (0 = trapezoid-family node) and irr_limits [N, 2] (IrregularSection.left/right_fp_limit)."""
        per_reach = np.ndim(geo["irr_npts"]) == 2          # one channel per reach: irr_npts [B, N], irr_x / irr_z [B, N, P], irr_limits [B, N, 2]
        lead = (self.B,) if per_reach else ()
        tab = np.empty(lead + (A.GEO_NPARAM, self.N), dtype=np.float64)
        for i, k in enumerate(A.GEO_ROWS):
            tab[..., i, :] = np.asarray(geo[k], dtype=np.float64)
        cnt = np.ascontiguousarray(geo["irr_npts"], dtype=np.int32)
        x = np.ascontiguousarray(geo["irr_x"], dtype=np.float64)
        z = np.ascontiguousarray(geo["irr_z"], dtype=np.float64)
        lim = np.ascontiguousarray(geo["irr_limits"], dtype=np.float64)
        assert cnt.shape == lead + (self.N,) and x.shape == z.shape and x.shape[:-1] == lead + (self.N,) and lim.shape == lead + (self.N, 2)
        ov = None
        if n_main_override is not None:
            ov = np.ascontiguousarray(n_main_override, dtype=np.float64)
            assert ov.shape == (self.B,)
        fn = self._lib.fs_batch_set_geometry_irregular_per_reach if per_reach else self._lib.fs_batch_set_geometry_irregular
        A.check(fn(self._h, _dptr(tab), cnt.ctypes.data_as(A._I), x.shape[-1], _dptr(x), _dptr(z), _dptr(lim),
                   _dptr(ov) if ov is not None else None), "set_geometry_irregular")

    def set_boundary(self, side: int, spec: BoundarySpec):
        if spec.kind == A.BC_HOST_ROW:
            # evaluated by the caller before every Newton iteration (set_host_rows / iterate)
            A.check(self._lib.fs_batch_set_bc(self._h, side, spec.kind, None, 3, 1, None), "set_boundary")
            return
        if spec.kind == A.BC_STORAGE_CURVE:
            # general LumpedStorage: the FS_SC_* scalars (missing ones default to 0, alpha to 1) + the area curve
            curve = np.asarray(spec.params.get("curve", np.empty((0, 2))), dtype=np.float64)
            sc = {"alpha": 1.0}
            sc.update({k: v for k, v in spec.params.items() if k != "curve"})
            per_reach = curve.ndim == 3 or any(np.ndim(v) > 0 for v in sc.values() if v is not None)
            if per_reach:       # one reservoir per reach: scalars [B] (or shared), curve [B, n_curve, 2] (or shared [n_curve, 2])
                curve = np.broadcast_to(curve.reshape((-1,) + curve.shape[-2:]) if curve.size else np.empty((1, 0, 2)), (self.B,) + curve.shape[-2:])
                nc = curve.shape[1]
                sc["n_curve"] = float(nc)
                p = np.empty((len(A.SC_NAMES) + 2 * nc, self.B), dtype=np.float64)
                for i, k in enumerate(A.SC_NAMES):
                    v = sc.get(k)
                    p[i] = 0.0 if v is None else v
                p[len(A.SC_NAMES):len(A.SC_NAMES) + nc] = curve[:, :, 0].T
                p[len(A.SC_NAMES) + nc:] = curve[:, :, 1].T
                A.check(self._lib.fs_batch_set_bc(self._h, side, spec.kind, _dptr(p), p.shape[0], 1, None), "set_boundary")
                return
            curve = curve.reshape(-1, 2)
            sc["n_curve"] = float(len(curve))
            p = np.concatenate([[float(sc.get(k) or 0.0) for k in A.SC_NAMES], curve[:, 0], curve[:, 1]])
            A.check(self._lib.fs_batch_set_bc(self._h, side, spec.kind, _dptr(p), len(p), 0, None), "set_boundary")
            return
        names = _KIND_PARAMS[spec.kind]
        vals = [np.asarray(spec.params[n], dtype=np.float64) for n in names]
        per_reach = any(v.ndim > 0 for v in vals)
        if per_reach:
            p = np.empty((len(names), self.B), dtype=np.float64)
            for i, v in enumerate(vals):
                p[i] = v
        else:
            p = np.array([float(v) for v in vals], dtype=np.float64)
        tgt = None
        if spec.target is not None:
            t = np.asarray(spec.target, dtype=np.float64)
            tgt = np.empty((self.L, self.B), dtype=np.float64)
            n = min(self.L, t.shape[0])
            tgt[:n] = t[:n, None] if t.ndim == 1 else t[:n]
            tgt[n:] = tgt[n - 1]
        A.check(self._lib.fs_batch_set_bc(self._h, side, spec.kind, _dptr(p) if len(names) else None, len(names),
                                          1 if per_reach else 0, _dptr(tgt) if tgt is not None else None),
                "set_boundary")

    def set_state(self, h, Q):
        h = np.ascontiguousarray(np.broadcast_to(np.asarray(h, dtype=np.float64), (self.B, self.N)))
        Q = np.ascontiguousarray(np.broadcast_to(np.asarray(Q, dtype=np.float64), (self.B, self.N)))
        A.check(self._lib.fs_batch_set_state(self._h, _dptr(h), _dptr(Q)), "set_state")

    def set_state_uniform(self, h, Q):
        """Steady-state IC of prismatic reaches: one depth and one flow per reach (channel.py:296-305)."""
        h = np.ascontiguousarray(np.broadcast_to(np.asarray(h, dtype=np.float64), (self.B,)))
        Q = np.ascontiguousarray(np.broadcast_to(np.asarray(Q, dtype=np.float64), (self.B,)))
        A.check(self._lib.fs_batch_set_state_uniform(self._h, _dptr(h), _dptr(Q)), "set_state_uniform")

    # -- the hot path ---------------------------------------------------------------------
    def step(self, n_steps: int = 1, sync: bool = True):
        A.check(self._lib.fs_batch_step(self._h, int(n_steps)), "step")
        if sync:
            self.sync()

    def sync(self):
        A.check(self._lib.fs_batch_sync(self._h), "sync")

    def iterate(self) -> int:
        """One Newton iteration of level `level + 1` for every reach still iterating on it; returns how many
        remain open (0: the level is complete and `level` has advanced)."""
        n = C.c_int32()
        A.check(self._lib.fs_batch_iterate(self._h, C.byref(n)), "iterate")
        return n.value

    def set_host_rows(self, side: int, dh, dq, res):
        """Boundary row (d/dh, d/dQ, residual) of every reach at the current Newton vector (BC_HOST_ROW sides; with per-reach
        kinds the entries of the reaches whose boundary the device evaluates are ignored)."""
        rows = np.empty((3, self.B), dtype=np.float64)
        rows[0], rows[1], rows[2] = dh, dq, res
        A.check(self._lib.fs_batch_set_host_rows(self._h, side, _dptr(rows)), "set_host_rows")

    def boundary_iterate(self):
        """[4, B]: h[0], Q[0], h[N-1], Q[N-1] of the current Newton vector."""
        out = np.empty((4, self.B))
        A.check(self._lib.fs_batch_get_boundary_iterate(self._h, _dptr(out)), "get_boundary_iterate")
        return out

    def restart(self, level: int, h, Q, h_guess, Q_guess, storage_stage=None):
        """Continue bit-exactly from `level`: state, Newton start vector of level + 1 and reservoir stage."""
        arrs = [np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (self.B, self.N)))
                for a in (h, Q, h_guess, Q_guess)]
        st = None if storage_stage is None else np.ascontiguousarray(np.broadcast_to(np.asarray(storage_stage, dtype=np.float64), (self.B,)))
        A.check(self._lib.fs_batch_restart(self._h, int(level), *[_dptr(a) for a in arrs], _dptr(st) if st is not None else None),
                "restart")

    @property
    def level(self):
        return self._lib.fs_batch_level(self._h)

    # -- results --------------------------------------------------------------------------
    def state(self, out=None):
        """depth[k], flow[k] of the current level, [B, N] each; out=(h, Q): filled in place (a caller that downloads every few
        levels reuses its buffers instead of paying the page faults of fresh ones each time)"""
        h, Q = out if out is not None else (np.empty((self.B, self.N)), np.empty((self.B, self.N)))
        assert h.shape == Q.shape == (self.B, self.N) and h.dtype == Q.dtype == np.float64 and h.flags.c_contiguous and Q.flags.c_contiguous
        A.check(self._lib.fs_batch_get_state(self._h, _dptr(h), _dptr(Q)), "get_state")
        return h, Q

    def guess(self):
        h = np.empty((self.B, self.N)); Q = np.empty((self.B, self.N))
        A.check(self._lib.fs_batch_get_guess(self._h, _dptr(h), _dptr(Q)), "get_guess")
        return h, Q

    def hydrographs(self, first=0, n=None):
        """[n, 4, B]: depth[k,0], flow[k,0], depth[k,-1], flow[k,-1]."""
        n = self.level + 1 - first if n is None else n
        out = np.empty((n, 4, self.B))
        A.check(self._lib.fs_batch_get_hydrographs(self._h, first, n, _dptr(out)), "get_hydrographs")
        return out

    def iterations(self, first=0, n=None):
        n = self.level + 1 - first if n is None else n
        out = np.empty((n, self.B), dtype=np.int32)
        A.check(self._lib.fs_batch_get_iterations(self._h, first, n, out.ctypes.data_as(C.POINTER(C.c_int32))),
                "get_iterations")
        return out

    def status(self):
        out = np.empty(self.B, dtype=np.int32)
        A.check(self._lib.fs_batch_get_status(self._h, out.ctypes.data_as(C.POINTER(C.c_int32))), "get_status")
        return out

    def history_arrays(self, first=0, n=None):
        """depth, flow [n, B, N] (needs history=True)."""
        n = self.level + 1 - first if n is None else n
        h = np.empty((n, self.B, self.N)); Q = np.empty((n, self.B, self.N))
        A.check(self._lib.fs_batch_get_history(self._h, first, n, _dptr(h), _dptr(Q)), "get_history")
        return h, Q

    def storage_stage(self):
        out = np.empty(self.B)
        A.check(self._lib.fs_batch_get_storage_stage(self._h, _dptr(out)), "get_storage_stage")
        return out

    def residual_trace(self, first=0, n=None):
        """[n, TRACE_CAP, B] ||R|| per Newton iteration (needs trace=True); zeros beyond a level's count."""
        n = self.level + 1 - first if n is None else n
        out = np.empty((n, A.TRACE_CAP, self.B))
        A.check(self._lib.fs_batch_get_residual_trace(self._h, first, n, _dptr(out)), "get_residual_trace")
        return out

    def storage_stages(self, first=0, n=None):
        """[n, B] reservoir stage per time level (storage boundary only)."""
        n = self.level + 1 - first if n is None else n
        out = np.empty((n, self.B))
        A.check(self._lib.fs_batch_get_storage_stages(self._h, first, n, _dptr(out)), "get_storage_stages")
        return out

    DERIVED = ("level", "area", "top_width", "froude_number", "velocity", "wave_celerity", "amplitude")

    def derive(self, first=0, n=None, fields=None):
        """Solver.prepare_results on the device (needs history=True): dict name -> [n, B, N] plus
        peak_amplitude [B, N]."""
        n = self.level + 1 - first if n is None else n
        want = set(self.DERIVED + ("peak_amplitude",)) if fields is None else set(fields)
        out = {k: np.empty((n, self.B, self.N)) for k in self.DERIVED if k in want}
        if "peak_amplitude" in want:
            out["peak_amplitude"] = np.empty((self.B, self.N))
        ptr = lambda k: _dptr(out[k]) if k in out else None
        A.check(self._lib.fs_batch_derive(self._h, first, n, *[ptr(k) for k in self.DERIVED], ptr("peak_amplitude")),
                "derive")
        return out

    def last_step_ms(self):
        return self._lib.fs_batch_last_step_ms(self._h)

    def kernel_info(self):
        v = [C.c_int32() for _ in range(4)]
        A.check(self._lib.fs_batch_kernel_info(self._h, *[C.byref(x) for x in v]), "kernel_info")
        return dict(cells_per_thread=v[0].value, waves_per_reach=v[1].value, lds_bytes=v[2].value, vgprs=v[3].value)

    def poly_tables(self):
        """polyline batches: 1 = stage tables, 0 = edge walk (tables beyond FS_POLY_TABLE_MAX_BYTES or FS_POLY_WALK=1), -1 = none set"""
        return self._lib.fs_batch_poly_tables(self._h)

    def kernel_index(self):
        """index into _abi.kernel_table() of the instantiation the last step / iterate launched"""
        return self._lib.fs_batch_kernel_index(self._h)

    def derive_device(self, first=0, n=None, fields=A.DERIVE_ALL):
        """prepare_results with the results left on the device; returns {name: device pointer}."""
        n = self.level + 1 - first if n is None else n
        A.check(self._lib.fs_batch_derive_device(self._h, first, n, int(fields)), "derive_device")
        names = self.DERIVED + ("peak_amplitude",)
        return {k: self._lib.fs_batch_derived_device_ptr(self._h, i) for i, k in enumerate(names) if fields & (1 << i)}

    def hydrograph_device_ptr(self):
        return self._lib.fs_batch_hydrograph_device_ptr(self._h)
