"""PreissmannSolver: the reference's entry point for the implicit 4-point scheme
(src/hydromodel/preissmann.py:9-163) with run() executed on an MI355X.

run() flattens the channel (node geometry table, two boundary specs, initial conditions) into a
one-reach PreissmannBatch, advances all time levels in one kernel launch and copies the
depth/flow[nt, N] history back.  Error behaviour follows the reference: ValueError when a level
does not converge within max_iter (preissmann.py:124-126) or the residual turns NaN (:135-137).
"""
import numpy as np

from .. import _abi as A
from ..batch import BoundarySpec, PreissmannBatch
from .hydraulics import froude_array
from .solver import Solver
from . import cross_section as XS

_KIND = {"flow": A.BC_FLOW_HYDROGRAPH, "stage": A.BC_STAGE_HYDROGRAPH, "fixed": A.BC_FIXED_DEPTH,
         "normal": A.BC_NORMAL_DEPTH, "power": A.BC_RATING_POWER, "poly": A.BC_RATING_POLY,
         "blend": A.BC_RATING_BLEND, "storage": A.BC_STORAGE, "storage_curve": A.BC_STORAGE_CURVE}


def boundary_to_spec(boundary, n_levels, dt, host_fallback=False) -> BoundarySpec:
    """Device form of a boundary; with host_fallback a plugin without one (NotImplementedError from its
    device_spec) becomes an FS_BC_HOST_ROW boundary instead, evaluated in Python every Newton iteration."""
    try:
        kind, params, target = boundary.device_spec(n_levels, dt)
    except NotImplementedError:
        if not host_fallback:
            raise
        return BoundarySpec(A.BC_HOST_ROW)
    return BoundarySpec(_KIND[kind], params, target)


class PreissmannSolver(Solver):
    def __init__(self, theta, **kwargs):
        super().__init__(**kwargs)
        self.theta = theta
        self.unknowns = None
        self.type = 'preissmann'
        self.iterations = None                     # Newton iterations per time level, filled by run()
        self.initialize_t0()

    def initialize_t0(self) -> None:
        super().initialize_t0()
        self.unknowns = self.channel.initial_conditions.flatten()     # x = [h0, Q0, h1, Q1, ...]

    # ---- derivatives of the dry-bed regularisation (preissmann.py:800-872); see Solver.area_at: the branch raises
    # TypeError in the reference as soon as it is entered (Channel.area_at has no `h` parameter), and here
    def dAreg_dA(self, i):
        """d A_reg / dA = (1 + (A - A_min) / sqrt((A - A_min)^2 + eps^2)) / 2"""
        A_min = self.channel.area_at(i=i, h=self.H_MIN)
        excess = self.area_at(i=i, regularization=False) - A_min
        return 0.5 * (1.0 + excess / np.sqrt(excess ** 2 + self.eps ** 2))

    def dQe_dA(self, i):
        """d(chi Q)/dA with chi = A_reg / (A_reg + A_min): A_min Q A_reg' / (A_reg + A_min)^2"""
        A_min = self.channel.area_at(i=i, h=self.H_MIN)
        A_reg = self.area_at(i=i, regularization=True)
        return A_min * self.flow_at(i=i, chi_scaling=False) * self.dAreg_dA(i=i) / (A_reg + A_min) ** 2

    def dQe_dQ(self, i):
        """d(chi Q)/dQ = chi"""
        A_min = self.channel.area_at(i=i, h=self.H_MIN)
        A_reg = self.area_at(i=i, regularization=True)
        return A_reg / (A_reg + A_min)

    def run(self, tolerance=1e-4, verbose=3, max_iter=100, diagnos=False, dtype="f64") -> None:
        """preissmann.py:101-163.  `diagnos`: in the reference it adds a NaN check and a SuperLU condition estimate per
        iteration and raises ValueError("Jacobian is ill-conditioned (rcond too small)") below 1e-12 (:133-144) - on current
        scipy that line itself fails (SuperLU objects have no `rcond`), SURVEY section 0.  Here the kernel watches the
        conditioning of its own elimination for free (FS_ILL_CONDITIONED, include/flowsim_abi.h): with diagnos=True such a
        run raises the reference's ValueError text, without it the run completes, `self.ill_conditioned` is set and a
        RuntimeWarning says that the results may differ from the reference's beyond 1e-8."""
        ch = self.channel
        if self.regularization:
            self.area_at(k=0, i=0)                 # the reference's first residual evaluation: raises TypeError (Solver.area_at)
        N, nt = self.number_of_nodes, self.number_of_time_levels
        geo = ch.node_geometry
        rect = ("irr_npts" not in geo and np.all(geo["is_compound"] < 0.5) and np.all(geo["m_main"] == 0) and np.all(geo["curvature"] == 0)
                and np.ptp(geo["b_main"]) == 0 and np.ptp(geo["n_main"]) == 0 and ch.input_xs is not None
                and len(ch.input_xs) == 2)
        us_spec = boundary_to_spec(ch.upstream_boundary, max(nt, 2), self.time_step, host_fallback=True)
        ds_spec = boundary_to_spec(ch.downstream_boundary, max(nt, 2), self.time_step, host_fallback=True)
        host_sides = [side for side, sp in ((A.UPSTREAM, us_spec), (A.DOWNSTREAM, ds_spec)) if sp.kind == A.BC_HOST_ROW]
        # the general reservoir row and host-evaluated rows are compiled into the table / polyline kernels only
        rect = rect and A.BC_STORAGE_CURVE not in (us_spec.kind, ds_spec.kind) and not host_sides
        poly = "irr_npts" in geo                   # any IrregularSection node (cross_section.py:207-543)
        mode = "irregular" if poly else ("rect_uniform" if rect else "table")
        with PreissmannBatch(1, N, max(nt, 2), dtype=dtype, section_mode=mode, history=True, trace=(verbose == 3)) as b:
            b.set_scheme(self.theta, self.time_step, self.spatial_step, tolerance, max_iter)
            if rect:
                b.set_geometry_uniform(geo["b_main"][0], geo["n_main"][0], geo["z_bed"][0], geo["z_bed"][-1])
            elif poly:
                b.set_geometry_irregular(geo)
            else:
                b.set_geometry_table(geo)
            b.set_boundary(A.UPSTREAM, us_spec)
            b.set_boundary(A.DOWNSTREAM, ds_spec)
            b.set_state(ch.initial_conditions[:, 0], ch.initial_conditions[:, 1])
            if nt > 1 and host_sides:
                self._run_with_host_rows(b, host_sides, nt)
            elif nt > 1:
                b.step(nt - 1)
            status = int(b.status()[0])
            self.kernel_entry = A.kernel_table()[b.kernel_index()] if nt > 1 else None      # the instantiation that ran (diagnostic)
            self.ill_conditioned = status == A.ILL_CONDITIONED
            if self.ill_conditioned:
                status = A.OK                      # a warning: the run is complete
            its = b.iterations(0, max(nt, 2))[:nt, 0]
            h, Q = b.history_arrays(0, max(nt, 2))
            gh, gQ = b.guess()
            self.residual_norms = b.residual_trace(0, max(nt, 2))[:nt, :, 0] if verbose == 3 else None
            stages = b.storage_stages(0, max(nt, 2))[:nt, 0] if ch.downstream_boundary.lumped_storage is not None else None
            self._derived = {k: v[:, 0] if v.ndim == 3 else v[0] for k, v in b.derive(0, nt).items()} \
                if status == A.OK and nt > 1 else None
        self.iterations = its
        self.unknowns = np.empty(2 * N)
        self.unknowns[0::2], self.unknowns[1::2] = gh[0], gQ[0]
        if status != A.OK:
            # levels before the failing one are complete; the failing level keeps the last Newton vector
            k_fail = int(np.flatnonzero(its > 0)[-1]) if np.any(its > 0) else 1
            self.time_level = k_fail
            self.depth[:k_fail], self.flow[:k_fail] = h[:k_fail, 0], Q[:k_fail, 0]
            self.depth[k_fail], self.flow[k_fail] = gh[0], gQ[0]
            self.check_criticality()
            if status == A.MAX_ITER:
                raise ValueError(f'Convergence within {max_iter} iterations couldn\'t be achieved.')
            if status == A.STORAGE_RANGE:
                raise ValueError("f(a) and f(b) must have different signs")      # what brentq raises in the reference
            if status == A.TEAM_STALL:
                raise RuntimeError("device: a workgroup of this reach's team did not arrive (FS_TEAM_STALL, include/flowsim_abi.h)")
            raise ValueError("NaN in system assembly")
        self.time_level = nt - 1
        self.depth[:], self.flow[:] = h[:nt, 0], Q[:nt, 0]
        if self.ill_conditioned:
            if diagnos:
                self.check_criticality()
                raise ValueError("Jacobian is ill-conditioned (rcond too small)")          # preissmann.py:144
            import warnings
            warnings.warn("Preissmann systems of this run are ill-conditioned (supercritical flow over a long stretch?): results "
                          "may differ from another solver's beyond 1e-8; run(diagnos=True) raises instead", RuntimeWarning)
        st = ch.downstream_boundary.lumped_storage
        if st is not None and A.DOWNSTREAM not in host_sides:     # (a host-evaluated storage keeps the list itself)
            st.stage_hydrograph = [[k * self.time_step, float(stages[k])] for k in range(1, nt)]
        if verbose >= 1:      # same lines as the reference prints while it runs (preissmann.py:116-159)
            for k in range(1, nt):
                print(f'\n> Time level #{k}')
                if verbose == 3:
                    for i in range(min(its[k], A.TRACE_CAP)):
                        print(f">> Iteration #{i + 1}: Error = {self.residual_norms[k, i]}")
                if verbose == 2:
                    print(f'>> {its[k]} iterations.')
        self._finalize(verbose)

    def _run_with_host_rows(self, b, host_sides, nt) -> None:
        """The Newton loop with boundary plugins that have no device form (any object honouring discharge(stage, time) /
        dQ_dz, a LumpedStorage with a callable rating curve): one kernel launch per Newton iteration, the rows of those
        boundaries evaluated here in between, in the reference's order - both residuals (preissmann.py:76-77), then the
        upstream and the downstream derivatives (:322-344) - because a plugin may carry state from call to call."""
        ch, dt = self.channel, self.time_step
        ends = {A.UPSTREAM: (ch.upstream_boundary, 0, 1), A.DOWNSTREAM: (ch.downstream_boundary, 2, 3)}
        for k in range(1, nt):
            t = k * dt
            Q_old = b.hydrographs(k - 1, 1)[0, :, 0]            # flow[k-1] at the two ends (vol_in, preissmann.py:314)
            while True:
                it = b.boundary_iterate()[:, 0]
                res, rows = {}, {}
                for side in sorted(host_sides):
                    bnd, ih, iq = ends[side]
                    vol = 0.5 * (it[iq] + Q_old[iq]) * dt
                    res[side] = bnd.condition_residual(depth=it[ih], flow=it[iq], time=t, duration=dt, vol_in=vol)
                for side in sorted(host_sides):
                    bnd, ih, iq = ends[side]
                    vol = 0.5 * (it[iq] + Q_old[iq]) * dt
                    rows[side] = (bnd.df_dh(depth=it[ih], flow_rate=it[iq], time=t),
                                  bnd.df_dQ(depth=it[ih], flow_rate=it[iq], duration=dt, time=t, vol_in=vol))
                for side in host_sides:
                    b.set_host_rows(side, rows[side][0], rows[side][1], res[side])
                if b.iterate() == 0:
                    break
            if int(b.status()[0]) not in (A.OK, A.ILL_CONDITIONED):
                return

    def check_criticality(self) -> None:
        """Froude diagnosis printed before a convergence failure is raised (preissmann.py:179-198)."""
        geo = self.channel.node_geometry
        k = self.time_level
        A_, T = XS.area_top(self.channel.xs_at_node, geo, self.depth[k] + geo["z_bed"])
        fr = froude_array(T, A_, self.flow[k])
        bad = False
        for x, f in zip(self.channel.ch_at_node, fr):
            if f == 1.0:
                bad = True
                print(f'WARNING: Flow goes critical at x = {x} m. Fr = {f}.')
            elif f > 1.0:
                bad = True
                print(f'WARNING: Flow goes supercritical at x = {x} m. Fr = {f}.')
        if not bad:
            print('Flow is subcritical.')
