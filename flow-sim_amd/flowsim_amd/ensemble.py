"""Manning-n ensembles: many copies of one channel that differ only in the main-channel roughness,
stepped together on the device (BASELINE.json configs[3]; the reference does this one member at a
time, cases/gerd_roseires/n_calibrate.py:55-67 -> model.run(n_main=...))."""
import numpy as np

from . import _abi as A
from .batch import PreissmannBatch
from .hydromodel.preissmann import boundary_to_spec


def run_manning_ensemble(solver, n_main_values, tolerance=1e-4, max_iter=100, dtype="f64", device=0, monitor=True):
    """`solver`: a set-up (not yet run) PreissmannSolver whose channel provides geometry, boundaries
    and initial conditions; the initial conditions are shared by all members (the reference's
    members start from the same downstream level and the same flow; their GVF profiles differ
    with n - pass `initial_conditions[B, N, 2]` through `solver.channel.member_ics` to override).

    Returns dict(hydrographs[nt, 4, B], iterations[nt, B], status[B])."""
    ch = solver.channel
    n_vals = np.ascontiguousarray(n_main_values, dtype=np.float64)
    B, N, nt = len(n_vals), solver.number_of_nodes, solver.number_of_time_levels
    ics = getattr(ch, "member_ics", None)
    with PreissmannBatch(B, N, max(nt, 2), dtype=dtype, section_mode="table", device=device, monitor=monitor) as b:
        b.set_scheme(solver.theta, solver.time_step, solver.spatial_step, tolerance, max_iter)
        b.set_geometry_table(ch.node_geometry, n_main_override=n_vals)
        b.set_boundary(A.UPSTREAM, boundary_to_spec(ch.upstream_boundary, max(nt, 2), solver.time_step))
        b.set_boundary(A.DOWNSTREAM, boundary_to_spec(ch.downstream_boundary, max(nt, 2), solver.time_step))
        if ics is None:
            b.set_state(ch.initial_conditions[:, 0], ch.initial_conditions[:, 1])
        else:
            b.set_state(ics[:, :, 0], ics[:, :, 1])
        b.step(nt - 1)
        return dict(hydrographs=b.hydrographs(0, nt), iterations=b.iterations(0, nt), status=b.status())


def gvf_profiles(channel, n_main_values):
    """Initial depth profiles of every ensemble member at once: the reference's GVF backwater march
    (channel.py:307-378, Heun predictor-corrector from the downstream depth) vectorised over the
    members - they share the geometry table and differ in the main-channel Manning n.
    Returns initial_conditions[B, N, 2]."""
    from .hydromodel import cross_section as XS
    from .hydromodel import hydraulics
    geo = channel.node_geometry
    n_vals = np.asarray(n_main_values, dtype=np.float64)
    B, N = len(n_vals), len(geo["z_bed"])
    Q = channel.initial_flow_rate
    dx = channel.length / (N - 1)

    def node_geo(i):
        g = {k: np.full(B, v[i]) for k, v in geo.items()}
        g["n_main"] = n_vals          # the override is applied to every input section (custom_functions.py:147)
        return g

    def slope(h, i, S0):
        g = node_geo(i)
        hw = h + g["z_bed"]
        A, P, R, T, over = XS.props(g, hw)
        K = XS.conveyance(g, hw, (A, P, R, T, over))
        Fr = hydraulics.froude_array(T, A, Q)
        if np.any(Fr > 1.0):
            raise RuntimeError(f"GVF Error: Flow became supercritical at node {i}. "
                               "Downstream boundary control is not valid for this Q.")
        den = np.maximum(1 - Fr ** 2, 0.01)
        Se = Q * abs(Q) / K ** 2
        curv = g["curvature"]
        if np.any(curv != 0):
            neq = XS.equivalent_n(g, hw, (A, P, R, T, over), K)
            with np.errstate(divide="ignore", invalid="ignore"):
                f = hydraulics.darcy_weisbach_f(neq, R)
                Sc = (2.86 * np.sqrt(f) + 2.07 * f) * h ** 2 * Fr ** 2 / ((0.565 + np.sqrt(f)) * (1.0 / curv) ** 2)
            Se = Se + np.where(curv != 0, Sc, 0.0)
        return np.where((T < 1e-6) | (A < 1e-6), 0.0, (S0 - Se) / den)

    ic = np.empty((B, N, 2))
    ic[:, :, 1] = Q
    h = np.full(B, channel.downstream_boundary.initial_depth, dtype=np.float64)
    ic[:, N - 1, 0] = h
    z = geo["z_bed"]
    for i in reversed(range(N - 1)):
        S0 = (z[i] - z[i + 1]) / dx
        k1 = slope(h, i + 1, S0)
        h_pred = h - k1 * dx
        h_pred = np.where(h_pred <= 0, 0.01, h_pred)
        k2 = slope(h_pred, i, S0)
        h = h - 0.5 * (k1 + k2) * dx
        h = np.where(h <= 0, 0.01, h)
        ic[:, i, 0] = h
    return ic
