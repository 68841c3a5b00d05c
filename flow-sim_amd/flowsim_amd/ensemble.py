"""Manning-n ensembles: many copies of one channel that differ only in the main-channel roughness,
stepped together on the device (BASELINE.json configs[3]; the reference does this one member at a
time, cases/gerd_roseires/n_calibrate.py:55-67 -> model.run(n_main=...))."""
import numpy as np

from . import _abi as A
from .batch import PreissmannBatch
from .hydromodel.preissmann import boundary_to_spec


def run_manning_ensemble(solver, n_main_values, tolerance=1e-4, max_iter=100, dtype="f64", device=0):
    """`solver`: a set-up (not yet run) PreissmannSolver whose channel provides geometry, boundaries
    and initial conditions; the initial conditions are shared by all members (the reference's
    members start from the same downstream level and the same flow; their GVF profiles differ
    with n - pass `initial_conditions[B, N, 2]` through `solver.channel.member_ics` to override).

    Returns dict(hydrographs[nt, 4, B], iterations[nt, B], status[B])."""
    ch = solver.channel
    n_vals = np.ascontiguousarray(n_main_values, dtype=np.float64)
    B, N, nt = len(n_vals), solver.number_of_nodes, solver.number_of_time_levels
    ics = getattr(ch, "member_ics", None)
    with PreissmannBatch(B, N, max(nt, 2), dtype=dtype, section_mode="table", device=device) as b:
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
