"""python -m cases.akbari_firoozi.main_preissmann   (BASELINE.json configs[1])

Flood wave of Akbari & Firoozi through a 29 km rectangular channel with the implicit four-point scheme,
written against the reference's import paths (src.hydromodel.*) and executed on the GPU."""
from src.hydromodel.boundary import Boundary
from src.hydromodel.channel import Channel
from src.hydromodel.hydrograph import Hydrograph
from src.hydromodel.preissmann import PreissmannSolver

from . import settings as cfg


def build():
    """channel + solver for the case; run() is left to the caller"""
    bed_drop = cfg.S_0 * cfg.length
    inflow = Boundary(condition='flow_hydrograph', hydrograph=Hydrograph(cfg.hydrograph), chainage=0, bed_level=bed_drop)
    outflow = Boundary(condition='normal_depth', chainage=cfg.length, bed_level=0)
    reach = Channel(upstream_boundary=inflow, downstream_boundary=outflow, interpolation_method='steady-state',
                    width=cfg.width, roughness=cfg.roughness, initial_flow=cfg.initial_flow)
    scheme = dict(theta=cfg.theta, time_step=cfg.preissmann_dt, spatial_step=cfg.spatial_step,
                  simulation_time=cfg.duration, regularization=False)
    return PreissmannSolver(channel=reach, **scheme)


def main():
    solver = build()
    solver.run(tolerance=cfg.tolerance, verbose=2)
    hours = [k * solver.time_step / 3600 for k in range(solver.number_of_time_levels)]
    for t, q_in, q_out in zip(hours, solver.flow[:, 0], solver.flow[:, -1]):
        print(f"t = {t:5.1f} h   inflow {q_in:8.3f}   outflow {q_out:8.3f} m3/s")


if __name__ == "__main__":
    main()
