"""python -m cases.akbari_firoozi.main_preissmann   (BASELINE.json configs[1])"""
from src.hydromodel.boundary import Boundary
from src.hydromodel.channel import Channel
from src.hydromodel.hydrograph import Hydrograph
from src.hydromodel.preissmann import PreissmannSolver

from . import settings as S


def build():
    us = Boundary(condition='flow_hydrograph', bed_level=S.S_0 * S.length, chainage=0,
                  hydrograph=Hydrograph(S.hydrograph))
    ds = Boundary(condition='normal_depth', bed_level=0, chainage=S.length)
    channel = Channel(width=S.width, initial_flow=S.initial_flow, roughness=S.roughness,
                      upstream_boundary=us, downstream_boundary=ds, interpolation_method='steady-state')
    return PreissmannSolver(channel=channel, theta=S.theta, time_step=S.preissmann_dt,
                            spatial_step=S.spatial_step, simulation_time=S.duration, regularization=False)


if __name__ == "__main__":
    solver = build()
    solver.run(verbose=2, tolerance=S.tolerance)
    print("outflow hydrograph [m3/s]:", solver.flow[:, -1].round(4))
