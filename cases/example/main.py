"""python -m cases.example.main   (BASELINE.json configs[0]: 20 km x 250 m channel draining into a
1.25 km2 reservoir; parameters of the reference case cases/example/main.py)"""
from src.hydromodel.boundary import Boundary
from src.hydromodel.channel import Channel
from src.hydromodel.hydrograph import Hydrograph
from src.hydromodel.lumped_storage import LumpedStorage
from src.hydromodel.preissmann import PreissmannSolver

BASE, PEAK = 1000.0, 10000.0
RISE, HOLD, FALL = 3 * 3600, 6 * 3600, 4 * 3600


def inflow(t):
    """trapezoidal flood wave"""
    if t <= 0:
        return BASE
    if t < RISE:
        return BASE + (PEAK - BASE) * t / RISE
    if t < RISE + HOLD:
        return PEAK
    if t < RISE + HOLD + FALL:
        return PEAK - (PEAK - BASE) * (t - RISE - HOLD) / FALL
    return BASE


def build():
    us = Boundary(condition='flow_hydrograph', bed_level=5, chainage=0, hydrograph=Hydrograph(function=inflow))
    ds = Boundary(condition='fixed_depth', initial_depth=5, bed_level=0, chainage=20000)
    ds.set_lumped_storage(LumpedStorage(surface_area=5000 * 250, min_stage=5, solution_boundaries=(0, 200)))
    channel = Channel(width=250, initial_flow=us.hydrograph.get_at(0), roughness=0.027,
                      upstream_boundary=us, downstream_boundary=ds)
    return PreissmannSolver(channel=channel, theta=0.8, time_step=3600, spatial_step=1000,
                            simulation_time=24 * 3600)


if __name__ == "__main__":
    solver = build()
    solver.run(verbose=1, max_iter=100)
    print("reservoir stage [m]:", solver.storage_stage.round(4))
    print('Finished Preissmann.')
