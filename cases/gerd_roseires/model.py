"""python -m cases.gerd_roseires.main  -  GERD -> Roseires reach of the Blue Nile
(set-up of the reference's cases/gerd_roseires/model.py:10-113 without its plotting / shapefile tail)."""
import numpy as np

from src.hydromodel.boundary import Boundary
from src.hydromodel.channel import Channel
from src.hydromodel.hydrograph import Hydrograph
from src.hydromodel.preissmann import PreissmannSolver

from . import settings
from .gerd_discharge import GerdHydrograph
from .inputs import import_hydrograph, import_table, load_trapezoid_sections
from .roseires_rating_curve import RoseiresRatingCurve


def build(n_main=None, n_fp=None, initial_roseires_level=settings.initial_roseires_level, theta=settings.theta,
          spatial_step=settings.spatial_step, time_step=settings.time_step, sim_duration=settings.sim_duration,
          inflow_hyd_path=settings.inflow_hyd_path, inflow_hyd_func=settings.inflow_hyd_func,
          coords_path=settings.coords_path, cross_sections_path=settings.cross_sections_path,
          jammed_spillways=settings.JAMMED_SPILLWAYS, jammed_sluice_gates=settings.JAMMED_SLUICEGATES,
          gerd_level=settings.initial_gerd_level, with_gerd=True):
    inflow = Hydrograph(table=import_hydrograph(inflow_hyd_path)) if inflow_hyd_func is None \
        else Hydrograph(function=inflow_hyd_func)
    if sim_duration is None:
        if inflow.table is None:
            raise ValueError("Simulation duration must be specified.")
        duration = int(inflow.table[-1, 0])
    else:
        duration = int(sim_duration)
    release = GerdHydrograph()
    release.build(inflow_hydrograph=inflow, time_step=time_step, duration=duration, initial_stage=gerd_level)
    initial_flow = release.get_at(time=0)

    chainages, sections = load_trapezoid_sections(cross_sections_path, n_main=n_main, n_fp=n_fp)
    dam_bed = sections[-1].z_min
    upstream = Boundary(condition='flow_hydrograph', hydrograph=release if with_gerd else inflow, chainage=chainages[0])
    gates = RoseiresRatingCurve(initial_stage=initial_roseires_level, initial_flow=initial_flow,
                                jammed_sluice_gates=jammed_sluice_gates, jammed_spillways=jammed_spillways)
    roseires = Boundary(initial_depth=initial_roseires_level - dam_bed, bed_level=dam_bed, condition='rating_curve',
                        rating_curve=gates, chainage=chainages[-1])
    river = Channel(initial_flow=initial_flow, upstream_boundary=upstream, downstream_boundary=roseires)
    if coords_path is not None:
        xy = import_table(coords_path)
        river.set_coords(coords=xy[:, 1:], chainages=xy[:, 0])
    river.set_cross_sections(chainages=chainages, sections=sections)
    solver = PreissmannSolver(channel=river, theta=theta, time_step=time_step, spatial_step=spatial_step,
                              simulation_time=duration)
    return solver, sections


def run(Q=None, tolerance=settings.tolerance, verbose=1, **kwargs):
    """returns the solver, or - when discharges Q are given - the GERD tail-water levels at those
    discharges read off the simulated upstream rating (what n_calibrate uses)"""
    solver, sections = build(**kwargs)
    solver.run(verbose=verbose - 1, tolerance=tolerance)
    if Q is not None:
        return np.interp(x=Q, xp=solver.flow[:, 0], fp=solver.depth[:, 0] + sections[0].z_min)
    return solver
