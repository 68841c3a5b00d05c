"""python -m cases.gerd_roseires.main  -  GERD -> Roseires reach of the Blue Nile.

The scenario is described by a `Scenario` record (defaults from settings.py, any field overridable by
keyword); `build()` turns it into channel + solver, `run()` solves it.  Same scenario parameters as the
reference's cases/gerd_roseires/model.py:10-113, without its plotting / shapefile export tail."""
from dataclasses import dataclass, fields, replace
from typing import Callable, Optional

import numpy as np

from src.hydromodel.boundary import Boundary
from src.hydromodel.channel import Channel
from src.hydromodel.hydrograph import Hydrograph
from src.hydromodel.preissmann import PreissmannSolver

from . import settings
from .gerd_discharge import GerdHydrograph
from .inputs import import_hydrograph, import_table, load_trapezoid_sections
from .roseires_rating_curve import RoseiresRatingCurve


@dataclass(frozen=True)
class Scenario:
    # channel roughness (None: the surveyed values of the section table)
    n_main: Optional[float] = None
    n_fp: Optional[float] = None
    # scheme
    theta: float = settings.theta
    spatial_step: float = settings.spatial_step
    time_step: float = settings.time_step
    sim_duration: Optional[float] = settings.sim_duration
    # hydrology: inflow into GERD as a function of time, or - when None - as the table at inflow_hyd_path
    inflow_hyd_func: Optional[Callable] = settings.inflow_hyd_func
    inflow_hyd_path: str = settings.inflow_hyd_path
    inflow_scale: float = 1.0                   # table inflow only: flood wave scaled about its base flow
    gerd_level: float = settings.initial_gerd_level
    with_gerd: bool = True                      # False: the inflow reaches the channel unregulated
    initial_roseires_level: float = settings.initial_roseires_level
    jammed_spillways: int = settings.JAMMED_SPILLWAYS
    jammed_sluice_gates: int = settings.JAMMED_SLUICEGATES
    smooth_gates: bool = True                   # False: operated gates (time- and history-dependent: evaluated on the host)
    # geometry tables (coords_path None: straight channel, no curvature)
    coords_path: Optional[str] = settings.coords_path
    cross_sections_path: str = settings.cross_sections_path


_FIELDS = {f.name for f in fields(Scenario)}


def build(scenario: Scenario = None, **overrides):
    """(solver, input sections) for a scenario; keyword arguments override single fields"""
    unknown = set(overrides) - _FIELDS
    if unknown:
        raise TypeError(f"unknown scenario parameter(s): {sorted(unknown)}")
    sc = replace(scenario or Scenario(), **overrides)

    if sc.inflow_hyd_func is not None:
        reservoir_inflow = Hydrograph(function=sc.inflow_hyd_func)
    else:
        table = import_hydrograph(sc.inflow_hyd_path)
        table[:, 1] = table[0, 1] + sc.inflow_scale * (table[:, 1] - table[0, 1])
        reservoir_inflow = Hydrograph(table=table)
    if sc.sim_duration is not None:
        duration = int(sc.sim_duration)
    elif reservoir_inflow.table is not None:
        duration = int(reservoir_inflow.table[-1, 0])
    else:
        raise ValueError("Simulation duration must be specified.")

    # GERD routes its inflow through the reservoir; what it releases enters the reach
    release = GerdHydrograph()
    release.build(inflow_hydrograph=reservoir_inflow, time_step=sc.time_step, duration=duration, initial_stage=sc.gerd_level)
    q0 = release.get_at(time=0)

    stations, sections = load_trapezoid_sections(sc.cross_sections_path, n_main=sc.n_main, n_fp=sc.n_fp)
    z_dam = sections[-1].z_min
    head = Boundary(condition='flow_hydrograph', chainage=stations[0],
                    hydrograph=release if sc.with_gerd else reservoir_inflow)
    gate_curve = RoseiresRatingCurve(initial_stage=sc.initial_roseires_level, initial_flow=q0,
                                     jammed_spillways=sc.jammed_spillways, jammed_sluice_gates=sc.jammed_sluice_gates,
                                     smooth=sc.smooth_gates)
    dam = Boundary(condition='rating_curve', rating_curve=gate_curve, chainage=stations[-1], bed_level=z_dam,
                   initial_depth=sc.initial_roseires_level - z_dam)

    reach = Channel(upstream_boundary=head, downstream_boundary=dam, initial_flow=q0)
    if sc.coords_path is not None:
        centreline = import_table(sc.coords_path)
        reach.set_coords(chainages=centreline[:, 0], coords=centreline[:, 1:])
    reach.set_cross_sections(chainages=stations, sections=sections)
    solver = PreissmannSolver(channel=reach, theta=sc.theta, time_step=sc.time_step, spatial_step=sc.spatial_step,
                              simulation_time=duration)
    return solver, sections


def run(Q=None, tolerance=settings.tolerance, verbose=1, **overrides):
    """Solves the scenario.  Returns the solver, or - when discharges Q are given - the water levels below
    GERD at those discharges, read off the simulated stage-discharge relation of the first node (the
    quantity n_calibrate compares with the surveyed tail-water curve)."""
    solver, sections = build(**overrides)
    solver.run(tolerance=tolerance, verbose=verbose - 1)
    if Q is None:
        return solver
    stage = solver.depth[:, 0] + sections[0].z_min
    return np.interp(x=Q, xp=solver.flow[:, 0], fp=stage)
