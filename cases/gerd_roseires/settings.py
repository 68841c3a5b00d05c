"""Blue Nile reach GERD -> Roseires (120.4 km, 21 surveyed sections as compound trapezoids):
simulation parameters and input tables (values of the reference case cases/gerd_roseires/settings.py;
the tables under data/ are the reference's input data, MIT licensed)."""
import os
from math import pi, sin

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "data")

spatial_step = 1000
time_step = 3600
theta = 0.6
sim_duration = 3600 * 384
tolerance = 1e-6

initial_roseires_level = 487.
initial_gerd_level = 637.
JAMMED_SPILLWAYS = 0
JAMMED_SLUICEGATES = 0

base_flow, peak_flow = 1562.5, 26000.
lag_time, time_to_peak, time_at_peak = 0.0, 3600 * 24, 3600 * 24


def sin_wave(time):
    """synthetic design flood: quarter-sine rise, plateau, quarter-sine fall"""
    t = time - lag_time
    if t <= 0:
        return base_flow
    if t < time_to_peak:
        return base_flow + sin(0.5 * pi * float(t) / time_to_peak) * (peak_flow - base_flow)
    if t < time_to_peak + time_at_peak:
        return peak_flow
    if t < 2 * time_to_peak + time_at_peak:
        return base_flow + sin(0.5 * pi * float(t - time_at_peak) / time_to_peak) * (peak_flow - base_flow)
    return base_flow


inflow_hyd_path = os.path.join(DATA, "inflow_hydrograph.csv")
inflow_hyd_small_path = os.path.join(DATA, "inflow_hydrograph_small.csv")
inflow_hyd_func = sin_wave
coords_path = os.path.join(DATA, "centerline_coords.csv")
cross_sections_path = os.path.join(DATA, "composite_trapezoids.csv")
gerd_volume_curve_path = os.path.join(DATA, "gerd_vol_curve.csv")
spillway_table_path = os.path.join(DATA, "roseires_spillway_releases.csv")
sluice_table_path = os.path.join(DATA, "roseires_deep_sluice_releases.csv")
