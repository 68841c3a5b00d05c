"""Blue Nile reach GERD -> Roseires (120.4 km, 21 surveyed sections as compound trapezoids):
simulation parameters and input tables.  The values are those of the reference case
(cases/gerd_roseires/settings.py); the tables under data/ are the reference's input data (MIT licensed)."""
import os
from math import pi, sin

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def _table(name):
    return os.path.join(_DATA, name)


# --- numerical scheme ---------------------------------------------------------------------------
theta, spatial_step, time_step = 0.6, 1000, 3600
sim_duration = 384 * time_step
tolerance = 1e-6

# --- reservoirs and gates -------------------------------------------------------------------------
initial_gerd_level, initial_roseires_level = 637., 487.
JAMMED_SPILLWAYS = JAMMED_SLUICEGATES = 0

# --- synthetic design flood: quarter-sine rise over a day, a day at the peak, quarter-sine recession --
base_flow, peak_flow = 1562.5, 26000.
lag_time, time_to_peak, time_at_peak = 0.0, 24 * 3600, 24 * 3600
_KNOTS = np.array([0.0, time_to_peak, time_to_peak + time_at_peak, 2 * time_to_peak + time_at_peak])


def sin_wave(time):
    """discharge of the design flood at `time` [s]: base + (peak - base) sin(pi/2 u), u running 0 -> 1 on the
    rising limb, staying at 1 over the plateau and running 1 -> 2 on the recession"""
    elapsed = float(time) - lag_time
    if elapsed <= _KNOTS[0] or elapsed >= _KNOTS[-1]:
        return base_flow
    u = np.interp(elapsed, _KNOTS, [0.0, 1.0, 1.0, 2.0])
    return base_flow + (peak_flow - base_flow) * sin(0.5 * pi * float(u))


inflow_hyd_func = sin_wave

# --- input tables -----------------------------------------------------------------------------------
inflow_hyd_path = _table("inflow_hydrograph.csv")
inflow_hyd_small_path = _table("inflow_hydrograph_small.csv")
coords_path = _table("centerline_coords.csv")
cross_sections_path = _table("composite_trapezoids.csv")
gerd_volume_curve_path = _table("gerd_vol_curve.csv")
spillway_table_path = _table("roseires_spillway_releases.csv")
sluice_table_path = _table("roseires_deep_sluice_releases.csv")
