"""Akbari & Firoozi test channel: 29 km x 120 m rectangle, sinusoidal flood wave 100 -> 300 m3/s
(parameters of the reference case cases/akbari_firoozi/settings.py)."""
from math import cos, pi, sin

width = 120
length = 29000
roughness = 0.023
S_0 = 0.00061

spatial_step = 1000
duration = 20 * 3600
tolerance = 1e-4
theta = 0.5
preissmann_dt = 3600

initial_flow = 100


def hydrograph(t):
    rise_end, fall_end = 5 * 3600, 15 * 3600
    amplitude = 200
    if t <= rise_end:
        return amplitude / 2 * sin(pi * t / rise_end - pi / 2) + amplitude / 2 + initial_flow
    if t <= fall_end:
        return amplitude / 2 * cos(pi * (t - rise_end) / (fall_end - rise_end)) + amplitude / 2 + initial_flow
    return initial_flow
