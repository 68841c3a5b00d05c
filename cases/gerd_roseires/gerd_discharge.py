"""Upstream boundary of the case: release hydrograph of the GERD reservoir obtained by level-pool
routing of the inflow through the dam's outlets (time-only pre-processing on the host; semantics of
the reference's cases/gerd_roseires/gerd_discharge.py:7-123)."""
import numpy as np
from scipy.optimize import brentq

from src.hydromodel.hydrograph import Hydrograph

from . import settings

TURBINE_FLOW = 1562.5
CREST_GATED, CREST_STEPPED, CREST_EMERGENCY = 624.9, 640.0, 642.0      # weir crests [m]
C_GATED, C_STEPPED, C_EMERGENCY = 196.4017, 447.3594, 654.6723         # Q = C * head^1.5


def weir(c, crest, level):
    return c * max(0, level - crest) ** (3 / 2)


def gate_fraction(level):
    """gated spillway is opened progressively between its crest and the full supply level"""
    if level <= CREST_GATED:
        return 0
    if level >= CREST_STEPPED:
        return 1
    return (level - CREST_GATED) / (CREST_STEPPED - CREST_GATED)


def outlet_capacity(level):
    return (weir(C_GATED, CREST_GATED, level) * gate_fraction(level) + weir(C_STEPPED, CREST_STEPPED, level)
            + weir(C_EMERGENCY, CREST_EMERGENCY, level) + 0 + TURBINE_FLOW)


class GerdHydrograph(Hydrograph):
    def __init__(self):
        super().__init__(function=None, table=None)
        self.turbine_flow = TURBINE_FLOW

    def release(self, inflow, stage, initial_stage):
        """above the initial level everything the outlets can pass is released, below it the dam
        passes the inflow (at least the turbine flow)"""
        cap = outlet_capacity(stage)
        if stage > initial_stage:
            return cap
        return max(min(inflow, cap), self.turbine_flow)

    def build(self, inflow_hydrograph, time_step, duration, initial_stage):
        curve = np.loadtxt(settings.gerd_volume_curve_path, delimiter=",", dtype=np.float64)
        vols, stages = curve[:, 0], curve[:, 1]                       # [1e6 m3], [m]
        volume = lambda z: np.interp(x=z, xp=stages, fp=vols)
        n = duration // time_step
        self.table = np.empty((n + 1, 2), dtype=np.float64)
        z0 = initial_stage
        qin0 = inflow_hydrograph.get_at(0)
        qout0 = self.release(qin0, z0, initial_stage)
        self.table[0] = (0, qout0)
        for k in range(1, n + 1):
            t = k * time_step
            qin1 = inflow_hydrograph.get_at(t)
            mean_in = 0.5 * (qin1 + qin0)
            v0 = volume(z0)

            def imbalance(z1):
                mean_out = 0.5 * (self.release(qin1, z1, initial_stage) + qout0)
                return (volume(z1) - v0) - (mean_in - mean_out) * time_step * 1e-6

            z1 = brentq(f=imbalance, a=CREST_GATED, b=645)
            qout1 = self.release(qin1, z1, initial_stage)
            self.table[k] = (t, qout1)
            z0, qin0, qout0 = z1, qin1, qout1
