"""Hydrograph: a value as a function of time, given as a callable or as a [time, value] table
(reference: src/hydromodel/hydrograph.py:4-33).  Boundaries pre-sample it at k*dt for the device."""
import numpy as np


class Hydrograph:
    def __init__(self, function=None, table: np.ndarray = None):
        self.table = table
        self.used_function = self.interpolate_hydrograph if function is None else function

    def interpolate_hydrograph(self, time):
        if self.table is None:
            raise ValueError("Hydrograph is not defined.")
        return float(np.interp(time, self.table[:, 0], self.table[:, 1]))

    def get_at(self, time):
        return self.used_function(time)

    def set_table(self, table: np.ndarray):
        self.table = table

    def set_function(self, func):
        self.used_function = func

    def sample(self, n_levels, dt):
        """Values at t = 0, dt, ..., (n_levels-1) dt: what the kernel reads as the boundary target."""
        return np.array([self.get_at(k * dt) for k in range(n_levels)], dtype=np.float64)
