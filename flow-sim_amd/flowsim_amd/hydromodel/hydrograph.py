"""Hydrograph: a value as a function of time, given as a callable or as a [time, value] table
(reference: src/hydromodel/hydrograph.py:4-33, same attributes and method names).  The boundaries
pre-sample it at k*dt for the device: the kernel reads one number per time level, never a callable."""
import numpy as np


class Hydrograph:
    def __init__(self, function=None, table: np.ndarray = None):
        self.table = table
        # a user function wins over the table; without one the table is interpolated linearly
        self.used_function = function if function is not None else self.interpolate_hydrograph

    # ---- evaluation --------------------------------------------------------------------------
    def interpolate_hydrograph(self, time):
        """linear interpolation in the table, end values held outside it (np.interp)"""
        tab = self.table
        if tab is None:
            raise ValueError("Hydrograph is not defined.")
        return float(np.interp(time, tab[:, 0], tab[:, 1]))

    def get_at(self, time):
        return self.used_function(time)

    __call__ = get_at

    def sample(self, n_levels, dt):
        """Values at t = 0, dt, ..., (n_levels-1) dt: what the kernel reads as the boundary target.
        A table-backed hydrograph is sampled in one np.interp call, a callable level by level."""
        times = np.arange(n_levels) * dt
        if self.used_function == self.interpolate_hydrograph and self.table is not None:
            return np.interp(times, self.table[:, 0], self.table[:, 1]).astype(np.float64)
        return np.array([self.get_at(t) for t in times], dtype=np.float64)

    # ---- definition ----------------------------------------------------------------------------
    def set_table(self, table: np.ndarray):
        """time [s] in the first column, value in the second"""
        self.table = table

    def set_function(self, func):
        self.used_function = func

    def __repr__(self):
        kind = "table[%d]" % len(self.table) if self.table is not None else "function"
        return f"Hydrograph({kind})"
