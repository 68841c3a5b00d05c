"""LumpedStorage: 0-D reservoir behind a fixed_depth boundary
(reference: src/hydromodel/lumped_storage.py:8-179).

The device path covers the configuration the bundled cases use: constant surface area, no
reservoir rating curve, no entrance losses - there the mass balance root (lumped_storage.py:24-35)
is Y_new = Y_old + vol_in / area, clamped at min_stage.  Area curves / outflow rating curves /
entrance losses are SURVEY section 8(f) rank 3 ("next")."""
import numpy as np


class LumpedStorage:
    def __init__(self, solution_boundaries: tuple, surface_area: float = None, min_stage: float = None,
                 rating_curve=None):
        self.rating_curve = rating_curve
        self.surface_area = surface_area
        self.min_stage = min_stage
        self.stage_hydrograph = []          # [[time, stage], ...] filled from the device result
        self.area_curve = None
        self.reservoir_length = None
        self.capture_losses = False
        self.Cc = 0.5
        self.K_q = 0
        if solution_boundaries is not None:
            self.Y_min, self.Y_max = solution_boundaries[0], solution_boundaries[1]

    def set_area_curve(self, table, alpha=1, beta=0, update_solution_boundaries=True):
        self.alpha, self.beta = alpha, beta
        self.area_curve = np.asarray(table, dtype=np.float64)
        self.area_gradient = np.gradient(self.area_curve[:, 1], self.area_curve[:, 0])
        if update_solution_boundaries:
            self.Y_min = np.min(self.area_curve[:, 0])
            self.Y_max = np.max(self.area_curve[:, 0])

    def area_at(self, stage):
        if self.area_curve is None:
            return self.surface_area
        return self.alpha * np.interp(stage + self.beta, self.area_curve[:, 0], self.area_curve[:, 1])

    def net_vol_change(self, Y1, Y2):
        """Volume between two stages (lumped_storage.py:168-179)."""
        if self.area_curve is None:
            return (Y2 - Y1) * self.surface_area
        step = np.min(np.abs(np.diff(self.area_curve[:, 0])))
        n = int(abs(Y2 - Y1) / step)
        if n > 2:
            ys = np.linspace(Y1, Y2, n)
            return np.trapezoid([self.area_at(y) for y in ys], ys)
        return 0.5 * (self.area_at(Y2) + self.area_at(Y1)) * (Y2 - Y1)

    def energy_loss(self, entry_area, flow, roughness, hydraulic_radius, A_str=None):
        if not self.capture_losses:
            return 0
        raise NotImplementedError("entrance losses of LumpedStorage are not part of the device path yet")

    def mass_balance(self, duration, vol_in, Y_old=None, time=None):
        """Closed form of the brentq root for the supported configuration."""
        self._check_supported()
        Y = Y_old + vol_in / self.surface_area
        if not (self.Y_min <= Y <= self.Y_max):
            raise ValueError("f(a) and f(b) must have different signs")
        return max(Y, self.min_stage)

    def _check_supported(self):
        if self.area_curve is not None or self.rating_curve is not None or self.capture_losses:
            raise NotImplementedError(
                "device path supports LumpedStorage with constant surface_area, no rating_curve and no "
                "entrance losses (SURVEY.md 8f rank 3 covers the general case)")

    def device_spec(self, bed_level):
        self._check_supported()
        if self.surface_area is None or self.min_stage is None:
            raise ValueError("Insufficient arguments for boundary condition.")
        return "storage", dict(surface_area=self.surface_area, min_stage=self.min_stage, Y_min=self.Y_min,
                               Y_max=self.Y_max, bed_level=bed_level)
