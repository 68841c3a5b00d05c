"""LumpedStorage: 0-D reservoir behind a fixed_depth boundary
(reference: src/hydromodel/lumped_storage.py:8-179).

Two device forms: the configuration the bundled cases use (constant surface area, no reservoir
rating curve, no entrance losses), where the mass-balance root (lumped_storage.py:24-35) is
Y_new = Y_old + vol_in / area clamped at min_stage (FS_BC_STORAGE); and the general one (area
curve, power / polynomial outflow rating curve, friction + empirical entrance losses), where the
kernel runs the Brent iteration itself (FS_BC_STORAGE_CURVE).  The expansion loss needs an A_str
that the reference's Boundary never passes (boundary.py:119-121), so it is always zero there too."""
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
        """Friction over reservoir_length + K_q V^2/2g (+ expansion when A_str is given); lumped_storage.py:47-74."""
        if not self.capture_losses:
            return 0
        from . import hydraulics
        V = flow / entry_area
        hf = hydraulics.Sf(Q=flow, K=hydraulics.conveyance(A=entry_area, n=roughness, R=hydraulic_radius)) * self.reservoir_length
        h_exp = 0 if A_str is None else (1 - entry_area / A_str) ** 2 * V ** 2 / (2 * hydraulics.g)
        return hf + h_exp + self.K_q * V ** 2 / (2 * hydraulics.g)

    def dhl_dA(self, entry_area, flow, roughness, hydraulic_radius, dR_dA, A_str=None):
        """d(head loss)/dA of the entry section (lumped_storage.py:76-113); expansion term as energy_loss"""
        if not self.capture_losses:
            return 0
        from . import hydraulics
        K = hydraulics.conveyance(A=entry_area, n=roughness, R=hydraulic_radius)
        dhf = hydraulics.dSf_dA(Q=flow, K=K, dK_dA=hydraulics.dK_dA(A=entry_area, n=roughness, R=hydraulic_radius, dR_dA=dR_dA)) \
            * self.reservoir_length
        V, dV = flow / entry_area, -flow / entry_area ** 2
        dexp = 0
        if A_str is not None:
            Kx = (1 - entry_area / A_str) ** 2
            dexp = (Kx * 2 * V * dV + V ** 2 * 2 * (1 - entry_area / A_str) * (-1 / A_str)) / (2 * hydraulics.g)
        return dhf + dexp + self.K_q * 2 * V * dV / (2 * hydraulics.g)

    def dhl_dQ(self, entry_area, flow, roughness, hydraulic_radius, A_str=None):
        """d(head loss)/dQ (lumped_storage.py:115-143; its expansion term differentiates V with -Q/A^2)"""
        if not self.capture_losses:
            return 0
        from . import hydraulics
        K = hydraulics.conveyance(A=entry_area, n=roughness, R=hydraulic_radius)
        V = flow / entry_area
        dexp = 0 if A_str is None else (1 - entry_area / A_str) ** 2 * 2 * V * (-flow / entry_area ** 2) / (2 * hydraulics.g)
        return hydraulics.dSf_dQ(Q=flow, K=K) * self.reservoir_length + dexp + self.K_q * 2 * V * (1. / entry_area) / (2 * hydraulics.g)

    def dY_new_dvol_in(self, duration, vol_in, Y_old, time=None) -> float:
        """d(new stage)/d(inflow volume) = 1 / surface area, 0 on the min_stage floor (lumped_storage.py:37-45)"""
        Y_new = self.mass_balance(duration, vol_in, Y_old, time)
        return 0.0 if Y_new <= self.min_stage else 1 / self.area_at(Y_new)

    def mass_balance(self, duration, vol_in, Y_old=None, time=None):
        """Stage after taking vol_in over `duration` (lumped_storage.py:24-35): host evaluation for set-up,
        post-processing and for storages whose rating curve has no device form (Boundary.condition_residual);
        otherwise the Newton loop uses the device form."""
        from scipy.optimize import brentq

        def f(Y_new):
            q_out = 0.5 * (self.rating_curve.discharge(Y_old, time) + self.rating_curve.discharge(Y_new, time)) \
                if self.rating_curve else 0.0
            return self.net_vol_change(Y_old, Y_new) - (vol_in - q_out * duration)
        return max(brentq(f, self.Y_min, self.Y_max), self.min_stage)

    def _is_simple(self):
        return self.area_curve is None and self.rating_curve is None and not self.capture_losses

    def device_spec(self, bed_level):
        if self.min_stage is None or (self.area_curve is None and self.surface_area is None):
            raise ValueError("Insufficient arguments for boundary condition.")
        if self._is_simple():
            return "storage", dict(surface_area=self.surface_area, min_stage=self.min_stage, Y_min=self.Y_min,
                                   Y_max=self.Y_max, bed_level=bed_level)
        p = dict(min_stage=self.min_stage, Y_min=self.Y_min, Y_max=self.Y_max, bed_level=bed_level,
                 surface_area=self.surface_area or 0.0, rc_type=0.0)
        if self.area_curve is not None:
            p.update(alpha=self.alpha, beta=self.beta, curve=self.area_curve)
        rc = self.rating_curve
        if rc is not None:
            from .rating_curve import RatingCurve
            if (getattr(rc, "function", None) is not None or getattr(rc, "type", None) not in ("power", "polynomial")
                    or type(rc).discharge is not RatingCurve.discharge):
                raise NotImplementedError("only RatingCurve.set('power' | 'polynomial', ...) reservoir rating curves run inside the "
                                          "kernel; this one is evaluated on the host (FS_BC_HOST_ROW)")
            p.update(rc_type=1.0 if rc.type == "power" else 2.0, rc_a=rc.a, rc_b=rc.b,
                     rc_c=getattr(rc, "c", 0.0) if rc.type == "polynomial" else 0.0, rc_shift=getattr(rc, "stage_shift", 0.0))
        if self.capture_losses:
            if self.reservoir_length is None:
                raise ValueError("Insufficient arguments for boundary condition.")
            p.update(capture_losses=1.0, reservoir_length=self.reservoir_length, K_q=self.K_q)
        return "storage_curve", p
