"""LumpedStorage: 0-D reservoir behind a fixed_depth boundary
(reference: src/hydromodel/lumped_storage.py:8-179).

Two device forms: the configuration the bundled cases use (constant surface area, no reservoir
rating curve, no entrance losses), where the mass-balance root (lumped_storage.py:24-35) is
Y_new = Y_old + vol_in / area clamped at min_stage (FS_BC_STORAGE); and the general one (area
curve, power / polynomial outflow rating curve, friction + empirical entrance losses), where the
kernel runs the Brent iteration itself (FS_BC_STORAGE_CURVE).  The expansion loss needs an A_str
that the reference's Boundary never passes (boundary.py:119-121), so it is always zero there too."""
import numpy as np

from . import hydraulics


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

    # ---- head loss between the storage and the channel end (lumped_storage.py:47-143) ---------------------------------
    # three parts, each with its two derivatives: friction over reservoir_length, sudden expansion into a stream of area
    # A_str (zero without one - and the reference's Boundary never passes one), an empirical K_q V^2 / 2g
    def friction_loss(self, A_ent, Q, n, R):
        return hydraulics.Sf(A=A_ent, Q=Q, n=n, R=R) * self.reservoir_length

    def dhf_dA(self, A_ent, Q, n, R, dR_dA):
        return hydraulics.dSf_dA(A=A_ent, Q=Q, n=n, R=R, dR_dA=dR_dA) * self.reservoir_length

    def dhf_dQ(self, A_ent, Q, n, R):
        return hydraulics.dSf_dQ(A=A_ent, Q=Q, n=n, R=R) * self.reservoir_length

    def dhl_dn(self, A_ent, Q, n, R):
        """lumped_storage.py:76-81 calls hydraulics.dSf_dn, which the reference's hydraulics module does not define: with
        capture_losses set the call raises AttributeError there, and so it does here"""
        if not self.capture_losses:
            return 0
        return hydraulics.dSf_dn(A=A_ent, Q=Q, n=n, R=R) * self.reservoir_length

    @staticmethod
    def _velocity_head(A_ent, Q):
        V = Q / A_ent
        return V, V ** 2 / (2 * hydraulics.g)

    def expansion_loss(self, A_ent, Q, A_str=None):
        if A_str is None:
            return 0
        return (1 - A_ent / A_str) ** 2 * self._velocity_head(A_ent, Q)[1]

    def d_h_exp_dA(self, A_ent, Q, A_str=None):
        if A_str is None:
            return 0
        V, _ = self._velocity_head(A_ent, Q)
        ratio = 1 - A_ent / A_str
        return (ratio ** 2 * 2 * V * (-Q / A_ent ** 2) + V ** 2 * 2 * ratio * (-1 / A_str)) / (2 * hydraulics.g)

    def d_h_exp_dQ(self, A_ent, Q, A_str=None):
        """lumped_storage.py:119-128 differentiates V with -Q/A^2 here (the A-derivative), reproduced"""
        if A_str is None:
            return 0
        V, _ = self._velocity_head(A_ent, Q)
        return (1 - A_ent / A_str) ** 2 * 2 * V * (-Q / A_ent ** 2) / (2 * hydraulics.g)

    def empirical_loss(self, Q, A_ent):
        return self.K_q * self._velocity_head(A_ent, Q)[1]

    def d_h_emp_dA(self, A_ent, Q):
        V, _ = self._velocity_head(A_ent, Q)
        return self.K_q * 2 * V * (-Q / A_ent ** 2) / (2 * hydraulics.g)

    def d_h_emp_dQ(self, A_ent, Q):
        V, _ = self._velocity_head(A_ent, Q)
        return self.K_q * 2 * V * (1. / A_ent) / (2 * hydraulics.g)

    def energy_loss(self, entry_area, flow, roughness, hydraulic_radius, A_str=None):
        if not self.capture_losses:
            return 0
        return (self.friction_loss(A_ent=entry_area, Q=flow, n=roughness, R=hydraulic_radius)
                + self.expansion_loss(A_ent=entry_area, A_str=A_str, Q=flow) + self.empirical_loss(A_ent=entry_area, Q=flow))

    def dhl_dA(self, entry_area, flow, roughness, hydraulic_radius, dR_dA, A_str=None):
        if not self.capture_losses:
            return 0
        return (self.dhf_dA(A_ent=entry_area, Q=flow, n=roughness, R=hydraulic_radius, dR_dA=dR_dA)
                + self.d_h_exp_dA(A_ent=entry_area, Q=flow, A_str=A_str) + self.d_h_emp_dA(A_ent=entry_area, Q=flow))

    def dhl_dQ(self, entry_area, flow, roughness, hydraulic_radius, A_str=None):
        if not self.capture_losses:
            return 0
        return (self.dhf_dQ(A_ent=entry_area, Q=flow, n=roughness, R=hydraulic_radius)
                + self.d_h_exp_dQ(A_ent=entry_area, Q=flow, A_str=A_str) + self.d_h_emp_dQ(A_ent=entry_area, Q=flow))

    def dA_dY(self, stage):
        """slope of the area curve at `stage` (np.gradient of the table, no beta shift: lumped_storage.py:159-163)"""
        if self.area_curve is None:
            return 0
        return self.alpha * np.interp(stage, self.area_curve[:, 0], self.area_gradient)

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
