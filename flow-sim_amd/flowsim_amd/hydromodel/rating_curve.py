"""RatingCurve: stage-discharge relation Q(stage) as power law or quadratic
(reference: src/hydromodel/rating_curve.py:4-162).

Plugin contract: `device_spec()` returns (kind, params) understood by the C ABI (FS_BC_RATING_POWER /
_POLY / _BLEND) and the boundary row is then evaluated inside the kernel.  A subclass with its own Python
`discharge` either overrides device_spec() (cases/gerd_roseires, smooth gates) or lets it raise
NotImplementedError: PreissmannSolver.run then evaluates that boundary on the host once per Newton
iteration (FS_BC_HOST_ROW) - any object with discharge(stage, time) / dQ_dz(stage, time) runs."""
import numpy as np


class RatingCurve:
    def __init__(self):
        self.function = None
        self.derivative = None
        self.defined = False
        self.type = None

    def set(self, type, a, b, c=None, stage_shift=None):
        # reference quirk (rating_curve.py:11-13): a non-None stage_shift argument is ignored
        if stage_shift is None:
            self.stage_shift = 0
        if type == 'polynomial':
            if c is None:
                raise ValueError("Insufficient arguments. c must be specified.")
            self.a, self.b, self.c = a, b, c
        elif type == 'power':
            self.a, self.b = a, b
        else:
            raise ValueError("Invalid type.")
        self.function = None
        self.derivative = None
        self.defined = True
        self.type = type

    def _need(self):
        if not self.defined:
            raise ValueError("Rating curve is undefined.")

    def discharge(self, stage, time=None):
        self._need()
        if self.function is not None:
            return self.function(stage)
        x = stage + self.stage_shift
        if self.type == 'polynomial':
            return self.a * x ** 2 + self.b * x + self.c
        return self.a * x ** self.b

    def dQ_dz(self, stage, time=None):
        x = stage + self.stage_shift
        self._need()
        if self.type == 'polynomial':
            return self.derivative(x) if self.function is not None else self.a * 2 * x + self.b
        return self.a * self.b * x ** (self.b - 1)

    def stage(self, discharge, trial_stage=None, time=None, tolerance=1e-2, rate=1):
        """Newton inversion of the curve (rating_curve.py:65-82)."""
        self._need()
        z = -self.stage_shift * 1.05 if trial_stage is None else trial_stage
        q = self.discharge(stage=z, time=time)
        while abs(q - discharge) > tolerance:
            z += -rate * (q - discharge) / self.dQ_dz(stage=z, time=time)
            q = self.discharge(stage=z, time=time)
        return z

    def fit(self, discharges, stages, stage_shift=0, type='polynomial', scale=True, degree=2):
        """Least-squares fit of either form (rating_curve.py:84-130)."""
        self.type = type
        Qs = np.asarray(discharges, dtype=np.float64)
        Ys = np.asarray(stages, dtype=np.float64)
        if Qs.size < 3:
            raise ValueError("Need at least 3 points.")
        if Qs.shape != Ys.shape:
            raise ValueError("Q and Y lists should have the same lengths.")
        self.stage_shift = stage_shift
        x = Ys + stage_shift
        if any(x <= 0):
            raise ValueError("All (stage - base) values must be positive for power-law fitting.")
        if type == 'polynomial':
            if scale:
                self.function = np.polynomial.polynomial.Polynomial.fit(x=x, y=Qs, deg=degree)
                self.derivative = self.function.deriv()
            else:
                if degree != 2:
                    print("WARNING: Polynomial degree defaults to 2 for unscaled fitting.")
                a, b, c = np.polyfit(x, Qs, deg=2)
                self.a, self.b, self.c = float(a), float(b), float(c)
        elif type == 'power':
            b, log_a = np.polyfit(np.log(x), np.log(Qs), deg=1)
            self.a, self.b = float(np.exp(log_a)), float(b)
        else:
            raise ValueError("Invalid rating curve type.")
        self.defined = True

    def tostring(self):
        self._need()
        s = str(self.stage_shift)
        if self.type == 'polynomial':
            if self.function is not None:
                return str(self.function)
            return f"{self.a} (Y+{s})^2 + {self.b} (Y+{s}) + {self.c}"
        return f"{self.a} (Y+{s})^{self.b}"

    # ---- device path ---------------------------------------------------------------------
    def device_spec(self, bed_level):
        """(kind name, params dict) for flowsim_amd.BoundarySpec."""
        self._need()
        if type(self).discharge is not RatingCurve.discharge or type(self).dQ_dz is not RatingCurve.dQ_dz:
            raise NotImplementedError(
                f"{type(self).__name__} overrides discharge()/dQ_dz() in Python and has no device_spec() "
                "(power, polynomial or smooth blend of two quadratics): evaluated on the host (FS_BC_HOST_ROW)")
        if self.function is not None:
            # fit(..., scale=True): discharge() evaluates the fitted polynomial at the STAGE, dQ_dz() its derivative at
            # stage + stage_shift (rating_curve.py:51-52, :139-141).  The device row uses one abscissa for both, so only a
            # fit without shift has a device form; any other callable (or a shifted fit) is evaluated on the host.
            poly = self.function
            if not isinstance(poly, np.polynomial.Polynomial) or self.stage_shift != 0:
                raise NotImplementedError("a rating curve given as a Python callable (or a scaled fit with a stage shift) "
                                          "has no device form: evaluated on the host (FS_BC_HOST_ROW)")
            coef = poly.convert().coef          # fitted, scaled polynomial -> plain a x^2 + b x + c
            if len(coef) > 3:
                raise NotImplementedError("fitted rating polynomials above degree 2 have no device form: evaluated on the host")
            c = list(coef) + [0.0] * (3 - len(coef))
            return "poly", dict(a=c[2], b=c[1], c=c[0], stage_shift=0.0, bed_level=bed_level)
        if self.type == 'polynomial':
            return "poly", dict(a=self.a, b=self.b, c=self.c, stage_shift=self.stage_shift, bed_level=bed_level)
        return "power", dict(a=self.a, b=self.b, stage_shift=self.stage_shift, bed_level=bed_level)
