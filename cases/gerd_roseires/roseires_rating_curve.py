"""Downstream boundary of the case: stage-discharge relation of the Roseires dam gates.

Two quadratic response surfaces are fitted to the operating tables (spillway release as a
function of (stage, gate opening), deep-sluice release as a function of (stage, tail-water
level)); a "closed" gate state passes the initial flow at the initial stage, the "open" state is
everything open, and the discharge blends from one to the other with a smoothstep over `buffer`
metres above the initial stage (semantics of the reference's
cases/gerd_roseires/roseires_rating_curve.py:18-257 with smooth=True, its default).

For a fixed gate state both releases are quadratics in the stage, so the smooth curve has an exact
device form: FS_BC_RATING_BLEND (`device_spec`).

smooth=False is the reference's operated-gate mode (roseires_rating_curve.py:65-78, :111-140): the
gates open once the stage seen by the previous call has risen `0.5 m` above the initial stage and
close once it has fallen 1 m below it, with a cool-down in simulation time between two moves.  That
curve depends on time and on its own history, so it has no device form: device_spec() raises
NotImplementedError and PreissmannSolver.run evaluates the boundary row on the host every Newton
iteration (FS_BC_HOST_ROW)."""
import numpy as np
from scipy.optimize import brentq

from src.hydromodel.rating_curve import RatingCurve

from . import settings
from .inputs import grid_table

HYDROPOWER_Q = 63.0 * 1e6 / (24 * 3600)
NUM_SLUICE_GATES, NUM_SPILLWAYS, MAX_SPILLWAY_OPENING = 5, 7, 13
MIN_STAGE, MAX_STAGE = 466.7, 492
TAIL_WATER_LEVEL_RANGE = [440, 455]


def fit_quadratic_surface(X, y):
    """Ordinary least squares of y on [x0, x1, x0^2, x0 x1, x1^2] with intercept, solved on centred
    data (what a linear regression over degree-2 polynomial features computes)."""
    F = np.column_stack([X[:, 0], X[:, 1], X[:, 0] ** 2, X[:, 0] * X[:, 1], X[:, 1] ** 2])
    Fm, ym = F.mean(axis=0), y.mean()
    coef = np.linalg.lstsq(F - Fm, y - ym, rcond=None)[0]
    return coef, float(ym - Fm @ coef)


def surface(model, x0, x1):
    c, icpt = model
    return float(icpt + c[0] * x0 + c[1] * x1 + c[2] * x0 * x0 + c[3] * x0 * x1 + c[4] * x1 * x1)


def stage_quadratic(model, x1):
    """the same surface at fixed second variable: (c0, c1, c2) of c0 + c1 z + c2 z^2"""
    c, icpt = model
    return np.array([icpt + c[1] * x1 + c[4] * x1 * x1, c[0] + c[3] * x1, c[2]])


class RoseiresRatingCurve(RatingCurve):
    def __init__(self, initial_stage=None, initial_flow=None, initially_open=False, jammed_spillways=0,
                 jammed_sluice_gates=0, max_cooldown=3600 * 5, smooth=True, buffer=0.5, deep_sluices_active=True):
        super().__init__()
        if initial_stage > MAX_STAGE or initial_stage < MIN_STAGE:
            raise ValueError(f"Roseires water stage must be between {MIN_STAGE} m and {MAX_STAGE} m.")
        self.spillway_model = fit_quadratic_surface(*grid_table(settings.spillway_table_path))
        self.sluice_model = fit_quadratic_surface(*grid_table(settings.sluice_table_path))
        self.initial_stage, self.buffer, self.smooth = initial_stage, buffer, bool(smooth)
        self.tail_water_level = float(np.average(TAIL_WATER_LEVEL_RANGE))
        n_sp = NUM_SPILLWAYS - jammed_spillways
        n_sl = NUM_SLUICE_GATES - (jammed_sluice_gates if deep_sluices_active else NUM_SLUICE_GATES)
        self.open_state = ([MAX_SPILLWAY_OPENING] * n_sp + [0] * jammed_spillways, n_sl)
        self.closed_state = self._closed_state(initial_flow, n_sp, n_sl)
        self.defined, self.type = True, "blend"
        # operated gates (smooth=False): position, stage seen by the last evaluation, cool-down clock
        self.open = bool(initially_open)
        self.current_stage = initial_stage
        self.max_cooldown, self.cooldown, self.prev_time = max_cooldown, 0, None

    # ---- releases -----------------------------------------------------------------------------
    def total_release(self, stage, state):
        openings, sluices = state
        q = surface(self.sluice_model, stage, self.tail_water_level) * sluices
        q += sum(surface(self.spillway_model, stage, o) for o in openings if o > 0)
        return q + HYDROPOWER_Q

    def _closed_state(self, initial_flow, n_sp, n_sl):
        """gate setting that passes the initial flow at the initial stage: whole sluices, then whole
        spillway gates, then one partial opening (rounded to the centimetre)"""
        z = self.initial_stage
        full = [MAX_SPILLWAY_OPENING] * n_sp
        sluices = None
        for i in range(1, n_sl + 1):
            sluices = i
            if self.total_release(z, (full, i)) > initial_flow:
                sluices = i - 1
                break
        whole = 0
        for i in range(1, n_sp + 1):
            if self.total_release(z, ([MAX_SPILLWAY_OPENING] * i, sluices)) > initial_flow:
                whole = i - 1
                break
        rest = NUM_SPILLWAYS - whole - 1
        gap = lambda p: initial_flow - self.total_release(z, ([MAX_SPILLWAY_OPENING] * whole + [p] + [0] * rest, sluices))
        partial = round(brentq(gap, 0, MAX_SPILLWAY_OPENING), 2)
        if whole + (1 if partial > 0 else 0) > n_sp:
            raise ValueError("initial flow exceeds what the operable gates can pass")
        return ([MAX_SPILLWAY_OPENING] * whole + [partial] + [0] * rest, sluices)

    def blend_weight(self, stage):
        s0, buf = self.initial_stage, self.buffer
        if stage >= s0 + buf:
            return 1.0
        if stage <= s0:
            return 0.0
        s = (stage - s0) / buf
        return 3 * s ** 2 - 2 * s ** 3

    def gate_control(self, time):
        """moves the gates at most once per cool-down period (roseires_rating_curve.py:111-133)"""
        if self.prev_time is not None:
            self.cooldown = max(0, self.cooldown - (time - self.prev_time))
        self.prev_time = time
        if self.cooldown > 0:
            return
        if not self.open and self.current_stage >= self.initial_stage + 0.5:
            self.cooldown, self.open = self.max_cooldown, True
        elif self.open and self.current_stage <= self.initial_stage - 1:
            self.cooldown, self.open = self.max_cooldown, False

    def discharge(self, stage, time=None, update_stage=True, update_gate_state=True, smooth=None):
        if self.smooth if smooth is None else smooth:
            w = self.blend_weight(stage)
            return (1.0 - w) * self.total_release(stage, self.closed_state) + w * self.total_release(stage, self.open_state)
        if update_gate_state:
            self.gate_control(time=time)
        q = self.total_release(stage, self.open_state if self.open else self.closed_state)
        if update_stage:
            self.current_stage = stage
        return q

    def dQ_dz(self, stage, time=None, dY=0.001):
        hold = dict(time=time, update_stage=False, update_gate_state=False)
        return (self.discharge(stage + dY, **hold) - self.discharge(stage - dY, **hold)) / (2 * dY)

    # ---- device form -----------------------------------------------------------------------------
    def _state_quadratic(self, state):
        openings, sluices = state
        q = sluices * stage_quadratic(self.sluice_model, self.tail_water_level)
        for o in openings:
            if o > 0:
                q = q + stage_quadratic(self.spillway_model, o)
        q[0] += HYDROPOWER_Q
        return q

    def device_spec(self, bed_level):
        if not self.smooth:
            raise NotImplementedError("operated gates (smooth=False) depend on time and on their own history: evaluated on the host")
        lo, hi = self._state_quadratic(self.closed_state), self._state_quadratic(self.open_state)
        return "blend", dict(stage0=self.initial_stage, buffer=self.buffer, lo0=lo[0], lo1=lo[1], lo2=lo[2],
                             hi0=hi[0], hi1=hi[1], hi2=hi[2], dY=0.001, bed_level=bed_level)
