"""Boundary: one end of a channel and the condition imposed there
(reference: src/hydromodel/boundary.py:7-247).

The five condition kinds and their arguments are the reference's; evaluation of the boundary row
happens on the device, so this class only validates, keeps the data and flattens itself into the
(kind, params, target table) triple of the C ABI (`device_spec`)."""
from .hydrograph import Hydrograph
from .lumped_storage import LumpedStorage
from .rating_curve import RatingCurve

CONDITIONS = ('flow_hydrograph', 'fixed_depth', 'normal_depth', 'rating_curve', 'stage_hydrograph')


class Boundary:
    def __init__(self, condition: str, chainage, bed_level: float = None, initial_depth: float = None,
                 rating_curve: RatingCurve = None, hydrograph: Hydrograph = None):
        if condition not in CONDITIONS:
            raise ValueError("Invalid boundary condition.")
        self.condition = condition
        self.cross_section = None            # set by Channel (channel.py:240-241)
        self.bed_level = bed_level
        self.initial_depth = initial_depth
        self.initial_stage = None if initial_depth is None else bed_level + initial_depth
        self.chainage = chainage
        self.rating_curve = rating_curve
        self.hydrograph = hydrograph
        self.lumped_storage = None

    def set_lumped_storage(self, lumped_storage: LumpedStorage):
        self.lumped_storage = lumped_storage

    def condition_type(self) -> bool:
        """True when the boundary equation fixes Q, False when it fixes the depth (boundary.py:244-247)."""
        return self.condition in ('flow_hydrograph', 'normal_depth', 'rating_curve')

    def device_spec(self, n_levels, dt):
        """(kind name, params, target[n_levels] or None) - same argument checks as
        Boundary.condition_residual (boundary.py:82-102)."""
        c = self.condition
        if c in ('flow_hydrograph', 'stage_hydrograph'):
            if self.hydrograph is None:
                raise ValueError("Insufficient arguments for boundary condition.")
            tgt = self.hydrograph.sample(n_levels, dt)
            if c == 'flow_hydrograph':
                return "flow", {}, tgt
            if self.bed_level is None:
                raise ValueError("Insufficient arguments for boundary condition.")
            return "stage", dict(bed_level=self.bed_level), tgt
        if c == 'normal_depth':
            xs = self.cross_section
            if xs is None or xs.bed_slope is None or self.bed_level is None:
                raise ValueError("Insufficient arguments for boundary condition.")
            return "normal", dict(bed_slope=xs.bed_slope, bed_level=self.bed_level), None
        if c == 'rating_curve':
            if self.rating_curve is None or self.bed_level is None:
                raise ValueError("Insufficient arguments for boundary condition.")
            kind, p = self.rating_curve.device_spec(self.bed_level)
            return kind, p, None
        if self.lumped_storage is None:
            if self.initial_depth is None:
                raise ValueError("Insufficient arguments for boundary condition.")
            return "fixed", dict(initial_depth=self.initial_depth), None
        kind, p = self.lumped_storage.device_spec(self.bed_level)
        return kind, p, None
