"""Boundary: one end of a channel and the condition imposed there
(reference: src/hydromodel/boundary.py:7-247).

The five condition kinds and their arguments are the reference's.  A boundary whose data has a device
form (hydrograph tables, normal depth, power / polynomial / blended rating curves, the two storage
forms) flattens itself into the (kind, params, target table) triple of the C ABI (`device_spec`) and its
row is evaluated inside the kernel.  A boundary carrying a Python plugin without such a form - any object
with discharge(stage, time) / dQ_dz(stage, time), a LumpedStorage with a callable rating curve - is
evaluated here, on the host, once per Newton iteration (`condition_residual`, `df_dh`, `df_dQ`: the
reference's own three methods, boundary.py:56-242) and handed to the kernel as FS_BC_HOST_ROW."""
from . import hydraulics
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

    # ---- host evaluation (plugins without a device form) --------------------------------------------
    def _storage_stage_before(self, depth, time, duration):
        """stage the reservoir had at the previous level; at level 1 the current iterate's (boundary.py:104-108)"""
        k = time // duration
        return depth + self.bed_level if k == 1 else self.lumped_storage.stage_hydrograph[int(k) - 2][1]

    def condition_residual(self, depth, flow, time=None, duration=None, vol_in=None):
        """unknown - target of the boundary equation (boundary.py:56-141)"""
        xs = self.cross_section
        hw = xs.z_min + depth
        c = self.condition
        if c == 'flow_hydrograph':
            if time is None:
                raise ValueError("Insufficient arguments for boundary condition.")
            return flow - self.hydrograph.get_at(time=time)
        if c == 'normal_depth':
            return flow - hydraulics.normal_flow(bed_slope=xs.bed_slope, K=xs.conveyance(hw=hw))
        if c == 'rating_curve':
            return flow - self.rating_curve.discharge(stage=self.bed_level + depth, time=time)
        if c == 'stage_hydrograph':
            if time is None:
                raise ValueError("Insufficient arguments for boundary condition.")
            return depth - (self.hydrograph.get_at(time=time) - self.bed_level)
        st = self.lumped_storage
        if st is None:
            return depth - self.initial_depth
        if duration is None or vol_in is None or time is None:
            raise ValueError("Insufficient arguments for boundary condition.")
        stage = st.mass_balance(duration=duration, vol_in=vol_in, Y_old=self._storage_stage_before(depth, time, duration), time=time)
        loss = st.energy_loss(entry_area=xs.area(hw=hw), flow=flow, roughness=xs.get_equivalent_n(hw=hw),
                              hydraulic_radius=xs.hydraulic_radius(hw=hw))
        log = st.stage_hydrograph                       # one [time, stage] row per level, rewritten while the level iterates
        if log and log[-1][0] == time:
            log[-1][1] = stage
        else:
            log.append([time, stage])
        return depth - (stage + loss - self.bed_level)

    def df_dh(self, depth, flow_rate, time=None):
        """boundary.py:143-187"""
        c = self.condition
        if c == 'flow_hydrograph':
            return 0
        xs = self.cross_section
        hw = depth + self.bed_level
        if c == 'stage_hydrograph':
            return 1
        if c == 'normal_depth':
            K_A = xs.dK_dA(hw=hw)
            return 0 - hydraulics.normal_flow(bed_slope=xs.bed_slope, K=K_A) * xs.dA_dh(hw=hw)
        if c == 'rating_curve':
            return 0 - self.rating_curve.dQ_dz(self.bed_level + depth, time=time)
        st = self.lumped_storage
        if st is None:
            return 1
        dloss = st.dhl_dA(entry_area=xs.area(hw=hw), flow=flow_rate, roughness=xs.get_equivalent_n(hw=hw),
                          hydraulic_radius=xs.hydraulic_radius(hw=hw), dR_dA=xs.dR_dA(hw=hw))
        return 1 - dloss * xs.dA_dh(hw=hw)

    def df_dQ(self, depth, flow_rate, duration=None, time=None, vol_in=None):
        """boundary.py:189-242"""
        if self.condition_type():
            return 1
        st = self.lumped_storage
        if self.condition == 'stage_hydrograph' or st is None:
            return 0
        if duration is None or time is None or vol_in is None:
            raise ValueError("Insufficient arguments for boundary condition.")
        xs = self.cross_section
        hw = depth + self.bed_level
        dstage = st.dY_new_dvol_in(duration=duration, vol_in=vol_in, Y_old=self._storage_stage_before(depth, time, duration), time=time)
        dloss = st.dhl_dQ(entry_area=xs.area(hw=hw), flow=flow_rate, roughness=xs.get_equivalent_n(hw=hw),
                          hydraulic_radius=xs.hydraulic_radius(hw=hw))
        return 0 - (dstage * 0.5 * duration + dloss)

    def has_device_form(self, n_levels=2, dt=1.0) -> bool:
        try:
            self.device_spec(n_levels, dt)
            return True
        except NotImplementedError:
            return False

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
