"""Solver: grid sizing, the depth/flow[nt, N] history, accessors and post-processing shared by
the schemes (reference: src/hydromodel/solver.py:10-329).

`save_results` (xlsx writer, needs openpyxl) is out of scope; the derived
fields of `prepare_results` are computed here with numpy over whole [nt, N] arrays."""
from abc import ABC, abstractmethod

import numpy as np

from . import cross_section as XS
from . import hydraulics
from .channel import Channel


class Solver(ABC):
    def __init__(self, channel: Channel, time_step, spatial_step, simulation_time, regularization: bool = False,
                 fit_spatial_step: bool = True):
        if regularization:
            raise NotImplementedError("regularization is dead code in the reference (SURVEY.md F7) and not carried over")
        self.channel = channel
        self.time_step, self.spatial_step = time_step, spatial_step
        self.time_level = 0
        self.number_of_nodes = self.channel.length // self.spatial_step + 1          # solver.py:34
        self.number_of_time_levels = simulation_time // self.time_step + 1           # solver.py:35
        if fit_spatial_step:
            self.fit_spatial_step()
        self.number_of_nodes = int(self.number_of_nodes)
        self.number_of_time_levels = int(self.number_of_time_levels)
        self.channel.initialize_conditions(n_nodes=self.number_of_nodes)
        self.num_celerity = self.spatial_step / self.time_step
        self.flow = np.empty((self.number_of_time_levels, self.number_of_nodes), dtype=np.float64)
        self.depth = np.empty_like(self.flow)
        self._type = None
        self._solved = False
        self.total_sim_duration = 0
        self.regularization = regularization

    def fit_spatial_step(self):
        self.number_of_nodes = round(self.channel.length / self.spatial_step) + 1
        self.spatial_step = self.channel.length / (self.number_of_nodes - 1)

    @abstractmethod
    def run(self, verbose: int = 1):
        ...

    def initialize_t0(self):
        self.depth[0, :] = self.channel.initial_conditions[:, 0]
        self.flow[0, :] = self.channel.initial_conditions[:, 1]

    # ---- accessors (solver.py:244-296) --------------------------------------------------------
    def _k(self, k):
        return self.time_level if k is None else self.time_level - 1 if k == -1 else k

    def depth_at(self, k=None, i=None):
        if i is None:
            raise ValueError("Spatial node must be specified.")
        return self.depth[self._k(k), i]

    def flow_at(self, k=None, i=None):
        if i is None:
            raise ValueError("Spatial node must be specified.")
        return self.flow[self._k(k), i]

    def water_level_at(self, k=None, i=None):
        return self.channel.bed_level_at(i=i) + self.depth_at(k=k, i=i)

    def area_at(self, k=None, i=None):
        if i is None:
            raise ValueError("Spatial node must be specified.")
        return self.channel.area_at(i=i, hw=self.water_level_at(k=k, i=i))

    def Se_at(self, k=None, i=None):
        return self.channel.Se(h=self.depth_at(k=k, i=i), Q=self.flow_at(k=k, i=i), i=i)

    def dA_dh(self, k=None, i=None):
        return self.channel.dA_dh(i=i, hw=self.water_level_at(k=k, i=i))

    # ---- post-processing (solver.py:65-127) ----------------------------------------------------------
    def prepare_results(self) -> None:
        k = self.time_level
        if k + 1 < self.number_of_time_levels:
            self.flow, self.depth = self.flow[:k + 1], self.depth[:k + 1]
        geo = self.channel.node_geometry
        self.bed_profile = np.array(geo["z_bed"], dtype=np.float64)
        dev = getattr(self, "_derived", None)
        if dev is not None and dev["level"].shape == self.depth.shape:
            # computed by the elementwise HIP kernel behind fs_batch_derive right after the run
            for name in ("level", "area", "top_width", "froude_number", "velocity", "wave_celerity", "amplitude",
                         "peak_amplitude"):
                setattr(self, name, dev[name])
        else:
            self.prepare_results_host()
        st = self.channel.downstream_boundary.lumped_storage
        if st is not None:
            # level 0 stage = initial interface stage (no entrance losses in the supported configuration)
            st.stage_hydrograph.insert(0, [0, self.level[0, -1]])
            self.storage_stage = np.array(st.stage_hydrograph, dtype=np.float64)[:, 1].flatten()
            out = np.empty(k + 1)
            out[0] = 0 if st.rating_curve is None else min(self.flow[0, -1], st.rating_curve.discharge(self.storage_stage[0], 0))
            qin = 0.5 * (self.flow[:-1, -1] + self.flow[1:, -1])
            dvol = np.array([st.net_vol_change(Y1=a, Y2=b) for a, b in zip(self.storage_stage[:-1], self.storage_stage[1:])])
            out[1:] = (qin - dvol / self.time_step) * self.flow[1:, -1] / qin
            self.storage_outflow = out

    def prepare_results_host(self):
        """the same fields with numpy (used when no device history is available, e.g. after a failed run)"""
        geo = self.channel.node_geometry
        self.level = self.depth + self.bed_profile
        A, P, R, T, _ = XS.props({n: v[None, :] for n, v in geo.items()}, self.level)
        self.area, self.top_width = A, T
        self.froude_number = hydraulics.froude_array(T, A, self.flow)
        self.velocity = self.flow / self.area
        self.wave_celerity = self.velocity + np.sqrt(hydraulics.g * self.area / self.top_width)
        self.amplitude = self.depth - self.depth[0, :]
        self.peak_amplitude = self.amplitude.max(axis=0)

    def save_results(self, folder_path: str, file_name: str = None) -> None:
        raise NotImplementedError("the xlsx writer is outside the accelerated path (SURVEY.md section 2); "
                                  "read solver.depth / solver.flow and the prepare_results fields instead")

    def _finalize(self, verbose):
        self._solved = True
        self.total_sim_duration = self.time_level * self.time_step
        self.prepare_results()
        if verbose >= 1:
            print("Simulation completed successfully.")
