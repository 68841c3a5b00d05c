"""Solver: grid sizing, the depth/flow[nt, N] history, accessors and post-processing shared by
the schemes (reference: src/hydromodel/solver.py:10-329).

`prepare_results` takes its derived fields from the elementwise HIP kernel behind fs_batch_derive
(numpy fallback for a failed run); `save_results` writes the reference's sheets and text summary."""
from abc import ABC, abstractmethod

import numpy as np

from . import cross_section as XS
from . import hydraulics
from .channel import Channel


class Solver(ABC):
    def __init__(self, channel: Channel, time_step, spatial_step, simulation_time, regularization: bool = False,
                 fit_spatial_step: bool = True):
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
        self.eps = 1e-4

    def fit_spatial_step(self):
        self.number_of_nodes = round(self.channel.length / self.spatial_step) + 1
        self.spatial_step = self.channel.length / (self.number_of_nodes - 1)

    @abstractmethod
    def run(self, verbose: int = 1):
        ...

    def initialize_t0(self):
        self.depth[0, :] = self.channel.initial_conditions[:, 0]
        self.flow[0, :] = self.channel.initial_conditions[:, 1]

    # ---- accessors (solver.py:244-329) --------------------------------------------------------
    # The `regularization` / `chi_scaling` arguments select the reference's dry-bed regularisation (a smoothed area floor
    # and a flow damped by chi = A_reg / (A_reg + A_min)).  That branch cannot run in the reference: it looks the floor
    # area up with Channel.area_at(i=..., h=...) - a parameter Channel.area_at does not have - and hands A_reg an `eps` it
    # does not take (solver.py:266, :283, :313), so it raises TypeError at the first evaluation.  The calls are kept as
    # they are there, and fail the same way; nothing in the kernel depends on them.
    H_MIN = 1e-4

    def _k(self, k):
        return self.time_level if k is None else self.time_level - 1 if k == -1 else k

    def depth_at(self, k=None, i=None, regularization=None):
        if i is None:
            raise ValueError("Spatial node must be specified.")
        return self.depth[self._k(k), i]

    def flow_at(self, k=None, i=None, chi_scaling=None):
        if i is None:
            raise ValueError("Spatial node must be specified.")
        k = self._k(k)
        Q = self.flow[k, i]
        if self.regularization if chi_scaling is None else chi_scaling:
            A_reg = self.area_at(k=k, i=i, regularization=True)
            A_min = self.channel.area_at(i=i, h=self.H_MIN)
            Q = Q * (A_reg / (A_reg + A_min))
        return Q

    def water_level_at(self, k=None, i=None, regularization=None):
        return self.channel.bed_level_at(i=i) + self.depth_at(k=k, i=i, regularization=regularization)

    def area_at(self, k=None, i=None, regularization=None):
        if i is None:
            raise ValueError("Spatial node must be specified.")
        A = self.channel.area_at(i=i, hw=self.water_level_at(k=k, i=i))
        if self.regularization if regularization is None else regularization:
            A_min = self.channel.area_at(i=i, hw=self.channel.bed_level_at(i=i) + self.H_MIN)
            A = self.A_reg(A=A, eps=A_min)
        return A

    def Se_at(self, k=None, i=None, regularization=None, chi_scaling=None):
        return self.channel.Se(h=self.depth_at(k=k, i=i, regularization=regularization),
                               Q=self.flow_at(k=k, i=i, chi_scaling=chi_scaling), i=i)

    def dA_dh(self, k=None, i=None, regularization=None):
        return self.channel.dA_dh(i=i, hw=self.water_level_at(k=k, i=i, regularization=regularization))

    def A_reg(self, A):
        """smoothed max(A, A_min): A_min + ((A - A_min) + sqrt((A - A_min)^2 + eps^2)) / 2  (solver.py:298-321)"""
        A_min = self.channel.area_at(i=0, h=self.H_MIN)
        excess = A - A_min
        return A_min + 0.5 * (excess + np.sqrt(excess ** 2 + self.eps ** 2))

    def Q_eff(self, Q, A_reg):
        """flow damped towards zero as the area approaches its floor (solver.py:323-329)"""
        A_min = self.channel.area_at(i=0, h=self.H_MIN)
        return Q * (A_reg / (A_reg + A_min))

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
            # level 0 stage = initial interface stage minus the entrance losses there (solver.py:100-108)
            Y0 = self.level[0, -1]
            xs_end = self.channel.xs_at_node[-1]
            loss0 = st.energy_loss(entry_area=self.area[0, -1], flow=self.flow[0, -1],
                                   roughness=xs_end.get_equivalent_n(hw=Y0), hydraulic_radius=xs_end.hydraulic_radius(hw=Y0))
            st.stage_hydrograph.insert(0, [0, Y0 - loss0])
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
        A, T = XS.area_top(self.channel.xs_at_node, geo, self.level)
        self.area, self.top_width = A, T
        self.froude_number = hydraulics.froude_array(T, A, self.flow)
        self.velocity = self.flow / self.area
        self.wave_celerity = self.velocity + np.sqrt(hydraulics.g * self.area / self.top_width)
        self.amplitude = self.depth - self.depth[0, :]
        self.peak_amplitude = self.amplitude.max(axis=0)

    def save_results(self, folder_path: str, file_name: str = None) -> None:
        """Result tables + text summary (solver.py:129-233).  The reference writes one .xlsx workbook
        through pandas/openpyxl; when openpyxl is not installed the same sheets go to a .npz archive
        (one array per sheet) next to the identical .txt summary.  Windows path separators in
        `folder_path` (the reference's case scripts use them) are accepted."""
        import os
        folder_path = folder_path.replace("\\", os.sep)
        os.makedirs(folder_path, exist_ok=True)
        file_name = 'results.xlsx' if file_name is None else file_name
        file_path = os.path.join(folder_path, file_name)
        sheets = {"Level": self.level, "Flow": self.flow, "Depth": self.depth, "Velocity": self.velocity,
                  "Area": self.area, "Top width": self.top_width, "Wave celerity": self.wave_celerity,
                  "Amplitude": self.amplitude, "Froude number": self.froude_number}
        nt = self.depth.shape[0]
        time = np.arange(nt) * self.time_step
        distance = np.array(self.channel.ch_at_node, dtype=np.float64)
        rows = {"Peak amplitude": self.peak_amplitude[None, :], "Bed level": self.bed_profile[None, :]}
        storage = self.channel.downstream_boundary.lumped_storage is not None
        try:
            import openpyxl  # noqa: F401
            import pandas as pd
            with pd.ExcelWriter(file_path, engine="openpyxl") as writer:
                for name, arr in sheets.items():
                    df = pd.DataFrame(arr, index=time, columns=distance)
                    df.index.name, df.columns.name = "Time", "Distance"
                    df.to_excel(writer, sheet_name=name)
                if storage:
                    df = pd.DataFrame({"outflow": self.storage_outflow}, index=time)
                    df.index.name = "Time"
                    df.to_excel(writer, sheet_name="Outflow")
                for name, arr in rows.items():
                    df = pd.DataFrame(arr, columns=distance, index=[name])
                    df.columns.name = "Distance"
                    df.to_excel(writer, sheet_name=name)
        except ImportError:
            extra = {"Outflow": self.storage_outflow, "Reservoir stage": self.storage_stage} if storage else {}
            np.savez(os.path.splitext(file_path)[0] + ".npz", Time=time, Distance=distance,
                     **{k.replace(" ", "_"): v for k, v in {**sheets, **rows, **extra}.items()})
        with open(os.path.splitext(file_path)[0] + '.txt', 'w') as out:
            out.write(self.summary())

    def summary(self) -> str:
        """The text block of the reference's results file (solver.py:188-233): steps, duration, mass
        imbalance, peak attenuation and median-volume travel time."""
        from .utility import seconds_to_hms
        q_in, q_out = np.asarray(self.flow)[:, 0], np.asarray(self.flow)[:, -1]
        imbalance = np.sum(q_in - q_out) * self.time_step
        pct = float(imbalance / self.time_step / np.sum(q_in)) * 100
        peak_in, peak_out = np.max(q_in), np.max(q_out)

        def median_time(q):
            cum = np.concatenate([[0.0], np.cumsum(q)[:-1]])          # sum(q[:i]) for i = 0..n-1
            return int(np.argmax(cum >= 0.5 * cum[-1])) * self.time_step
        t_in, t_out = median_time(q_in), median_time(q_out)
        lines = [f'Spatial step = {self.spatial_step} m', f'Time step = {self.time_step} s',
                 f'Simulation duration = {seconds_to_hms(self.total_sim_duration)}',
                 f'Mass imbalance (total inflow - total outflow) = {imbalance:.2f} m^3 = {pct:.4f}% of inflow.',
                 f'Peak inflow = {peak_in:.2f} m^3/s', f'Peak outflow = {peak_out:.2f} m^3/s',
                 f'Attenuation = {(peak_in - peak_out) / peak_in * 100:.2f}%',
                 f'Median volume entry time = {seconds_to_hms(t_in)}',
                 f'Median volume arrival time = {seconds_to_hms(t_out)}',
                 f'Median volume travel time = {seconds_to_hms(t_out - t_in)}']
        return "\n".join(lines) + "\n"

    def _finalize(self, verbose):
        self._solved = True
        self.total_sim_duration = self.time_level * self.time_step
        self.prepare_results()
        if verbose >= 1:
            print("Simulation completed successfully.")
