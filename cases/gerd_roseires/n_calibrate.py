"""python -m cases.gerd_roseires.n_calibrate  -  RMSE of simulated GERD tail-water levels against
gauge readings over a range of main-channel Manning n (the study of the reference's
cases/gerd_roseires/n_calibrate.py:5-74), with all members stepped in ONE device batch."""
import numpy as np

from flowsim_amd.ensemble import run_manning_ensemble

from . import settings
from .model import build

H_target = np.array([497.5, 500, 502, 505, 507, 510])          # m
Q_gauge = np.array([1562.5, 3850, 6000, 10000, 14000, 21000])   # m3/s


def member_setup(n):
    return build(n_main=n, inflow_hyd_path=settings.inflow_hyd_small_path, coords_path=None,
                 inflow_hyd_func=None, sim_duration=None)


def rmse_curve(n_values):
    solvers = [member_setup(float(n)) for n in n_values]          # host set-up per member (GVF profile depends on n)
    lead, sections = solvers[0]
    lead.channel.member_ics = np.stack([s.channel.initial_conditions for s, _ in solvers])
    res = run_manning_ensemble(lead, n_values, tolerance=settings.tolerance)
    if np.any(res["status"] != 0):
        raise ValueError("ensemble member did not converge")
    z0 = sections[0].z_min
    out = []
    for i in range(len(n_values)):
        Qup, hup = res["hydrographs"][:, 1, i], res["hydrographs"][:, 0, i]
        levels = np.interp(x=Q_gauge, xp=Qup, fp=hup + z0)
        out.append(float(np.mean((levels - H_target) ** 2) ** 0.5))
    return out


if __name__ == "__main__":
    ns = np.linspace(0.020, 0.060, 10)
    for n, e in zip(ns, rmse_curve(ns)):
        print(f"{n:.6f},{e:.6f}")
