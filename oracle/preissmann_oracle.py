"""CPU oracle for the batched Preissmann Newton step.

TEST INFRASTRUCTURE ONLY.  A numpy restatement (vectorised over the nodes of ONE reach, fp64) of
the algorithm in the reference's hot path; it exists so that `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg have an independent checker.  The product path (the HIP
kernels behind the C-ABI in flow-sim_amd/csrc) never imports, links or calls anything in oracle/.

Pinned by: tests/test_oracle_golden.py checks every function here against the fixtures in
tests/golden/*.npz, which oracle/gen_golden.py produced by running the reference itself
(cve-mohd/flow-sim @ 2026-02-13) in the build container.  The reference has no tests of its own
(SURVEY.md section 4), so those fixtures are the only golden vectors that exist.

Reference map (paths relative to the reference repository root):
  section_props        <- src/hydromodel/cross_section.py:623-679 (properties), :681-708, :792-793
  conveyance/equiv_n   <- cross_section.py:710-754, src/hydromodel/hydraulics.py:15-26
  dR_dA / dK_dA        <- cross_section.py:756-790, hydraulics.py:28-40
  energy_slope         <- src/hydromodel/channel.py:53-105, cross_section.py:114-175,
                          hydraulics.py:42-204
  boundary_eval        <- src/hydromodel/boundary.py:56-242, rating_curve.py:32-63,:132-147,
                          lumped_storage.py:24-45
  assemble             <- src/hydromodel/preissmann.py:61-81, :220-320, :322-344, :407-798, :899-910
  newton_run           <- preissmann.py:101-177 (incl. the "stored iterate is the pre-update one"
                          semantics)
The third-party arithmetic the reference calls is scipy.sparse.linalg.spsolve (SuperLU; scipy
unpinned in the reference's requirements.txt:2, 1.15.3 here); this file calls the same routine.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

G = 9.80665  # scipy.constants.g, hydraulics.py:2

GEO_KEYS = ("z_bed", "b_main", "m_main", "n_main", "n_left", "n_right", "is_compound", "h_bf",
            "b_fp_l", "b_fp_r", "m_fp", "curvature")
# polyline nodes (IrregularSection, cross_section.py:207-543): NaN-padded stations/elevations
# [N, P], vertex counts [N] (0 = trapezoid-family node) and roughness strip limits [N, 2]
IRR_KEYS = ("irr_x", "irr_z", "irr_npts", "irr_limits")


# ------------------------------------------------------------------------------------------------
# problem description (plain data)
# ------------------------------------------------------------------------------------------------
@dataclass
class BC:
    """One boundary (boundary.py:10-46) flattened to data."""
    kind: str                       # flow_hydrograph | stage_hydrograph | fixed_depth | normal_depth | rating_curve
    bed_level: Optional[float] = None
    target: Optional[np.ndarray] = None     # hydrograph pre-sampled at k*dt, [nt]
    initial_depth: Optional[float] = None
    bed_slope: Optional[float] = None       # of the boundary node's section (boundary.py:79)
    # rating curve (rating_curve.py:10-63): 'power' | 'polynomial' | 'blend'
    rc_type: Optional[str] = None
    rc: dict = field(default_factory=dict)
    # lumped storage behind fixed_depth (lumped_storage.py:8-45), constant surface area only
    storage: Optional[dict] = None          # {area, min_stage, Y_min, Y_max}


@dataclass
class Problem:
    geo: dict                       # GEO_KEYS -> [N]
    h0: np.ndarray
    Q0: np.ndarray
    us: BC
    ds: BC
    theta: float
    dt: float
    dx: float
    nt: int                         # number of time levels incl. t=0  (solver.py:35)
    tol: float = 1e-4
    max_iter: int = 100

    @property
    def N(self):
        return len(self.h0)


# ------------------------------------------------------------------------------------------------
# trapezoidal section family
# ------------------------------------------------------------------------------------------------
def _pow(x, p):
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.power(x, p)


def section_props(geo, h):
    """(A, P, R, T, over) for depth h at every node.  cross_section.py:623-679.

    Over bank the reference's total area omits the water column above the main channel while the
    top width includes it; kept as is."""
    b, m = geo["b_main"], geo["m_main"]
    comp = geo["is_compound"] > 0.5
    d = np.maximum(0.0, h)
    T = b + 2.0 * m * d
    A = (b + T) / 2.0 * d
    P = b + 2.0 * d * np.sqrt(1.0 + m * m)
    over = comp & (d > geo["h_bf"])
    if np.any(over):
        hb = geo["h_bf"]
        dfp = d - hb
        Tb = b + 2.0 * m * hb
        A_main = (b + Tb) / 2.0 * hb
        P_main = b + 2.0 * hb * np.sqrt(1.0 + m * m)
        sfp = np.sqrt(1.0 + geo["m_fp"] ** 2)
        A_l = (geo["b_fp_l"] + 0.5 * geo["m_fp"] * dfp) * dfp
        P_l = geo["b_fp_l"] + dfp * sfp
        A_r = (geo["b_fp_r"] + 0.5 * geo["m_fp"] * dfp) * dfp
        P_r = geo["b_fp_r"] + dfp * sfp
        A = np.where(over, A_main + A_l + A_r, A)
        P = np.where(over, P_main + P_l + P_r, P)
        T = np.where(over, (geo["b_fp_l"] + Tb + geo["b_fp_r"]) + 2.0 * geo["m_fp"] * dfp, T)
    with np.errstate(divide="ignore", invalid="ignore"):
        R = np.where(P > 0.0, A / P, 0.0)
    dry = d <= 0.0
    if np.any(dry):
        A, P, R, T = (np.where(dry, 0.0, v) for v in (A, P, R, T))
    return A, P, R, T, over


def _k(A, n, R):
    return A * _pow(R, 2.0 / 3.0) / n          # hydraulics.py:15-26


def conveyance(geo, h, props=None):
    """Horton-Einstein style sum over sub-sections for compound shapes. cross_section.py:741-754."""
    A, P, R, T, over = props if props is not None else section_props(geo, h)
    comp = geo["is_compound"] > 0.5
    K = _k(A, geo["n_main"], R)
    if np.any(comp):
        # inside the banks the reference still goes through the 1.5 / (2/3) power round trip
        Km = K
        if np.any(over):
            b, m, hb = geo["b_main"], geo["m_main"], geo["h_bf"]
            d = np.maximum(0.0, h)
            dfp = d - hb
            Tb = b + 2.0 * m * hb
            A_m = (b + Tb) / 2.0 * hb + Tb * dfp                         # :694 (includes the column)
            P_m = b + 2.0 * hb * np.sqrt(1.0 + m * m)
            sfp = np.sqrt(1.0 + geo["m_fp"] ** 2)
            A_l = (geo["b_fp_l"] + 0.5 * geo["m_fp"] * dfp) * dfp
            P_l = geo["b_fp_l"] + dfp * sfp
            A_r = (geo["b_fp_r"] + 0.5 * geo["m_fp"] * dfp) * dfp
            P_r = geo["b_fp_r"] + dfp * sfp
            with np.errstate(divide="ignore", invalid="ignore"):
                R_m = np.where(P_m > 0, A_m / P_m, 0.0)
                R_l = np.where(P_l > 0, A_l / P_l, 0.0)
                R_r = np.where(P_r > 0, A_r / P_r, 0.0)
            K_l = np.where(over, _k(A_l, geo["n_left"], R_l), 0.0)
            K_r = np.where(over, _k(A_r, geo["n_right"], R_r), 0.0)
            Km = np.where(over, _k(A_m, geo["n_main"], R_m), K)
        else:
            K_l = np.zeros_like(K)
            K_r = np.zeros_like(K)
        Kc = _pow(_pow(K_l, 1.5) + _pow(Km, 1.5) + _pow(K_r, 1.5), 2.0 / 3.0)
        K = np.where(comp, Kc, K)
    return K


def equivalent_n(geo, h, props=None, K=None):
    """cross_section.py:710-739."""
    A, P, R, T, over = props if props is not None else section_props(geo, h)
    comp = geo["is_compound"] > 0.5
    n = geo["n_main"].copy() if isinstance(geo["n_main"], np.ndarray) else np.full_like(A, geo["n_main"])
    if np.any(comp):
        K = conveyance(geo, h, (A, P, R, T, over)) if K is None else K
        with np.errstate(divide="ignore", invalid="ignore"):
            neq = A * _pow(R, 2.0 / 3.0) / K
        ok = comp & (A > 0) & (R > 0) & (K > 0)
        n = np.where(ok, neq, n)
    return n


def dR_dA(geo, h, props):
    """cross_section.py:766-790."""
    A, P, R, T, over = props
    dP_dh = 2.0 * np.sqrt(1.0 + geo["m_main"] ** 2)
    dP_dh = np.where(over, 2.0 * np.sqrt(1.0 + geo["m_fp"] ** 2), dP_dh)
    with np.errstate(divide="ignore", invalid="ignore"):
        out = (P - A * (dP_dh * (1.0 / T))) / (P ** 2)
    return np.where((P <= 0) | (T <= 0), 0.0, out)


def dK_dA(geo, h, props, n_eq, dRdA):
    """cross_section.py:756-764 with hydraulics.py:28-40 (n_eq frozen)."""
    A, P, R, T, over = props
    out = (_pow(R, 2.0 / 3.0) + A * 2.0 / 3.0 * _pow(R, 2.0 / 3.0 - 1) * dRdA) / n_eq
    return np.where(A <= 0, 0.0, out)


def froude(T, A, Q):
    """hydraulics.py:155-168 with its three 1e-6 clamps."""
    V = Q / np.maximum(A, 1e-6)
    D = A / np.maximum(T, 1e-6)
    return V / np.sqrt(G * np.maximum(D, 1e-6))


def node_terms(geo, h, Q):
    """Everything the residual/Jacobian needs at the nodes for state (h, Q).

    Returns dict with A, T(=dA_dh), Se, dSe_dA, dSe_dQ where dSe_dA follows the reference's mixed
    convention (channel.py:71-87): friction part per unit AREA, curvature part already multiplied
    by dA/dh (cross_section.py:164)."""
    if "irr_npts" in geo and np.any(geo["irr_npts"] > 0) and not geo.get("_quiet"):
        with np.errstate(all="ignore"):                      # trapezoid formulas are meaningless at polyline nodes
            return node_terms(dict(geo, _quiet=True), h, Q)
    props = section_props(geo, h)
    A, P, R, T, over = props
    K = conveyance(geo, h, props)
    n_eq = equivalent_n(geo, h, props, K)
    dRdA = dR_dA(geo, h, props)
    dKdA = dK_dA(geo, h, props, n_eq, dRdA)
    Sf = Q * np.abs(Q) / K ** 2                                    # hydraulics.py:57
    dSf_dA = -2.0 * Sf * (dKdA / K)                                 # hydraulics.py:75
    dSf_dQ = 2.0 * np.abs(Q) / K ** 2                               # hydraulics.py:92
    Se, dSe_dA, dSe_dQ = Sf, dSf_dA, dSf_dQ
    curv = geo["curvature"]
    if np.any(curv != 0):
        with np.errstate(divide="ignore", invalid="ignore"):
            rc = 1.0 / curv
            Fr = froude(T, A, Q)
            C = _pow(R, 1.0 / 6.0) / n_eq
            f = 8 * G / C ** 2                                      # hydraulics.py:227-229
            sq = np.sqrt(f)
            num = (2.86 * sq + 2.07 * f) * h ** 2 * Fr ** 2
            den = (0.565 + sq) * rc ** 2
            Sc = num / den                                          # hydraulics.py:94-117
            # hydraulics.py:119-137
            dh_dA = 1.0 / T
            V = Q / A
            D = A / T
            dFr_dA = -0.5 * V * (G * D) ** (-1.5) * G * (1.0 / T) + (-Q / A ** 2) * (G * D) ** (-0.5)
            df_dA = -(8.0 / 3.0) * G * n_eq ** 2 * _pow(R, -4.0 / 3.0) * dRdA
            dnum = (2.86 / (2 * sq) * df_dA + 2.07 * df_dA) * h ** 2 * Fr ** 2 \
                + (2.86 * sq + 2.07 * f) * (2 * h * dh_dA * Fr ** 2 + h ** 2 * 2 * Fr * dFr_dA)
            dden = (1.0 / (2 * sq) * df_dA) * rc ** 2
            dSc_dA = (dnum * den - num * dden) / den ** 2 * T       # x dA_dh, cross_section.py:164
            # hydraulics.py:139-153
            dFr_dQ = (1.0 / A) * (G * D) ** (-0.5)
            dnumq = (2.86 * sq + 2.07 * f) * h ** 2 * 2 * Fr * dFr_dQ
            dSc_dQ = (dnumq * den) / den ** 2
        Se = Sf + np.where(curv != 0, Sc, 0.0)                      # ==0 guard, cross_section.py:145
        small = np.abs(curv) <= 1e-12                               # <=1e-12 guard, :156,:168
        dSe_dA = dSf_dA + np.where(small, 0.0, dSc_dA)
        dSe_dQ = dSf_dQ + np.where(small, 0.0, dSc_dQ)
    out = dict(A=A, T=T, Se=Se, dSe_dA=dSe_dA, dSe_dQ=dSe_dQ, K=K, dKdA=dKdA, P=P, R=R, n_eq=n_eq,
               top_width=T)
    if "irr_npts" in geo and np.any(geo["irr_npts"] > 0):
        _splice_irregular(geo, h, Q, out)
    return out


def _splice_irregular(geo, h, Q, out):
    """Overwrites the entries of polyline nodes with oracle/irregular_oracle.py (scalar per node);
    for those nodes T is the finite-difference dA/dh the solver uses (solver.py:295-296) and
    top_width the geometric one."""
    from . import irregular_oracle as IO
    out.update({k: np.array(v, dtype=np.float64, copy=True) for k, v in out.items()})
    for i in np.nonzero(geo["irr_npts"] > 0)[0]:
        c = int(geo["irr_npts"][i])
        rough = (geo["n_left"][i], geo["n_main"][i], geo["n_right"][i], *geo["irr_limits"][i])
        t = IO.node_terms(geo["irr_x"][i, :c], geo["irr_z"][i, :c], rough, float(geo["curvature"][i]),
                          float(h[i]), float(Q[i]))
        for k, v in t.items():
            out[k][i] = v


def _one(geo, i):
    i = i % len(geo["z_bed"])
    g = {k: np.atleast_1d(np.asarray(v, dtype=np.float64)[i]) for k, v in geo.items() if k in GEO_KEYS}
    if "irr_npts" in geo:
        g.update({k: np.asarray(geo[k])[i:i + 1] for k in IRR_KEYS})
    return g


# ------------------------------------------------------------------------------------------------
# boundary conditions
# ------------------------------------------------------------------------------------------------
def rating_discharge(bc: BC, stage):
    """rating_curve.py:32-63 and the flattened smooth Roseires curve
    (cases/gerd_roseires/roseires_rating_curve.py:65-109)."""
    rc = bc.rc
    if bc.rc_type == "power":
        return rc["a"] * (stage + rc.get("shift", 0.0)) ** rc["b"]
    if bc.rc_type == "polynomial":
        x = stage + rc.get("shift", 0.0)
        return rc["a"] * x ** 2 + rc["b"] * x + rc["c"]
    if bc.rc_type == "blend":
        s0, buf = rc["initial_stage"], rc["buffer"]
        if stage >= s0 + buf:
            al = 1.0
        elif stage <= s0:
            al = 0.0
        else:
            s = (stage - s0) / buf
            al = 3 * s ** 2 - 2 * s ** 3
        lo = rc["low"][0] + rc["low"][1] * stage + rc["low"][2] * stage * stage
        hi = rc["high"][0] + rc["high"][1] * stage + rc["high"][2] * stage * stage
        return (1.0 - al) * lo + al * hi
    raise ValueError("Rating curve is undefined.")


def rating_dQ_dz(bc: BC, stage):
    """rating_curve.py:132-147; blend: central difference, roseires_rating_curve.py:202-208."""
    rc = bc.rc
    if bc.rc_type == "power":
        return rc["a"] * rc["b"] * (stage + rc.get("shift", 0.0)) ** (rc["b"] - 1)
    if bc.rc_type == "polynomial":
        return rc["a"] * 2 * (stage + rc.get("shift", 0.0)) + rc["b"]
    dY = rc.get("dY", 1e-3)
    return (rating_discharge(bc, stage + dY) - rating_discharge(bc, stage - dY)) / (2 * dY)


def storage_area_at(st, Y):
    """lumped_storage.py:152-157."""
    if st.get("curve") is None:
        return st["area"]
    c = st["curve"]
    return st["alpha"] * np.interp(Y + st["beta"], c[:, 0], c[:, 1])


def storage_net_vol(st, Y1, Y2):
    """lumped_storage.py:166-179 (incl. the n-point trapezoid whose n depends on |Y2 - Y1|)."""
    if st.get("curve") is None:
        return (Y2 - Y1) * st["area"]
    c = st["curve"]
    step = np.min(np.abs(c[1:, 0] - c[:-1, 0]))
    n = int(abs(Y2 - Y1) / step)
    if n > 2:
        ys = np.linspace(Y1, Y2, n)
        return np.trapezoid([storage_area_at(st, y) for y in ys], ys)
    return 0.5 * (storage_area_at(st, Y2) + storage_area_at(st, Y1)) * (Y2 - Y1)


def storage_outflow(st, Y):
    rc = st.get("rc")                                                # rating_curve.py:32-63
    if rc is None:
        return 0.0
    x = Y + rc.get("shift", 0.0)
    return rc["a"] * x ** 2 + rc["b"] * x + rc["c"] if rc["type"] == "polynomial" else rc["a"] * x ** rc["b"]


def storage_mass_balance(st, dt, vol_in, Y_old):
    """lumped_storage.py:24-35: the reference's own root finder (scipy.optimize.brentq, default tolerances)."""
    from scipy.optimize import brentq

    def f(Y_new):
        q_out = 0.5 * (storage_outflow(st, Y_old) + storage_outflow(st, Y_new)) if st.get("rc") is not None else 0.0
        return storage_net_vol(st, Y_old, Y_new) - (vol_in - q_out * dt)
    Y = brentq(f, st["Y_min"], st["Y_max"])
    return st["min_stage"] if Y < st["min_stage"] else Y


def boundary_eval(bc: BC, geo_node, h, Q, k, dt, Q_old=None, store=None):
    """(residual, d/dh, d/dQ) of one boundary equation at time level k.  boundary.py:56-242.

    `store` is the mutable storage state {"Y_prev": stage kept for the previous level} of the
    lumped-storage variant; the stage implied by this evaluation is left in store["Y_eval"]."""
    z_min = float(geo_node["z_bed"][0])
    kind = bc.kind
    if kind == "flow_hydrograph":
        return Q - bc.target[k], 0.0, 1.0
    if kind == "stage_hydrograph":
        return h - (bc.target[k] - bc.bed_level), 1.0, 0.0
    if kind == "fixed_depth" and bc.storage is None:
        return h - bc.initial_depth, 1.0, 0.0
    if kind == "normal_depth":
        S0 = bc.bed_slope
        sgn = -1.0 if S0 < 0 else 1.0
        rt = abs(S0) ** 0.5
        hh = np.array([h])
        nt_r = node_terms(geo_node, hh, np.array([Q]))                     # residual: hw = z_min + h
        hd = np.array([h + bc.bed_level - z_min])                          # df_dh: hw = h + bed_level
        nt_d = node_terms(geo_node, hd, np.array([Q]))
        target = sgn * float(nt_r["K"][0]) * rt                            # hydraulics.py:4-13
        return Q - target, 0.0 - sgn * float(nt_d["dKdA"][0]) * rt * float(nt_d["T"][0]), 1.0
    if kind == "rating_curve":
        stage = bc.bed_level + h
        return Q - rating_discharge(bc, stage), 0.0 - rating_dQ_dz(bc, stage), 1.0
    if kind == "fixed_depth":
        st = bc.storage
        vol_in = 0.5 * (Q_old + Q) * dt                                    # preissmann.py:314
        Y_old = (h + bc.bed_level) if k == 1 else store["Y_prev"]          # boundary.py:104-108
        general = st.get("curve") is not None or st.get("rc") is not None or st.get("losses") is not None
        if general:
            Y_new = storage_mass_balance(st, dt, vol_in, Y_old)            # lumped_storage.py:24-35 (brentq)
        else:
            Y_new = Y_old + vol_in / st["area"]        # root of lumped_storage.py:25-28 with :170
            if not (st["Y_min"] <= Y_new <= st["Y_max"]):
                raise ValueError("f(a) and f(b) must have different signs")   # what brentq raises
            if Y_new < st["min_stage"]:
                Y_new = st["min_stage"]
        store["Y_eval"] = Y_new
        dY = 0.0 if Y_new <= st["min_stage"] else 1.0 / storage_area_at(st, Y_new)     # lumped_storage.py:37-45
        hl = dhl_dA = dhl_dQ = 0.0
        dA_dh = 0.0
        if st.get("losses") is not None:                                   # boundary.py:118-124, :152-164, :230-235
            Lr, Kq = st["losses"]["reservoir_length"], st["losses"]["K_q"]
            r = node_terms(geo_node, np.array([h]), np.array([Q]))                          # hw = z_min + depth
            A, R, n = float(r["A"][0]), float(r["R"][0]), float(r["n_eq"][0])
            hl = Q * abs(Q) / _k(A, n, R) ** 2 * Lr + Kq * (Q / A) ** 2 / (2 * G)
            d = node_terms(geo_node, np.array([h + bc.bed_level - z_min]), np.array([Q]))  # hw = depth + bed_level
            A, R, n = float(d["A"][0]), float(d["R"][0]), float(d["n_eq"][0])
            if "irr_npts" in geo_node and int(geo_node["irr_npts"][0]) > 0:     # polyline: central difference, cross_section.py:523-531
                from . import irregular_oracle as IO
                c = int(geo_node["irr_npts"][0])
                dRdA = float(IO.dR_dA(geo_node["irr_x"][0, :c], geo_node["irr_z"][0, :c], h + bc.bed_level))
            else:
                dRdA = float(dR_dA(geo_node, None, section_props(geo_node, np.array([h + bc.bed_level - z_min])))[0])
            K = _k(A, n, R)
            dK = (R ** (2.0 / 3.0) + A * 2.0 / 3.0 * R ** (2.0 / 3.0 - 1) * dRdA) / n
            V = Q / A
            dhl_dA = -2 * (Q * abs(Q) / K ** 2) * (dK / K) * Lr + Kq * 2 * V * (-Q / A ** 2) / (2 * G)
            dhl_dQ = 2 * abs(Q) / K ** 2 * Lr + Kq * 2 * V * (1.0 / A) / (2 * G)
            dA_dh = float(d["T"][0])
        return h - (Y_new + hl - bc.bed_level), 1.0 - dhl_dA * dA_dh, 0.0 - (dY * 0.5 * dt + dhl_dQ)
    raise ValueError("Invalid boundary condition.")


# ------------------------------------------------------------------------------------------------
# residual + Jacobian in the reference's layout
# ------------------------------------------------------------------------------------------------
def assemble(p: Problem, h, Q, h_old, Q_old, k, store=None, old_terms=None):
    """R[2N] (preissmann.py:61-81) and data[8N-4] (preissmann.py:322-344) for iterate (h, Q) at
    level k with level k-1 = (h_old, Q_old)."""
    N, th, dt, dx = p.N, p.theta, p.dt, p.dx
    geo = p.geo
    new = node_terms(geo, h, Q)
    old = old_terms if old_terms is not None else node_terms(geo, h_old, Q_old)
    z = geo["z_bed"]
    A1, A0 = new["A"], old["A"]
    lo, hi = slice(0, N - 1), slice(1, N)

    def tdiff(f1, f0):                                              # preissmann.py:899-900
        return (f1[hi] + f1[lo] - f0[hi] - f0[lo]) / (2 * dt)

    def sdiff(f1, f0):                                              # :902-905
        return th * ((f1[hi] - f1[lo]) / dx) + (1 - th) * ((f0[hi] - f0[lo]) / dx)

    def cavg(f1, f0):                                               # :907-910
        return 0.5 * th * (f1[hi] + f1[lo]) + 0.5 * (1 - th) * (f0[hi] + f0[lo])

    C = tdiff(A1, A0) + sdiff(Q, Q_old)                             # :220-249
    avgA = cavg(A1, A0)
    dYdx = sdiff(z + h, z + h_old)
    avgSe = cavg(new["Se"], old["Se"])
    M = tdiff(Q, Q_old) + sdiff(Q ** 2 / A1, Q_old ** 2 / A0) + G * avgA * (dYdx + avgSe)   # :251-301

    R = np.empty(2 * N)
    rU, uh, uq = boundary_eval(p.us, _one(geo, 0), h[0], Q[0], k, dt)
    rD, dh_, dq_ = boundary_eval(p.ds, _one(geo, N - 1), h[-1], Q[-1], k, dt, Q_old=Q_old[-1], store=store)
    R[0], R[-1] = rU, rD
    R[1:-1:2] = C
    R[2:-1:2] = M

    T = new["T"]
    cq = th / dx
    half_t = 0.5 * th
    data = np.empty(8 * N - 4)
    data[0], data[1] = uh, uq
    blk = data[2:-2].reshape(N - 1, 8)
    blk[:, 0] = T[lo] / (2 * dt)                                    # dC_dh_i   :431-447
    blk[:, 1] = -cq                                                 # dC_dQ_i   :476-491
    blk[:, 2] = T[hi] / (2 * dt)                                    # dC_dh_ip1 :407-422
    blk[:, 3] = cq                                                  # dC_dQ_ip1 :456-471
    for col, sl, s in ((4, lo, -1.0), (6, hi, 1.0)):                # dM_dh_i :558-612 / dM_dh_ip1 :496-550
        Ai, Qi, Ti = A1[sl], Q[sl], T[sl]
        blk[:, col] = -(s * cq) * (Qi / Ai) ** 2 * Ti + G * (
            avgA * (s * cq + half_t * new["dSe_dA"][sl] * Ti) + half_t * Ti * (dYdx + avgSe))
        blk[:, col + 1] = 1 / (2 * dt) + (s * cq) * 2 * Qi / Ai + G * (avgA * (half_t * new["dSe_dQ"][sl]))  # :619-733
    data[-2], data[-1] = dh_, dq_
    return R, data, new


def csr_pattern(N):
    """preissmann.py:874-897."""
    rows = [0, 0]
    cols = [0, 1]
    for r in range(1, 2 * N - 1, 2):
        for rr in (r, r + 1):
            rows += [rr] * 4
            cols += [r - 1, r, r + 1, r + 2]
    rows += [2 * N - 1] * 2
    cols += [2 * N - 2, 2 * N - 1]
    return np.array(rows), np.array(cols)


def newton_run(p: Problem, n_steps=None, trace=False):
    """Time loop x Newton loop, preissmann.py:101-163.  Returns dict(depth, flow [nt,N], iters[nt],
    status, norms).  The row written for level k is the iterate whose residual norm passed the
    test, the updated vector seeds level k+1 (preissmann.py:128,146-154)."""
    N = p.N
    nt = p.nt if n_steps is None else min(p.nt, n_steps + 1)
    depth = np.empty((nt, N))
    flow = np.empty((nt, N))
    depth[0], flow[0] = p.h0, p.Q0
    x = np.empty(2 * N)
    x[0::2], x[1::2] = p.h0, p.Q0
    rows, cols = csr_pattern(N)
    J = None
    iters = np.zeros(nt, dtype=np.int32)
    norms = []
    store = {"Y_prev": None} if p.ds.storage is not None else None
    stages = []
    status = 0
    old_terms = None
    for k in range(1, nt):
        it = 0
        while True:
            it += 1
            if it - 1 >= p.max_iter:
                status = 1
                break
            depth[k], flow[k] = x[0::2], x[1::2]
            R, data, new = assemble(p, depth[k], flow[k], depth[k - 1], flow[k - 1], k, store, old_terms)
            if J is None:
                J = sp.coo_matrix((data, (rows, cols)), shape=(2 * N, 2 * N)).tocsr()
            else:
                J.data[:] = data
            x = x + spla.spsolve(J, -R)
            err = float(np.sum(np.square(R)) ** 0.5)
            if trace:
                norms.append((k, err))
            if not np.isfinite(err):
                status = 2
                break
            if err < p.tol:
                break
        iters[k] = it if status == 0 else it - 1
        if status:
            depth, flow = depth[:k + 1], flow[:k + 1]
            break
        old_terms = new
        if store is not None:
            store["Y_prev"] = store["Y_eval"]
            stages.append(store["Y_eval"])
    return dict(depth=depth, flow=flow, iters=iters, status=status, norms=norms,
                x_next=x, storage_stage=np.array(stages))


# ------------------------------------------------------------------------------------------------
# fixture -> Problem
# ------------------------------------------------------------------------------------------------
def problem_from_fixture(fx, meta, member=None):
    """Builds a Problem from a tests/golden/*.npz written by oracle/gen_golden.py."""
    def pick(name):
        a = fx[name]
        return a[member] if member is not None and a.ndim > 1 else a
    geo = {k: np.array(pick("geo_" + k), dtype=np.float64) for k in GEO_KEYS}
    if "geo_irr_npts" in fx:
        geo.update({k: np.array(fx["geo_" + k]) for k in IRR_KEYS})
    ic = pick("initial_conditions")
    slope = pick("geo_bed_slope")

    def mk(side):
        kind = meta[f"{side}_condition"]
        bc = BC(kind=kind, bed_level=meta.get(f"{side}_bed_level"))
        idx = 0 if side == "us" else -1
        bc.bed_slope = None if np.isnan(slope[idx]) else float(slope[idx])
        if kind in ("flow_hydrograph", "stage_hydrograph"):
            own = f"{side}_target"          # older fixtures: one hydrograph per case, stored as us_target
            bc.target = np.array(pick(own if own in fx else "us_target"), dtype=np.float64)
        bc.initial_depth = meta.get(f"{side}_initial_depth")
        if kind == "rating_curve":
            if f"{side}_rc_type" in meta:
                bc.rc_type = meta[f"{side}_rc_type"]
                bc.rc = dict(a=meta[f"{side}_rc_a"], b=meta[f"{side}_rc_b"], c=meta.get(f"{side}_rc_c"),
                             shift=meta.get(f"{side}_rc_shift", 0.0))
            elif "rating" in meta:
                bc.rc_type, bc.rc = "blend", meta["rating"]
            elif "rc_type" in meta:
                bc.rc_type = meta["rc_type"]
                bc.rc = dict(a=meta["rc_a"], b=meta["rc_b"], c=meta.get("rc_c"), shift=meta.get("rc_shift", 0.0))
            else:
                prm = pick("params")
                bc.rc_type, bc.rc = "power", dict(a=float(prm[6]), b=float(prm[7]), shift=0.0)
        if kind == "fixed_depth" and ("storage_area" in meta or "storage_curve" in meta):
            bc.storage = dict(area=meta.get("storage_area"), min_stage=meta["storage_min_stage"],
                              Y_min=meta["storage_bounds"][0], Y_max=meta["storage_bounds"][1])
            if "storage_curve" in meta:
                bc.storage.update(curve=np.array(meta["storage_curve"], dtype=np.float64),
                                  alpha=meta["storage_alpha"], beta=meta["storage_beta"])
            if meta.get("storage_rc_type"):
                bc.storage["rc"] = dict(type=meta["storage_rc_type"], **meta["storage_rc"])
            if "storage_losses" in meta:
                bc.storage["losses"] = meta["storage_losses"]
        return bc
    return Problem(geo=geo, h0=np.array(ic[:, 0]), Q0=np.array(ic[:, 1]), us=mk("us"), ds=mk("ds"),
                   theta=meta["theta"], dt=meta["dt"], dx=meta["dx"], nt=meta["nt"],
                   tol=meta["tolerance"], max_iter=meta["max_iter"])


def load_fixture(path):
    import json
    fx = np.load(path, allow_pickle=False)
    meta = json.loads(str(fx["meta"]))
    return fx, meta


def sweep_cases(path):
    """(index, arrays, meta) of every case of a multi-case fixture (oracle/gen_random_sweep.py: arrays of case i stored
    as c{i:02d}_<name>, metadata in meta["cases"][i]); each pair goes through problem_from_fixture like a single fixture"""
    fx, meta = load_fixture(path)
    width = 2 if len(meta["cases"]) <= 100 else 4
    for i, m in enumerate(meta["cases"]):
        pre = f"c{i:0{width}d}_"
        yield i, {k[len(pre):]: fx[k] for k in fx.files if k.startswith(pre)}, m
