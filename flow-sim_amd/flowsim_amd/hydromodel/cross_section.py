"""Cross-sections on the host: the trapezoid family of the reference
(src/hydromodel/cross_section.py:549-846), polyline sections (IrregularSection, :207-543) and the
distance-weighted interpolation between two sections (cross_section.py:857-968).

Geometry is kept as plain parameter records; `section_table()` turns a list of sections into the
[param][node] structure-of-arrays block the kernel reads (include/flowsim_abi.h, FS_GEO_*), and
`props()` evaluates area / perimeter / top width / conveyance for whole arrays of nodes at once -
used for initial conditions and post-processing, never inside the Newton loop.

Polyline sections are evaluated here only for set-up work (initial conditions, diagnostics); inside
the Newton loop the kernel walks the same polylines itself (csrc/fs_poly.hpp, FS_SEC_IRREGULAR).
"""
from abc import ABC, abstractmethod

import numpy as np
from scipy.optimize import brentq

from . import hydraulics

GEO_ROWS = ("z_bed", "b_main", "m_main", "n_main", "n_left", "n_right", "is_compound", "h_bf",
            "b_fp_l", "b_fp_r", "m_fp", "curvature")


def props(geo, hw):
    """A, P, R, T at water level(s) hw for parameter arrays `geo` (cross_section.py:623-679).
    Over bank the total area leaves out the column above the main channel while the top width
    keeps it - the reference's behaviour, reproduced on purpose."""
    z, b, m = geo["z_bed"], geo["b_main"], geo["m_main"]
    d = np.maximum(0.0, hw - z)
    T = b + 2.0 * m * d
    A = (b + T) / 2.0 * d
    P = b + 2.0 * d * np.sqrt(1.0 + m * m)
    over = (geo["is_compound"] > 0.5) & (d > geo["h_bf"])
    if np.any(over):
        hb, mf = geo["h_bf"], geo["m_fp"]
        dfp = d - hb
        Tb = b + 2.0 * m * hb
        sf = np.sqrt(1.0 + mf * mf)
        A_o = (b + Tb) / 2.0 * hb + (geo["b_fp_l"] + 0.5 * mf * dfp) * dfp + (geo["b_fp_r"] + 0.5 * mf * dfp) * dfp
        P_o = (b + 2.0 * hb * np.sqrt(1.0 + m * m)) + (geo["b_fp_l"] + dfp * sf) + (geo["b_fp_r"] + dfp * sf)
        T_o = (geo["b_fp_l"] + Tb + geo["b_fp_r"]) + 2.0 * mf * dfp
        A, P, T = np.where(over, A_o, A), np.where(over, P_o, P), np.where(over, T_o, T)
    with np.errstate(divide="ignore", invalid="ignore"):
        R = np.where(P > 0.0, A / P, 0.0)
    dry = d <= 0.0
    if np.any(dry):
        A, P, R, T = (np.where(dry, 0.0, v) for v in (A, P, R, T))
    return A, P, R, T, over


def _pw(x, p):
    """x ** p as the reference's scalar code gets it: for single values (what the section objects pass, one node at a time)
    Python's pow, i.e. libm.  numpy's array loop rounds about one result in twenty differently (1 ulp) - harmless by itself, but
    the brentq behind normal_depth then stops one ulp away and an unstable GVF march (explicit steps on a steep reach) blows
    that up to percents.  Whole-ensemble arrays (flowsim_amd.ensemble) keep the vectorised loop."""
    x = np.asarray(x, dtype=np.float64)
    if x.size > 8:
        return np.power(x, p)
    with np.errstate(all="ignore"):
        return np.array([float(v) ** p if v >= 0.0 else np.nan for v in x.ravel()], dtype=np.float64).reshape(x.shape)


def conveyance(geo, hw, pr=None):
    """Total conveyance; compound sections sum K^1.5 over left / main / right (cross_section.py:741-754)."""
    A, P, R, T, over = pr if pr is not None else props(geo, hw)
    with np.errstate(divide="ignore", invalid="ignore"):
        K = A * _pw(R, 2.0 / 3.0) / geo["n_main"]
        comp = geo["is_compound"] > 0.5
        if np.any(comp):
            z, b, m, hb, mf = geo["z_bed"], geo["b_main"], geo["m_main"], geo["h_bf"], geo["m_fp"]
            d = np.maximum(0.0, hw - z)
            dfp = np.where(over, d - hb, 0.0)
            Tb = b + 2.0 * m * hb
            sf = np.sqrt(1.0 + mf * mf)
            A_m = (b + Tb) / 2.0 * hb + Tb * dfp
            P_m = b + 2.0 * hb * np.sqrt(1.0 + m * m)
            A_l = (geo["b_fp_l"] + 0.5 * mf * dfp) * dfp; P_l = geo["b_fp_l"] + dfp * sf
            A_r = (geo["b_fp_r"] + 0.5 * mf * dfp) * dfp; P_r = geo["b_fp_r"] + dfp * sf
            k = lambda a, n, p: a * _pw(np.where(p > 0, a / np.where(p > 0, p, 1.0), 0.0), 2.0 / 3.0) / n
            K_m = np.where(over, k(A_m, geo["n_main"], P_m), K)
            K_l = np.where(over, k(A_l, geo["n_left"], P_l), 0.0)
            K_r = np.where(over, k(A_r, geo["n_right"], P_r), 0.0)
            K = np.where(comp, _pw(_pw(K_l, 1.5) + _pw(K_m, 1.5) + _pw(K_r, 1.5), 2.0 / 3.0), K)
    return K


def equivalent_n(geo, hw, pr=None, K=None):
    """Manning n that reproduces K with the total A and R (cross_section.py:710-739)."""
    A, P, R, T, over = pr if pr is not None else props(geo, hw)
    n = np.array(np.broadcast_to(geo["n_main"], np.shape(A)), dtype=np.float64)
    comp = geo["is_compound"] > 0.5
    if np.any(comp):
        K = conveyance(geo, hw, (A, P, R, T, over)) if K is None else K
        with np.errstate(divide="ignore", invalid="ignore"):
            neq = A * _pw(R, 2.0 / 3.0) / K
        n = np.where(comp & (A > 0) & (R > 0) & (K > 0), neq, n)
    return n


def area_top(sections, geo, level):
    """Wetted area and top width at water levels `level` [..., N] for the node sections: the array
    formulas above for trapezoid nodes, a per-node polyline walk otherwise (diagnostics and the
    host fallback of the post-processing only)."""
    level = np.asarray(level, dtype=np.float64)
    with np.errstate(all="ignore"):
        A, P, R, T, _ = props({n: v for n, v in geo.items() if n in GEO_ROWS}, level)
    A, T = np.array(A, dtype=np.float64), np.array(T, dtype=np.float64)
    for i, s in enumerate(sections):
        if isinstance(s, IrregularSection):
            for idx in np.ndindex(level.shape[:-1]):
                a, _, t = s._walk(s.x, s.z, float(level[idx + (i,)]))
                A[idx + (i,)], T[idx + (i,)] = a, t
    return A, T


def section_table(sections):
    """list of sections -> dict of arrays: the rows of the FS_GEO_* table and, when any node is a
    polyline, irr_x / irr_z [N, P] (rows padded with their last vertex), irr_npts [N] (0 for
    trapezoid nodes) and irr_limits [N, 2] as fs_batch_set_geometry_irregular takes them."""
    poly = [isinstance(s, IrregularSection) for s in sections]

    def f(get, get_poly=lambda s: 0.0):
        return np.array([get_poly(s) if p else get(s) for s, p in zip(sections, poly)], dtype=np.float64)
    tab = dict(
        z_bed=f(lambda s: s.z_bed, lambda s: s.z_min), b_main=f(lambda s: s.b_main), m_main=f(lambda s: s.m_main),
        n_main=f(lambda s: s.n_main, lambda s: s.n_main), n_left=f(lambda s: s.n_left, lambda s: s.n_left),
        n_right=f(lambda s: s.n_right, lambda s: s.n_right),
        is_compound=f(lambda s: 1.0 if s._is_compound else 0.0),
        h_bf=f(lambda s: s.bankfull_depth if s._is_compound else 0.0),
        b_fp_l=f(lambda s: s.b_fp_left), b_fp_r=f(lambda s: s.b_fp_right), m_fp=f(lambda s: s.m_fp),
        curvature=f(lambda s: s.curvature, lambda s: s.curvature))
    if any(poly):
        P = max(s.x.size for s, p in zip(sections, poly) if p)
        X = np.zeros((len(sections), P)); Z = np.zeros((len(sections), P))
        cnt = np.zeros(len(sections), dtype=np.int32)
        lim = np.zeros((len(sections), 2))
        for i, (s, p) in enumerate(zip(sections, poly)):
            if p:
                cnt[i] = s.x.size
                X[i, :cnt[i]], Z[i, :cnt[i]] = s.x, s.z
                X[i, cnt[i]:], Z[i, cnt[i]:] = s.x[-1], s.z[-1]
                lim[i] = (s.left_fp_limit, s.right_fp_limit)
        tab.update(irr_x=X, irr_z=Z, irr_npts=cnt, irr_limits=lim)
    return tab


class CrossSection(ABC):
    """Common interface (cross_section.py:6-202)."""

    def __init__(self, n=None, bed_slope=None, curvature=0.0):
        self.n_left = self.n_main = self.n_right = n
        self.left_fp_limit = self.right_fp_limit = 0.0
        self.curvature = curvature
        self.bed_slope = bed_slope

    @property
    @abstractmethod
    def z_min(self):
        ...

    @property
    @abstractmethod
    def width(self):
        """total width of the section"""

    @abstractmethod
    def properties(self, hw):
        """(A, P, R, T) at water level hw"""

    @abstractmethod
    def get_equivalent_n(self, hw):
        """Manning n that reproduces the section's conveyance with its total A and R"""

    @abstractmethod
    def conveyance(self, hw):
        ...

    @abstractmethod
    def dK_dA(self, hw):
        ...

    @abstractmethod
    def dR_dA(self, hw):
        ...

    @abstractmethod
    def z_at(self, x):
        """bed elevation at the lateral coordinate x"""

    def area(self, hw):
        return self.properties(hw)[0]

    def wetted_perimeter(self, hw):
        return self.properties(hw)[1]

    def hydraulic_radius(self, hw):
        return self.properties(hw)[2]

    def top_width(self, hw):
        return self.properties(hw)[3]

    def dA_dh(self, hw):
        return self.top_width(hw)

    def get_roughness_para(self):
        return (self.n_left, self.n_main, self.n_right, self.left_fp_limit, self.right_fp_limit)

    def set_roughness_para(self, parameters):
        self.n_left, self.n_main, self.n_right, self.left_fp_limit, self.right_fp_limit = parameters

    def friction_slope(self, h, Q):
        return hydraulics.Sf(Q=Q, K=self.conveyance(hw=h + self.z_min))

    def dSf_dA(self, h, Q):
        """one contiguous wetted channel (cross_section.py:124-131); IrregularSection sums over sub-channels"""
        hw = h + self.z_min
        return hydraulics.dSf_dA(Q=Q, K=self.conveyance(hw=hw), dK_dA=self.dK_dA(hw=hw))

    def dSf_dQ(self, h, Q):
        return hydraulics.dSf_dQ(Q=Q, K=self.conveyance(hw=h + self.z_min))

    def _bend(self, h):
        """what the three curvature-slope methods hand to hydraulics: level, n_eq, (A, P, R, T), bend radius"""
        hw = h + self.z_min
        return hw, self.get_equivalent_n(hw=hw), self.properties(hw), 1.0 / self.curvature

    def curvature_slope(self, h, Q):
        if self.curvature == 0:
            return 0.0
        _, n, (A, P, R, T), rc = self._bend(h)
        return hydraulics.Sc(h=h, T=T, A=A, Q=Q, n=n, R=R, rc=rc)

    def dSc_dA(self, h, Q):
        """dSc/dA times dA/dh (cross_section.py:154-164: the product is what Channel.dSe_dA adds to dSf/dA)"""
        if abs(self.curvature) <= 1e-12:
            return 0.0
        hw, n, (A, P, R, T), rc = self._bend(h)
        return hydraulics.dSc_dA(h=h, A=A, Q=Q, n=n, R=R, rc=rc, dR_dA=self.dR_dA(hw=hw), T=T) * self.dA_dh(hw=hw)

    def dSc_dQ(self, h, Q):
        if abs(self.curvature) <= 1e-12:
            return 0.0
        _, n, (A, P, R, T), rc = self._bend(h)
        return hydraulics.dSc_dQ(h=h, T=T, A=A, Q=Q, n=n, R=R, rc=rc)

    def normal_flow(self, hw):
        if self.bed_slope is None or self.bed_slope <= 0.0:
            return 0.0
        return hydraulics.normal_flow(bed_slope=self.bed_slope, K=self.conveyance(hw=hw))

    def normal_depth(self, Q_target, hw_max=None):
        """Depth at which normal_flow == Q_target (cross_section.py:184-202; same bracket, same root finder)."""
        z = self.z_min
        hw_max = z + 100 if hw_max is None else hw_max
        f = lambda hw: Q_target - self.normal_flow(hw=hw)
        try:
            return brentq(f, z, hw_max) - z
        except ValueError:
            if f(z) < 0:
                return 0.0
            if f(hw_max) > 0:
                return hw_max - z
            return 0.0


class TrapezoidalSection(CrossSection):
    """Rectangle / simple trapezoid / compound trapezoid with trapezoidal flood plains
    (cross_section.py:549-613 for the parameters)."""

    def __init__(self, z_bed, b_main, m_main, n_main, z_bank=None, b_fp_left=0.0, b_fp_right=0.0, m_fp=0.0,
                 n_left=0.03, n_right=0.03, **kwargs):
        super().__init__(n=n_main, **kwargs)
        self.z_bed, self.b_main, self.m_main = float(z_bed), float(b_main), float(m_main)
        self._is_compound = z_bank is not None
        if self._is_compound:
            self.z_bank = float(z_bank)
            if self.z_bank <= self.z_bed:
                raise ValueError("Bank elevation z_bank must be above bed z_bed")
            self.b_fp_left, self.b_fp_right, self.m_fp = float(b_fp_left), float(b_fp_right), float(m_fp)
            self.bankfull_depth = self.z_bank - self.z_bed
            self.T_main_at_bank = self.b_main + 2.0 * self.m_main * self.bankfull_depth
            lim = self.T_main_at_bank / 2.0
        else:
            self.z_bank = None
            self.b_fp_left = self.b_fp_right = self.m_fp = 0.0
            lim = np.inf
        self._is_rect = (not self._is_compound) and self.m_main == 0.0
        self._width = np.inf
        self.set_roughness_para((n_left, n_main, n_right, -lim, lim))

    @property
    def z_min(self):
        return self.z_bed

    @property
    def width(self):
        return self._width

    def _geo(self):
        return {k: v[0:1] for k, v in section_table([self]).items()}

    def properties(self, hw):
        A, P, R, T, _ = props(self._geo(), np.array([float(hw)]))
        return (float(A[0]), float(P[0]), float(R[0]), float(T[0]))

    def conveyance(self, hw):
        return float(conveyance(self._geo(), np.array([float(hw)]))[0])

    def get_equivalent_n(self, hw):
        return float(equivalent_n(self._geo(), np.array([float(hw)]))[0])

    def dR_dA(self, hw):
        """cross_section.py:766-790."""
        A, P, R, T = self.properties(hw)
        if P <= 0.0 or T <= 0.0:
            return 0.0
        over = self._is_compound and max(0.0, hw - self.z_bed) > self.bankfull_depth
        dP_dh = 2.0 * np.sqrt(1.0 + (self.m_fp if over else self.m_main) ** 2)
        return (P - A * (dP_dh * (1.0 / T))) / (P ** 2)

    def dK_dA(self, hw):
        """cross_section.py:756-764 (n_eq frozen)."""
        A, P, R, T = self.properties(hw)
        if A <= 0.0:
            return 0.0
        return (R ** (2 / 3) + A * 2. / 3. * R ** (2 / 3 - 1) * self.dR_dA(hw)) / self.get_equivalent_n(hw)

    def z_at(self, x):
        """Bed elevation at lateral coordinate x (cross_section.py:795-846)."""
        x = float(x)
        half = self.b_main / 2.0
        if self._is_rect:
            return self.z_bed if abs(x) < half else np.inf
        if not self._is_compound or abs(x) <= self.T_main_at_bank / 2.0:
            return self.z_bed if abs(x) <= half else self.z_bed + (abs(x) - half) / self.m_main
        # flood plains: flat bed of width b_fp_left / b_fp_right, then the outer wall at 1 : m_fp
        beyond = abs(x) - self.T_main_at_bank / 2.0 - (self.b_fp_left if x < 0 else self.b_fp_right)
        return self.z_bank if beyond <= 0 else self.z_bank + beyond / self.m_fp


class IrregularSection(CrossSection):
    """Section given by a polyline of (x, z) stations (cross_section.py:207-543): possibly several
    wetted sub-channels, composite roughness over a left / main / right strip, finite-difference
    dR/dA and dA/dh (dh = 1e-6).  Same numbers as the reference, including what it does with a
    vertex lying exactly on the water surface and with the water's-edge points of temporary
    sub-sections (see csrc/fs_poly.hpp, which evaluates the same thing on the device)."""
    DH = 1e-6

    def __init__(self, x, z, **kwargs):
        super().__init__(**kwargs)
        x = np.ascontiguousarray(x, dtype=float)
        z = np.ascontiguousarray(z, dtype=float)
        if x.shape != z.shape:
            raise ValueError("x and z must have the same shape")
        if x.ndim != 1:
            raise ValueError("x and z must be 1-D arrays")
        order = np.argsort(x, kind="stable")
        self.x, self.z = x[order], z[order]
        self._z_min = float(np.min(self.z))
        self._width = float(self.x[-1] - self.x[0])
        self.left_fp_limit, self.right_fp_limit = self.x[0], self.x[-1]

    @property
    def z_min(self):
        return self._z_min

    @property
    def width(self):
        return self._width

    @staticmethod
    def _walk(x, z, hw):
        """(A, P, T) of the polyline (x, z) below stage hw, all edges at once: an edge counts in
        full when both ends are wet, is cut at the surface when one end is wet and the other
        strictly above, and is skipped otherwise (cross_section.py:262-322)."""
        if x.size < 2:
            return 0.0, 0.0, 0.0
        x0, x1, z0, z1 = x[:-1], x[1:], z[:-1], z[1:]
        d0, d1 = hw - z0, hw - z1
        w0, w1 = d0 > 0.0, d1 > 0.0
        full = w0 & w1
        cut_l = w1 & (z0 > hw)
        cut_r = w0 & (z1 > hw)
        with np.errstate(divide="ignore", invalid="ignore"):
            xi = x0 + (hw - z0) / (z1 - z0) * (x1 - x0)           # where the edge meets the surface
        dx = np.where(full, x1 - x0, np.where(cut_l, x1 - xi, np.where(cut_r, xi - x0, 0.0)))
        dz = np.where(full, z1 - z0, np.where(cut_l, d1, np.where(cut_r, d0, 0.0)))
        mean_d = np.where(full, 0.5 * (d0 + d1), np.where(cut_l, 0.5 * d1, np.where(cut_r, 0.5 * d0, 0.0)))
        dx = np.where(full | cut_l | cut_r, dx, 0.0)
        return float(np.sum(mean_d * dx)), float(np.sum(np.sqrt(dx * dx + dz * dz))), float(np.sum(dx))

    def properties(self, hw):
        A, P, T = self._walk(self.x, self.z, float(hw))
        return (A, P, A / P if P > 0.0 else 0.0, T)

    def get_subchannels(self, hw):
        """Wetted runs of at least two stations with their water's-edge points (cross_section.py:330-370)."""
        wet = np.concatenate(([False], self.z < hw, [False]))
        starts = np.flatnonzero(wet[1:] & ~wet[:-1])
        ends = np.flatnonzero(~wet[1:] & wet[:-1])               # one past the last wet station
        out = []
        n = self.x.size
        for s, e in zip(starts, ends):
            if e - s < 2:
                continue
            xs, zs = self.x[s:e], self.z[s:e]
            if s > 0 and self.z[s - 1] > hw:
                # the reference calls np.interp with a decreasing abscissa here, which lands on x[s]
                xs, zs = np.r_[self.x[s], xs], np.r_[hw, zs]
            if e < n and self.z[e - 1] < hw and self.z[e] > hw:
                xe = (self.x[e] - self.x[e - 1]) / (self.z[e] - self.z[e - 1]) * (hw - self.z[e - 1]) + self.x[e - 1]
                xs, zs = np.r_[xs, xe], np.r_[zs, hw]
            out.append({"x": xs, "z": zs})
        return out

    def _strip_K(self, lo, hi, n_val, hw):
        m = (self.x >= lo) & (self.x <= hi)
        A, P, _ = self._walk(self.x[m], self.z[m], hw)
        return hydraulics.conveyance(A=A, n=n_val, R=A / P) if (A > 0 and P > 0) else 0.0

    def get_equivalent_n(self, hw):
        """Horton-Einstein composite of the three strips (cross_section.py:449-503)."""
        A, P, R, _ = self.properties(hw)
        if A <= 0 or P <= 0:
            return self.n_main
        K = (self._strip_K(self.x[0], self.left_fp_limit, self.n_left, hw) ** 1.5
             + self._strip_K(self.left_fp_limit, self.right_fp_limit, self.n_main, hw) ** 1.5
             + self._strip_K(self.right_fp_limit, self.x[-1], self.n_right, hw) ** 1.5) ** (2.0 / 3.0)
        return self.n_main if K <= 0.0 else (A * R ** (2.0 / 3.0)) / K

    def conveyance(self, hw):
        A, P, R, _ = self.properties(hw)
        return 0.0 if A <= 0.0 else hydraulics.conveyance(A=A, n=self.get_equivalent_n(hw), R=R)

    def dR_dA(self, hw, dh=DH):
        A1, _, R1, _ = self.properties(hw - dh)
        A2, _, R2, _ = self.properties(hw + dh)
        return 0.0 if A2 - A1 == 0.0 else (R2 - R1) / (A2 - A1)

    def dA_dh(self, hw, dh=DH):
        return (self.area(hw + dh) - self.area(hw - dh)) / (2 * dh)

    def dK_dA(self, hw):
        A, P, R, _ = self.properties(hw)
        if A <= 0.0:
            return 0.0
        return (R ** (2 / 3) + A * 2. / 3. * R ** (2 / 3 - 1) * self.dR_dA(hw)) / self.get_equivalent_n(hw)

    def _channel_K(self, hw):
        """Conveyance the friction slope uses: the whole section, or the 1.5-power sum over the
        temporary sub-sections when the surface splits it (cross_section.py:372-392)."""
        subs = self.get_subchannels(hw)
        if len(subs) <= 1:
            return self.conveyance(hw)
        total = 0.0
        for sc in subs:
            part = IrregularSection(x=sc["x"], z=sc["z"])
            part.set_roughness_para(self.get_roughness_para())
            total += part.conveyance(hw) ** 1.5
        return total ** (2.0 / 3.0)

    def friction_slope(self, h, Q):
        return hydraulics.Sf(Q=Q, K=self._channel_K(h + self.z_min))

    def _channel_K_and_slope(self, hw):
        """(K, dK/dA) of the 1.5-power composite over the sub-channels, None when the surface does not split the section
        (cross_section.py:394-420): K = S^(2/3), dK/dA = (2/3) S^(-1/3) sum(1.5 K_j^0.5 dK_j/dA), S = sum K_j^1.5"""
        subs = self.get_subchannels(hw)
        if len(subs) <= 1:
            return None
        S = dS = 0.0
        for sc in subs:
            part = IrregularSection(x=sc["x"], z=sc["z"])
            part.set_roughness_para(self.get_roughness_para())
            Kj = part.conveyance(hw=hw)
            S += Kj ** 1.5
            dS += 1.5 * Kj ** 0.5 * part.dK_dA(hw=hw)
        return S ** (2.0 / 3.0), (2.0 / 3.0) * S ** (-1.0 / 3.0) * dS

    def dSf_dA(self, h, Q):
        both = self._channel_K_and_slope(h + self.z_min)
        if both is None:
            return super().dSf_dA(h, Q)
        return hydraulics.dSf_dA(Q=Q, K=both[0], dK_dA=both[1])

    def dSf_dQ(self, h, Q):
        hw = h + self.z_min
        if len(self.get_subchannels(hw)) <= 1:
            return super().dSf_dQ(h, Q)
        return hydraulics.dSf_dQ(Q=Q, K=self._channel_K(hw))

    def z_at(self, x):
        return np.interp(x, self.x, self.z, left=self.z[0], right=self.z[-1])


def interpolate_cross_section(xs1: CrossSection, xs2: CrossSection, dist1: float, dist2: float) -> CrossSection:
    """Section at a point between xs1 (dist1 away) and xs2 (dist2 away): every trapezoid
    parameter, the three roughness values, bed slope and curvature are weighted by the opposite
    distance (cross_section.py:857-930)."""
    total = dist1 + dist2
    if total < 1e-9 or dist1 < 1e-9:
        return xs1
    if dist2 < 1e-9:
        return xs2
    w1, w2 = dist2 / total, dist1 / total
    mix = lambda a, b: a * w1 + b * w2
    if not (isinstance(xs1, TrapezoidalSection) and isinstance(xs2, TrapezoidalSection)):
        # at least one polyline: blend bed elevations over the union of the polyline stations
        # (cross_section.py:932-968); a trapezoid neighbour is sampled through its z_at()
        stations = [s.x for s in (xs1, xs2) if isinstance(s, IrregularSection)]
        if not stations:
            raise TypeError("Cannot interpolate: no x-coordinates found.")
        xm = stations[0] if len(stations) == 1 else np.union1d(stations[0], stations[1])
        zm = np.array([xs1.z_at(v) for v in xm]) * w1 + np.array([xs2.z_at(v) for v in xm]) * w2
        slope = None if (xs1.bed_slope is None or xs2.bed_slope is None) else mix(xs1.bed_slope, xs2.bed_slope)
        new = IrregularSection(x=xm, z=zm, n=mix(xs1.n_main, xs2.n_main), bed_slope=slope,
                               curvature=mix(xs1.curvature, xs2.curvature))
        new.set_roughness_para((mix(xs1.n_left, xs2.n_left), mix(xs1.n_main, xs2.n_main), mix(xs1.n_right, xs2.n_right),
                                mix(xs1.left_fp_limit, xs2.left_fp_limit), mix(xs1.right_fp_limit, xs2.right_fp_limit)))
        return new
    y1 = (xs1.z_bank - xs1.z_bed) if xs1._is_compound else 0.0
    y2 = (xs2.z_bank - xs2.z_bed) if xs2._is_compound else 0.0
    z_bed = mix(xs1.z_bed, xs2.z_bed)
    y_bank = mix(y1, y2)
    slope = None if (xs1.bed_slope is None or xs2.bed_slope is None) else mix(xs1.bed_slope, xs2.bed_slope)
    return TrapezoidalSection(
        z_bed=z_bed, b_main=mix(xs1.b_main, xs2.b_main), m_main=mix(xs1.m_main, xs2.m_main),
        z_bank=(z_bed + y_bank) if y_bank > 1e-6 else None,
        b_fp_left=mix(xs1.b_fp_left, xs2.b_fp_left), b_fp_right=mix(xs1.b_fp_right, xs2.b_fp_right),
        m_fp=mix(xs1.m_fp, xs2.m_fp), n_main=mix(xs1.n_main, xs2.n_main), n_left=mix(xs1.n_left, xs2.n_left),
        n_right=mix(xs1.n_right, xs2.n_right), bed_slope=slope, curvature=mix(xs1.curvature, xs2.curvature))
