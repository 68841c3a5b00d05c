"""CPU oracle, polyline ("irregular") cross-sections.  TEST INFRASTRUCTURE ONLY (see
preissmann_oracle.py for the rules: nothing under flow-sim_amd/ imports this).

Restates, as pure functions of plain arrays, what the reference's IrregularSection does
(src/hydromodel/cross_section.py:207-543) together with the CrossSection base-class slopes
(:114-175).  A section is (x[P], z[P], rough) with x ascending and
rough = (n_left, n_main, n_right, left_limit, right_limit) (cross_section.py:105-111).

Pinned by the `probe` table and the full runs in tests/golden/irr_*.npz (oracle/gen_golden.py
case_irregular, produced by running the reference here): tests/test_oracle_irregular.py.

Behaviour of the reference that is kept on purpose:
  * a vertex lying exactly on the water surface is neither wet nor "above": the edge next to it
    contributes nothing (:262, :292, :300);
  * with two or more wetted sub-channels (>= 2 vertices each) the friction slope switches to the
    sum over temporary sub-sections (:372-447); those are built with np.interp on a DEcreasing
    abscissa for the left water's-edge point, which returns the x of the first wet vertex (:357),
    and, evaluated at their own creation stage, lose the edge triangles (first bullet) -- but gain
    them back at stage + 1e-6, which is what their finite-difference dR/dA sees (:523-531);
  * dR/dA and dA/dh are central differences with dh = 1e-6 (:523-538).
"""
from __future__ import annotations

import numpy as np

G = 9.80665
DH = 1e-6


def properties(x, z, hw):
    """(A, P, R, T) of the polyline below stage hw; cross_section.py:248-328, edge by edge.

    Every edge (j, j+1) is one of: both ends wet -> full trapezoid; one end wet and the other
    strictly above the surface -> cut at the surface; anything else -> nothing."""
    x = np.asarray(x, dtype=np.float64); z = np.asarray(z, dtype=np.float64)
    if hw <= np.min(z):
        return 0.0, 0.0, 0.0, 0.0
    d = hw - z
    A = P = T = 0.0
    for j in range(x.size - 1):
        d0, d1 = d[j], d[j + 1]
        w0, w1 = d0 > 0.0, d1 > 0.0
        if w0 and w1:
            dx = x[j + 1] - x[j]
            A += 0.5 * (d0 + d1) * dx
            P += np.sqrt(dx * dx + (z[j + 1] - z[j]) ** 2)
            T += dx
        elif w1 and z[j] > hw:                      # left water's edge, :289-296
            t = (hw - z[j]) / (z[j + 1] - z[j])
            xl = x[j] + t * (x[j + 1] - x[j])
            dx = x[j + 1] - xl
            A += 0.5 * d1 * dx
            P += np.sqrt(dx * dx + (z[j + 1] - hw) ** 2)
            T += dx
        elif w0 and z[j + 1] > hw:                  # right water's edge, :298-305
            t = (hw - z[j]) / (z[j + 1] - z[j])
            xr = x[j] + t * (x[j + 1] - x[j])
            dx = xr - x[j]
            A += 0.5 * d0 * dx
            P += np.sqrt(dx * dx + (hw - z[j]) ** 2)
            T += dx
    R = A / P if P > 0.0 else 0.0
    return float(A), float(P), float(R), float(T)


def subchannels(x, z, hw):
    """Wetted runs with at least two vertices, each with its water's-edge points; :330-370."""
    x = np.asarray(x, dtype=np.float64); z = np.asarray(z, dtype=np.float64)
    wet = z < hw
    n = x.size
    out = []
    i = 0
    while i < n:
        if not wet[i]:
            i += 1
            continue
        s = i
        while i < n and wet[i]:
            i += 1
        e = i
        if e - s < 2:
            continue
        xs, zs = x[s:e], z[s:e]
        if s > 0 and z[s - 1] > hw:
            # np.interp(hw, [z[s-1], z[s]], ...) with a decreasing abscissa: hw > xp[-1] -> fp[-1]
            xs = np.concatenate(([x[s]], xs)); zs = np.concatenate(([hw], zs))
        if e < n and z[e - 1] < hw and z[e] > hw:
            t = (hw - z[e - 1]) / (z[e] - z[e - 1])
            xs = np.concatenate((xs, [x[e - 1] + t * (x[e] - x[e - 1])])); zs = np.concatenate((zs, [hw]))
        out.append((xs, zs))
    return out


def _k(A, n, R):
    return A * R ** (2.0 / 3.0) / n                 # hydraulics.py:15-26


def equivalent_n(x, z, rough, hw):
    """Composite roughness over left / main / right strips; :449-503."""
    x = np.asarray(x, dtype=np.float64); z = np.asarray(z, dtype=np.float64)
    n_l, n_m, n_r, lim_l, lim_r = rough

    def strip(lo, hi, n_val):
        m = (x >= lo) & (x <= hi)
        if m.sum() < 2:
            return 0.0
        A, P, _, _ = properties(x[m], z[m], hw)
        if A <= 0 or P <= 0:
            return 0.0
        return _k(A, n_val, A / P)

    K_l = strip(x[0], lim_l, n_l)
    K_m = strip(lim_l, lim_r, n_m)
    K_r = strip(lim_r, x[-1], n_r)
    A, P, _, _ = properties(x, z, hw)
    if A <= 0 or P <= 0:
        return n_m
    K = (K_l ** 1.5 + K_m ** 1.5 + K_r ** 1.5) ** (2.0 / 3.0)
    if K <= 0.0:
        return n_m
    return (A * (A / P) ** (2.0 / 3.0)) / K


def conveyance(x, z, rough, hw):
    A, P, R, T = properties(x, z, hw)               # :505-513
    if A <= 0.0:
        return 0.0
    return _k(A, equivalent_n(x, z, rough, hw), R)


def dR_dA(x, z, hw):
    A1, _, R1, _ = properties(x, z, hw - DH)        # :523-531
    A2, _, R2, _ = properties(x, z, hw + DH)
    if A2 - A1 == 0.0:
        return 0.0
    return (R2 - R1) / (A2 - A1)


def dA_dh(x, z, hw):
    return (properties(x, z, hw + DH)[0] - properties(x, z, hw - DH)[0]) / (2 * DH)    # :533-538


def dK_dA(x, z, rough, hw):
    A, P, R, T = properties(x, z, hw)               # :515-521 with hydraulics.py:28-40
    if A <= 0.0:
        return 0.0
    n = equivalent_n(x, z, rough, hw)
    return (R ** (2.0 / 3.0) + A * 2.0 / 3.0 * R ** (2.0 / 3.0 - 1) * dR_dA(x, z, hw)) / n


def friction(x, z, rough, h, Q):
    """(Sf, dSf_dA, dSf_dQ, K, dK_dA); :372-447 over base-class :114-143."""
    hw = h + float(np.min(z))
    subs = subchannels(x, z, hw)
    if len(subs) <= 1:
        K = conveyance(x, z, rough, hw)
        dK = dK_dA(x, z, rough, hw)
    else:
        Ks = 0.0; dKs = 0.0
        for xs, zs in subs:
            Kj = conveyance(xs, zs, rough, hw)
            Ks += Kj ** 1.5
            dKs += 1.5 * Kj ** 0.5 * dK_dA(xs, zs, rough, hw)
        K = Ks ** (2.0 / 3.0)
        dK = (2.0 / 3.0) * Ks ** (-1.0 / 3.0) * dKs
    with np.errstate(divide="ignore", invalid="ignore"):
        Sf = Q * abs(Q) / K ** 2                        # hydraulics.py:57
        dSf_dA = -2.0 * Sf * (dK / K)                   # :75
        dSf_dQ = 2.0 * abs(Q) / K ** 2                  # :92
    return Sf, dSf_dA, dSf_dQ, K, dK


def _froude(T, A, Q):
    V = Q / max(A, 1e-6)                               # hydraulics.py:155-168
    D = A / max(T, 1e-6)
    return V / np.sqrt(G * max(D, 1e-6))


def curvature_terms(x, z, rough, curv, h, Q):
    """(Sc, dSc_dA * dA_dh, dSc_dQ) of cross_section.py:145-175 with hydraulics.py:94-153."""
    if curv == 0:
        return 0.0, 0.0, 0.0
    hw = h + float(np.min(z))
    A, P, R, T = properties(x, z, hw)
    n = equivalent_n(x, z, rough, hw)
    rc = 1.0 / curv
    with np.errstate(divide="ignore", invalid="ignore"):
        Fr = _froude(T, A, Q)
        C = R ** (1.0 / 6.0) / n
        f = 8 * G / C ** 2
        sq = np.sqrt(f)
        num = (2.86 * sq + 2.07 * f) * h ** 2 * Fr ** 2
        den = (0.565 + sq) * rc ** 2
        Sc = num / den
        if abs(curv) <= 1e-12:
            return Sc, 0.0, 0.0
        dRdA = dR_dA(x, z, hw)
        V = Q / A
        D = A / T
        dFr_dA = -0.5 * V * (G * D) ** (-1.5) * G * (1.0 / T) + (-Q / A ** 2) * (G * D) ** (-0.5)
        df_dA = -(8.0 / 3.0) * G * n ** 2 * R ** (-4.0 / 3.0) * dRdA
        dnum = (2.86 / (2 * sq) * df_dA + 2.07 * df_dA) * h ** 2 * Fr ** 2 \
            + (2.86 * sq + 2.07 * f) * (2 * h * (1.0 / T) * Fr ** 2 + h ** 2 * 2 * Fr * dFr_dA)
        dden = (1.0 / (2 * sq) * df_dA) * rc ** 2
        dSc_dA = (dnum * den - num * dden) / den ** 2 * dA_dh(x, z, hw)
        dFr_dQ = (1.0 / A) * (G * D) ** (-0.5)
        dSc_dQ = ((2.86 * sq + 2.07 * f) * h ** 2 * 2 * Fr * dFr_dQ * den) / den ** 2
    return Sc, dSc_dA, dSc_dQ


def node_terms(x, z, rough, curv, h, Q):
    """Same keys as preissmann_oracle.node_terms for ONE polyline node (scalars)."""
    hw = h + float(np.min(z))
    A, P, R, T = properties(x, z, hw)
    Sf, dSf_dA, dSf_dQ, K, dK = friction(x, z, rough, h, Q)
    Sc, dSc_dA, dSc_dQ = curvature_terms(x, z, rough, curv, h, Q)
    return dict(A=A, T=dA_dh(x, z, hw), Se=Sf + Sc, dSe_dA=dSf_dA + dSc_dA, dSe_dQ=dSf_dQ + dSc_dQ,
                K=conveyance(x, z, rough, hw), dKdA=dK_dA(x, z, rough, hw), P=P, R=R,
                n_eq=equivalent_n(x, z, rough, hw), top_width=T)
