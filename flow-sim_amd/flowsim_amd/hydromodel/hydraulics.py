"""Scalar hydraulic formulas of the public surface (reference: src/hydromodel/hydraulics.py:4-229), used on the host for
initial conditions, boundary set-up, host-evaluated boundary rows and post-processing.  The per-node evaluation inside
the Newton loop lives in csrc/fs_device.hpp; tests/test_api_values.py holds every function here against values the
reference produced (tests/golden/api_values.json).

Signatures follow the reference (a conveyance may be handed over as K, or is built from A, n, R)."""
import numpy as np

g = 9.80665   # scipy.constants.g, hydraulics.py:2


# ---- Manning conveyance and what follows from it --------------------------------------------------------------------
def conveyance(A: float, n: float, R: float) -> float:
    """K = A R^(2/3) / n  (hydraulics.py:15-26)."""
    return A * R ** (2 / 3) / n


def dK_dA_(A, n, R, dR_dA):
    """dK/dA = (R^(2/3) + (2/3) A R^(-1/3) dR/dA) / n  (hydraulics.py:28-40)."""
    return (R ** (2 / 3) + A * 2. / 3. * R ** (2 / 3 - 1) * dR_dA) / n


dK_dA = dK_dA_          # the name the rest of this package uses


def _K(A, n, R, K):
    return conveyance(A=A, n=n, R=R) if K is None else K


def normal_flow(bed_slope, area: float = None, roughness: float = None, hydraulic_radius: float = None, K: float = None):
    """Q_n = K sqrt(|S0|), signed with the slope (hydraulics.py:4-13)."""
    Q = _K(area, roughness, hydraulic_radius, K) * np.abs(bed_slope) ** 0.5
    return -Q if bed_slope < 0 else Q


def dQn_dA(S_0, A=None, n=None, R=None, dR_dA=None, dK_dA=None):
    """d(normal flow)/dA = dK/dA sqrt(|S0|), signed with the slope (hydraulics.py:204-215)."""
    slope = dK_dA_(A=A, n=n, R=R, dR_dA=dR_dA) if dK_dA is None else dK_dA
    out = slope * np.abs(S_0) ** 0.5
    return -out if S_0 < 0 else out


def Sf(Q: float, A: float = None, n: float = None, R: float = None, K: float = None) -> float:
    """Manning friction slope Q|Q|/K^2 (hydraulics.py:42-57)."""
    return Q * np.abs(Q) / _K(A, n, R, K) ** 2


def dSf_dA(Q: float, A: float = None, n: float = None, R: float = None, dR_dA: float = None, K: float = None,
           dK_dA: float = None) -> float:
    """dSf/dA = -2 Sf (dK/dA) / K; K and dK/dA are taken as given only when BOTH are (hydraulics.py:59-75)."""
    if K is None or dK_dA is None:
        K, dK_dA = conveyance(A=A, n=n, R=R), dK_dA_(A=A, n=n, R=R, dR_dA=dR_dA)
    return -2 * Sf(Q=Q, K=K) * (dK_dA / K)


def dSf_dQ(Q: float, A: float = None, n: float = None, R: float = None, K: float = None) -> float:
    """dSf/dQ = 2|Q|/K^2 (hydraulics.py:77-92)."""
    return 2 * abs(Q) / _K(A, n, R, K) ** 2


# ---- Froude number ---------------------------------------------------------------------------------------------------
def froude_num(T: float, A: float, Q: float):
    """Froude number with the reference's 1e-6 clamps (hydraulics.py:155-168)."""
    V = Q / max(A, 1e-6)
    D = A / max(T, 1e-6)
    return V / np.sqrt(g * max(D, 1e-6))


def froude_array(T, A, Q):
    V = Q / np.maximum(A, 1e-6)
    D = A / np.maximum(T, 1e-6)
    return V / np.sqrt(g * np.maximum(D, 1e-6))


def dFr_dA(T: float, A: float, Q: float) -> float:
    """d Fr / dA at fixed top width, unclamped: Fr = V (gD)^-1/2, V = Q/A, D = A/T (hydraulics.py:170-187)."""
    gD = g * (A / T)
    return -0.5 * (Q / A) * gD ** (-1.5) * g * (1.0 / T) + (-Q / A ** 2) * gD ** (-0.5)


def dFr_dQ(T: float, A: float):
    """d Fr / dQ = 1 / (A sqrt(g A / T))  (hydraulics.py:189-202)."""
    return (1.0 / A) * (g * (A / T)) ** (-0.5)


# ---- transverse-circulation (curvature) slope ------------------------------------------------------------------------
def darcey_weisbach_f(n: float, R: float):
    """f = 8 g / C^2 with the Chezy coefficient C = R^(1/6) / n  (hydraulics.py:217-229)."""
    C = R ** (1 / 6) / n
    return 8 * g / C ** 2


darcy_weisbach_f = darcey_weisbach_f


def _bend_terms(h, T, A, Q, n, R, rc):
    """numerator and denominator of Sc = (2.86 sqrt f + 2.07 f) h^2 Fr^2 / ((0.565 + sqrt f) rc^2) and their parts"""
    Fr = froude_num(T=T, A=A, Q=Q)
    f = darcey_weisbach_f(n=n, R=R)
    rf = np.sqrt(f)
    shape = 2.86 * rf + 2.07 * f
    return Fr, f, rf, shape, shape * h ** 2 * Fr ** 2, (0.565 + rf) * rc ** 2


def Sc(h: float, T: float, A: float, Q: float, n: float, R: float, rc: float) -> float:
    """Energy gradient of the transverse circulation in a bend of radius rc (hydraulics.py:94-117)."""
    _, _, _, _, num, den = _bend_terms(h, T, A, Q, n, R, rc)
    return num / den


curvature_slope = Sc


def dSc_dA(h, A, Q, n, R, rc, dR_dA, T):
    """dSc/dA by the quotient rule, with dh/dA = 1/T, f = 8 g n^2 R^(-1/3) and the unclamped dFr/dA (hydraulics.py:119-137)."""
    Fr, f, rf, shape, num, den = _bend_terms(h, T, A, Q, n, R, rc)
    df = -(8.0 / 3.0) * g * n ** 2 * R ** (-4.0 / 3.0) * dR_dA
    dshape = 2.86 / (2 * rf) * df + 2.07 * df
    dnum = dshape * h ** 2 * Fr ** 2 + shape * (2 * h * (1. / T) * Fr ** 2 + h ** 2 * 2 * Fr * dFr_dA(A=A, Q=Q, T=T))
    dden = (1.0 / (2 * rf) * df) * rc ** 2
    return (dnum * den - num * dden) / (den ** 2)


def dSc_dQ(h, T, A, Q, n, R, rc):
    """dSc/dQ: only Fr depends on Q (hydraulics.py:139-153)."""
    Fr, _, _, shape, num, den = _bend_terms(h, T, A, Q, n, R, rc)
    dnum = shape * h ** 2 * 2 * Fr * dFr_dQ(T=T, A=A)
    return (dnum * den - num * 0.0) / (den ** 2)
