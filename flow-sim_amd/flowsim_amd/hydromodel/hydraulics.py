"""Scalar hydraulic formulas used on the host for initial conditions, boundary set-up and
post-processing (reference: src/hydromodel/hydraulics.py:4-229).  The per-node evaluation inside
the Newton loop lives in csrc/fs_device.hpp."""
import numpy as np

g = 9.80665   # scipy.constants.g, hydraulics.py:2


def conveyance(A, n, R):
    """K = A R^(2/3) / n  (hydraulics.py:15-26)."""
    return A * R ** (2 / 3) / n


def normal_flow(bed_slope, K):
    """Q_n = K sqrt(|S0|), signed with the slope (hydraulics.py:4-13)."""
    Q = K * np.abs(bed_slope) ** 0.5
    return -Q if bed_slope < 0 else Q


def Sf(Q, K):
    """Manning friction slope (hydraulics.py:42-57)."""
    return Q * np.abs(Q) / K ** 2


def dK_dA(A, n, R, dR_dA):
    """hydraulics.py:28-40"""
    return (R ** (2 / 3) + A * 2. / 3. * R ** (2 / 3 - 1) * dR_dA) / n


def dSf_dA(Q, K, dK_dA):
    """hydraulics.py:59-75"""
    return -2 * Sf(Q=Q, K=K) * (dK_dA / K)


def dSf_dQ(Q, K):
    """hydraulics.py:77-92"""
    return 2 * abs(Q) / K ** 2


def froude_num(T, A, Q):
    """Froude number with the reference's 1e-6 clamps (hydraulics.py:155-168)."""
    V = Q / max(A, 1e-6)
    D = A / max(T, 1e-6)
    return V / np.sqrt(g * max(D, 1e-6))


def froude_array(T, A, Q):
    V = Q / np.maximum(A, 1e-6)
    D = A / np.maximum(T, 1e-6)
    return V / np.sqrt(g * np.maximum(D, 1e-6))


def darcy_weisbach_f(n, R):
    """f = 8 g n^2 / R^(1/3)  (hydraulics.py:217-229)."""
    C = R ** (1 / 6) / n
    return 8 * g / C ** 2


def curvature_slope(h, T, A, Q, n, R, rc):
    """Energy gradient of the transverse circulation (hydraulics.py:94-117)."""
    Fr = froude_num(T=T, A=A, Q=Q)
    f = darcy_weisbach_f(n, R)
    return (2.86 * np.sqrt(f) + 2.07 * f) * h ** 2 * Fr ** 2 / ((0.565 + np.sqrt(f)) * rc ** 2)
