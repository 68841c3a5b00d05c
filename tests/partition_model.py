"""Numpy model of the on-chip linear solve used by the HIP kernel (flow-sim_amd/csrc/fs_device.hpp, fs_kernel.hpp).

Not the product and not the oracle: a lane-by-lane executable description of the device algorithm, kept in tests/ so
that its numerics (no pivoting, fixed elimination order) can be compared on the CPU with what the reference calls,
scipy.sparse.linalg.spsolve (preissmann.py:146), on the Jacobians of the golden cases.

System (preissmann.py:874-897): unknowns d_i = (dh_i, dQ_i), i = 0..N-1;
  U row      : u . d_0                          = ru
  cell i     : C row  a_i . d_i + b_i . d_{i+1} = rc_i      a_i = (T_i/(2dt), -theta/dx), b_i = (T_{i+1}/(2dt), +theta/dx)
               M row  c_i . d_i + e_i . d_{i+1} = rm_i      i = 0..N-2
  D row      : w . d_{N-1}                      = rd

Characteristic-like unknowns.  The continuity row has the same two numbers on both nodes (t_i = T_i/(2dt) on dh,
-+cq = theta/dx on dQ), so with
      p_i = t_i dh_i + cq dQ_i ,   m_i = t_i dh_i - cq dQ_i
it reads  m_i + p_{i+1} = rc_i : every p but the first is an m and a number, no division.  What is left is ONE scalar
tridiagonal system in (p_0, m_0, ..., m_{N-1}): row i is the momentum row of cell i,
      alpha_i p_i + (beta_i - gamma_i) m_i + delta_i m_{i+1} = rm_i - gamma_i rc_i ,   p_i = rc_{i-1} - m_{i-1} (i > 0)
      alpha, beta = c0/(2t_i) +- c1/(2cq) ,  gamma, delta = e0/(2t_{i+1}) +- e1/(2cq)
followed by the D row (on p_{N-1}, m_{N-1}) and identity rows up to the lane grid; the U row closes the system on the
left.  Frictionless and subcritical the rows are diagonally dominant, |beta - gamma| = 2(a + k) against |a - k + v| +
|a - k - v| with a = theta dt (c^2 - v^2)/dx, k = dx/(4 theta dt): no pivoting, and none of the singular pivot blocks
the 2x2 block order of the classical double sweep runs into on steep shallow reaches.

Algorithm: `T` lanes, lane t owns the rows [t*m, (t+1)*m), m >= 2.
  1. local fold, top to bottom: the running "down" row  d1 p_a + d2 m_j + d3 m_{j+1} = rd  (forward elimination with
     the fill-in column of the lane's first unknown) and the running "up" row  u1 p_a + u2 m_a + u3 m_j = ru  (the
     lane's first row with its last unknown substituted as the sweep moves on); per row three numbers (R1, R2, R3) for
     the back-substitution  m_{j-1} = R3 - R1 p_a - R2 m_j.
  2. tree: two adjacent segments (up, down, rc of the last row) merge by eliminating the left one's last unknown
     from {left.down, right.up} - one 2x2 determinant; only m_{b-1} has to be substituted, m_b drops out; log2 T levels.
     The root and the U row give (p_0, m_0, m_last).  On the way down every lane carries two numbers, the p of its
     group's first row and the m of its last one; a level's record is (A1, A2, A3, rc_left).
  3. local back-substitution, then dh = (p + m)/(2t), dQ = (p - m)/(2cq).
"""
import numpy as np


def cells_from_reference_layout(data, R):
    """Split the reference's 8N-4 Jacobian entries / 2N residuals into rows of the form above."""
    N = len(R) // 2
    blk = np.asarray(data[2:-2]).reshape(N - 1, 8)
    a, b = blk[:, 0:2], blk[:, 2:4]
    c, e = blk[:, 4:6], blk[:, 6:8]
    rc, rm = -np.asarray(R[1:-1:2]), -np.asarray(R[2:-1:2])
    return (np.array(data[0:2]), -R[0]), (a, b, rc, c, e, rm), (np.array(data[-2:]), -R[-1])


def scalar_rows(data, R, dtype=np.float64):
    """(alpha, D, delta, rho0, rc)[rows] of the tridiagonal system + what the closure and the conversion back need.
    rho0 = rm - gamma rc; the fold subtracts alpha * rc of the row above (rows that do not start a lane)."""
    N = len(R) // 2
    (u, ru), (a, b, rc, c, e, rm), (w, rd) = cells_from_reference_layout(data, R)
    cq = b[0, 1]
    assert np.allclose(a[:, 1], -cq) and np.allclose(b[:, 1], cq)
    t = np.empty(N)                       # t_i = T_i / (2 dt): the dh coefficient of the continuity rows at node i
    t[:-1] = a[:, 0]; t[-1] = b[-1, 0]
    assert np.allclose(b[:, 0], t[1:])
    i2t, i2c = 0.5 / t, 0.5 / cq
    X0, Y0 = c[:, 0] * i2t[:-1], c[:, 1] * i2c
    X1, Y1 = e[:, 0] * i2t[1:], e[:, 1] * i2c
    alpha, beta, gamma, delta = X0 + Y0, X0 - Y0, X1 + Y1, X1 - Y1
    rows = dict(alpha=np.append(alpha, w[0] * i2t[-1] + w[1] * i2c),
                D=np.append(beta - gamma, w[0] * i2t[-1] - w[1] * i2c),
                delta=np.append(delta, 0.0),
                rho0=np.append(rm - gamma * rc, rd),
                rc=np.append(rc, 0.0))
    rows = {k: v.astype(dtype) for k, v in rows.items()}
    ub = (dtype(u[0] * i2t[0] + u[1] * i2c), dtype(u[0] * i2t[0] - u[1] * i2c), dtype(ru))      # aU p_0 + bU m_0 = ru
    return rows, ub, i2t.astype(dtype), dtype(i2c)


def merge(X, Y):
    """Segments over lanes (arrays).  X left, Y right; returns the merged segment and the level's record."""
    ru2 = Y["ru"] - Y["u1"] * X["rc"]                 # right rows on m_{b-1}: p_b = rc_left - m_{b-1}
    rd2 = Y["rd"] - Y["d1"] * X["rc"]
    det = X["d2"] * Y["u2"] + X["d3"] * Y["u1"]
    r = 1.0 / det
    g1, g2 = Y["u2"] * r, X["d3"] * r
    A1, A2, A3 = g1 * X["d1"], g2 * Y["u3"], g1 * X["rd"] - g2 * ru2      # m_{b-1} = -A1 p_a + A2 m_{c-1} + A3
    Z = dict(u1=X["u1"] - X["u3"] * A1, u2=X["u2"], u3=X["u3"] * A2, ru=X["ru"] - X["u3"] * A3,
             d1=Y["d1"] * A1, d2=Y["d2"] - Y["d1"] * A2, d3=Y["d3"], rd=rd2 + Y["d1"] * A3, rc=Y["rc"])
    return Z, dict(A1=A1, A2=A2, A3=A3, rc=X["rc"], det=det, scale=np.abs(X["d2"] * Y["u2"]) + np.abs(X["d3"] * Y["u1"]))


def continuant_close(seg, aU, bU, rU, dtype):
    """The cross-wave step of the multi-wave kernels compiled without the conditioning monitor (fs_kernel.hpp, FS_XWAVE_CONT): the W
    remaining segments and the upstream row as ONE tridiagonal system in y = (p_0, x_0 .. x_{W-1}), x_w = m of segment w's last row,
    solved by forward and backward continuants with one reciprocal.  Returns (pL[W], mR[W]): the p of every segment's first row
    and the m of its last one - what the way down starts from."""
    W = len(seg["u1"])
    n = W
    g = lambda k, w: seg[k][w]
    A, B, C, Rr, AC = (np.zeros(n + 1, dtype) for _ in range(5))
    B[0] = aU - bU * g("u1", 0); C[0] = -(bU * g("u3", 0)); Rr[0] = rU - bU * g("ru", 0)
    for k in range(W):
        i = k + 1
        A[i] = g("d1", 0) if k == 0 else -g("d1", k)
        rdk = g("rd", 0) if k == 0 else g("rd", k) - g("d1", k) * g("rc", k - 1)
        if k + 1 < W:
            B[i] = g("d2", k) + g("d3", k) * g("u1", k + 1); C[i] = -(g("d3", k) * g("u3", k + 1))
            Rr[i] = rdk - g("d3", k) * (g("ru", k + 1) - g("u1", k + 1) * g("rc", k))
        else:
            B[i] = g("d2", k); Rr[i] = rdk
        AC[i] = A[i] * C[i - 1]
    th, rho, ph, sg = np.zeros(n + 1, dtype), np.zeros(n + 1, dtype), np.zeros(n + 2, dtype), np.zeros(n + 2, dtype)
    th[0] = B[0]; rho[0] = Rr[0]
    th[1] = B[1] * th[0] - AC[1]; rho[1] = th[0] * Rr[1] - A[1] * rho[0]
    for i in range(2, n + 1):
        th[i] = B[i] * th[i - 1] - AC[i] * th[i - 2]; rho[i] = th[i - 1] * Rr[i] - A[i] * rho[i - 1]
    ph[n + 1] = 1; ph[n] = B[n]; sg[n] = Rr[n]
    ph[n - 1] = B[n - 1] * ph[n] - AC[n]; sg[n - 1] = ph[n] * Rr[n - 1] - C[n - 1] * sg[n]
    for i in range(n - 2, 0, -1):
        ph[i] = B[i] * ph[i + 1] - AC[i + 1] * ph[i + 2]; sg[i] = ph[i + 1] * Rr[i] - C[i] * sg[i + 1]
    rdet = dtype(1) / th[n]
    y = np.zeros(n + 1, dtype)
    y[0] = (ph[1] * rho[0] - C[0] * sg[1]) * rdet
    for i in range(1, n):
        y[i] = (ph[i + 1] * rho[i] - C[i] * th[i - 1] * sg[i + 1]) * rdet
    y[n] = rho[n] * rdet
    pL = np.array([y[0]] + [g("rc", w - 1) - y[w] for w in range(1, W)], dtype)
    return pL, y[1:].astype(dtype)


def solve(data, R, m, T=None, dtype=np.float64, info=None, xwave=0):
    """Returns (delta[2N] with J delta = -R computed with the device's elimination order, smallest pivot relative to
    its terms).  xwave = W > 1: the tree stops at W segments (the waves of a reach) and continuant_close takes over."""
    N = len(R) // 2
    assert m >= 2
    rows, (aU, bU, rU), i2t, i2c = scalar_rows(data, R, dtype)
    nr = N                                 # N - 1 cells + the D row
    if T is None:
        T = 1
        while T * m < nr:
            T *= 2
    assert T * m >= nr
    one, zero = dtype(1), dtype(0)

    def pad(x, fill):
        out = np.full(T * m, fill, dtype=dtype); out[:nr] = x; return out.reshape(T, m)
    al, D, de, rho0, rc = (pad(rows["alpha"], zero), pad(rows["D"], one), pad(rows["delta"], zero),
                           pad(rows["rho0"], zero), pad(rows["rc"], zero))
    # 1. local fold
    d1, d2, d3, rd = al[:, 0].copy(), D[:, 0].copy(), de[:, 0].copy(), rho0[:, 0].copy()
    u1, u2, u3, ru = np.zeros(T, dtype), np.ones(T, dtype), -np.ones(T, dtype), np.zeros(T, dtype)
    rec = []
    worst = np.inf
    cu = np.ones(T, dtype)         # largest |d ru / d rhs_j| over the lane's rows (sensitivity of the up row's right-hand side)
    cd = np.ones(T, dtype)         # the same for the down row
    for j in range(1, m):
        r = one / d2
        cu = np.maximum(cu, np.abs(u3 * r) * cd)
        cd = np.maximum(1.0, np.abs(al[:, j] * r) * cd)
        R1, R2, R3 = d1 * r, d3 * r, rd * r                   # m_{j-1} = R3 - R1 p_a - R2 m_j
        rec.append((R1, R2, R3))
        rho = rho0[:, j] - al[:, j] * rc[:, j - 1]
        newd2 = D[:, j] + al[:, j] * R2
        worst = min(worst, np.min(np.abs(newd2) / (np.abs(D[:, j]) + np.abs(al[:, j] * R2))))
        d1, d2, d3, rd = al[:, j] * R1, newd2, de[:, j], rho + al[:, j] * R3
        u1, u3, ru = u1 - u3 * R1, -u3 * R2, ru - u3 * R3
    seg = dict(u1=u1, u2=u2, u3=u3, ru=ru, d1=d1, d2=d2, d3=d3, rd=rd, rc=rc[:, m - 1])
    sens = [[cu.copy(), cd.copy()], [np.ones(T, dtype), np.ones(T, dtype)]]      # variant 0: exact in-lane start; variant 1: lanes start at 1
    growth = np.abs(u3)                                      # the kernel's conditioning monitor: largest |u3| of any segment
    aux = dict(d1d2=float(np.max(np.abs(d1 / d2))), A1=0.0, A2=0.0, R1=max(float(np.max(np.abs(r[0]))) for r in rec) if rec else 0.0,
               R2=max(float(np.max(np.abs(r[1]))) for r in rec) if rec else 0.0, u1=float(np.max(np.abs(u1))))
    # 2. tree
    levels = []
    while len(seg["u1"]) > max(1, xwave):
        X = {k: v[0::2] for k, v in seg.items()}
        Y = {k: v[1::2] for k, v in seg.items()}
        seg, el = merge(X, Y)
        levels.append(el)
        rr = 1.0 / el["det"]
        for v in range(2):
            cuX, cuY, cdX, cdY = sens[v][0][0::2], sens[v][0][1::2], sens[v][1][0::2], sens[v][1][1::2]
            g2a = np.abs(X["d3"] * rr)
            sens[v] = [np.maximum(np.maximum(cuX, np.abs(X["u3"] * rr) * cdX), np.abs(X["u3"]) * g2a * cuY),
                       np.maximum(np.maximum(cdY, np.abs(Y["d1"] * rr) * cdX), np.abs(Y["d1"]) * g2a * cuY)]
        growth = np.maximum(np.maximum(growth[0::2], growth[1::2]), np.abs(seg["u3"]))
        aux["A1"] = max(aux["A1"], float(np.max(np.abs(el["A1"])))); aux["A2"] = max(aux["A2"], float(np.max(np.abs(el["A2"]))))
        aux["d1d2"] = max(aux["d1d2"], float(np.max(np.abs(seg["d1"] / seg["d2"])))); aux["u1"] = max(aux["u1"], float(np.max(np.abs(seg["u1"]))))
        worst = min(worst, np.min(np.abs(el["det"]) / el["scale"]))
    if xwave > 1 and len(seg["u1"]) == xwave:
        assert np.all(seg["u2"] == 1)
        pL, mR = continuant_close(seg, aU, bU, rU, dtype)
        return _way_down(levels, pL, mR, rec, rc, i2t, i2c, N, T, m, dtype), worst
    S = {k: v[0] for k, v in seg.items()}
    # root: up  u1 p_0 + u2 m_0 + u3 m_last = ru ; down  d1 p_0 + d2 m_last = rd (nothing right of the last row) ; U row
    r = one / S["d2"]
    e1, e3 = S["u1"] - S["u3"] * S["d1"] * r, S["ru"] - S["u3"] * S["rd"] * r        # e1 p_0 + u2 m_0 = e3
    det = aU * S["u2"] - bU * e1
    if info is not None:     # what the kernel's conditioning monitor sees (fs_device.hpp: close_root)
        sp = [abs(1.0 / det) * max(1.0, abs(bU) * max(c[0][0], abs(S["u3"] * r) * c[1][0])) for c in sens]       # of p_0
        sl = [abs(r) * max(c[1][0], abs(S["d1"]) * q) for c, q in zip(sens, sp)]                                   # of m_last
        info.update(growth=float(growth[0]), sens=[float(max(a_, b_)) for a_, b_ in zip(sp, sl)], aux=aux, root=dict(S), aU=aU, bU=bU, e1=e1, det=det, f=S["u3"] * r)
    worst = min(worst, abs(det) / (abs(aU * S["u2"]) + abs(bU * e1)))
    p0 = (rU * S["u2"] - bU * e3) / det
    mlast = (S["rd"] - S["d1"] * p0) * r
    pL, mR = np.array([p0], dtype), np.array([mlast], dtype)
    return _way_down(levels, pL, mR, rec, rc, i2t, i2c, N, T, m, dtype), worst


def _way_down(levels, pL, mR, rec, rc, i2t, i2c, N, T, m, dtype):
    for el in reversed(levels):
        sep = -el["A1"] * pL + el["A2"] * mR + el["A3"]              # m of the left half's last row
        npL = np.empty(2 * len(pL), dtype); nmR = np.empty_like(npL)
        npL[0::2] = pL; npL[1::2] = el["rc"] - sep
        nmR[0::2] = sep; nmR[1::2] = mR
        pL, mR = npL, nmR
    # 3. local back-substitution
    mm = np.empty((T, m), dtype); pp = np.empty((T, m), dtype)
    mm[:, m - 1] = mR; pp[:, 0] = pL
    for j in range(m - 1, 0, -1):
        R1, R2, R3 = rec[j - 1]
        mm[:, j - 1] = R3 - R1 * pL - R2 * mm[:, j]
        pp[:, j] = rc[:, j - 1] - mm[:, j - 1]
    p, mv = pp.reshape(-1)[:N], mm.reshape(-1)[:N]
    out = np.empty(2 * N, dtype)
    out[0::2] = (p + mv) * i2t
    out[1::2] = (p - mv) * i2c
    return out
