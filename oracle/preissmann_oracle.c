/* preissmann_oracle.c - plain-C restatement of the reference's Preissmann Newton step.
 *
 * TEST INFRASTRUCTURE ONLY (checker + CPU baseline).  Nothing in the product path
 * (flow-sim_amd/) includes, links or calls this file.  Same algorithm as
 * oracle/preissmann_oracle.py, scalar loops, fp64, no FMA contraction (-ffp-contract=off):
 *   section / hydraulics  <- src/hydromodel/cross_section.py:623-793, hydraulics.py:4-229
 *   boundary rows         <- src/hydromodel/boundary.py:56-242, rating_curve.py:32-63,:132-147,
 *                            lumped_storage.py:24-45
 *   residual + Jacobian   <- src/hydromodel/preissmann.py:61-81,:220-344,:407-733,:899-910
 *   Newton / time loop    <- preissmann.py:101-177 (stored iterate = pre-update one)
 * The reference hands the 2N x 2N system (kl = ku = 2) to scipy's SuperLU; here it is solved by a
 * banded LU with partial pivoting (LAPACK dgbsv layout, own code).  Pinned by
 * tests/test_oracle_c.py against the golden vectors generated from the reference.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define G 9.80665

enum { BC_FLOW = 0, BC_STAGE = 1, BC_FIXED = 2, BC_NORMAL = 3, BC_POWER = 4, BC_POLY = 5, BC_BLEND = 6, BC_STORAGE = 7 };

typedef struct {
  int kind;
  double p[10];            /* same parameter order as include/flowsim_abi.h FS_BC_* */
  const double *target;    /* [nt] or NULL */
} fso_bc;

typedef struct {
  int N, nt, max_iter;
  double theta, dt, dx, tol;
  /* geometry rows, each [N] */
  const double *z_bed, *b_main, *m_main, *n_main, *n_left, *n_right, *is_compound, *h_bf, *b_fp_l, *b_fp_r, *m_fp, *curv;
  const double *h0, *Q0;
  fso_bc us, ds;
} fso_problem;

typedef struct { double A, P, R, T, K, neq, dRdA, dKdA; int over; } props_t;

static props_t props(const fso_problem *p, int i, double h) {
  props_t g;
  double b = p->b_main[i], m = p->m_main[i], d = h > 0 ? h : 0;
  int comp = p->is_compound[i] > 0.5;
  double sm = sqrt(1.0 + m * m);
  double T = b + 2.0 * m * d, A = (b + T) / 2.0 * d, P = b + 2.0 * d * sm, dPdh = 2.0 * sm, K;
  g.over = comp && d > p->h_bf[i];
  if (g.over) {
    double hb = p->h_bf[i], mf = p->m_fp[i], bl = p->b_fp_l[i], br = p->b_fp_r[i];
    double dfp = d - hb, Tb = b + 2.0 * m * hb, sf = sqrt(1.0 + mf * mf);
    double A_main = (b + Tb) / 2.0 * hb, P_main = b + 2.0 * hb * sm;
    double A_l = (bl + 0.5 * mf * dfp) * dfp, P_l = bl + dfp * sf;
    double A_r = (br + 0.5 * mf * dfp) * dfp, P_r = br + dfp * sf;
    double A_m = A_main + Tb * dfp;
    double R_m = P_main > 0 ? A_m / P_main : 0, R_l = P_l > 0 ? A_l / P_l : 0, R_r = P_r > 0 ? A_r / P_r : 0;
    double Kl = A_l * pow(R_l, 2.0 / 3.0) / p->n_left[i], Km = A_m * pow(R_m, 2.0 / 3.0) / p->n_main[i];
    double Kr = A_r * pow(R_r, 2.0 / 3.0) / p->n_right[i];
    A = A_main + A_l + A_r; P = P_main + P_l + P_r;
    T = (bl + Tb + br) + 2.0 * mf * dfp; dPdh = 2.0 * sf;
    K = pow(pow(Kl, 1.5) + pow(Km, 1.5) + pow(Kr, 1.5), 2.0 / 3.0);
    g.R = P > 0 ? A / P : 0;
  } else {
    g.R = P > 0 ? A / P : 0;
    K = A * pow(g.R, 2.0 / 3.0) / p->n_main[i];
    if (comp) K = pow(pow(0.0, 1.5) + pow(K, 1.5) + pow(0.0, 1.5), 2.0 / 3.0);
  }
  {
    double R23 = pow(g.R, 2.0 / 3.0), neq = p->n_main[i];
    if (comp && A > 0 && g.R > 0 && K > 0) neq = A * R23 / K;
    g.dRdA = (P <= 0 || T <= 0) ? 0.0 : (P - A * (dPdh * (1.0 / T))) / (P * P);
    g.dKdA = A <= 0 ? 0.0 : (R23 + A * 2.0 / 3.0 * pow(g.R, 2.0 / 3.0 - 1) * g.dRdA) / neq;
    g.neq = neq;
  }
  g.A = A; g.P = P; g.T = T; g.K = K;
  return g;
}

typedef struct { double A, T, Se, dSeA, dSeQ; } node_t;

static node_t node_terms(const fso_problem *p, int i, double h, double Q) {
  props_t g = props(p, i, h);
  node_t t;
  double Sf = Q * fabs(Q) / (g.K * g.K);
  t.A = g.A; t.T = g.T;
  t.Se = Sf; t.dSeA = -2.0 * Sf * (g.dKdA / g.K); t.dSeQ = 2.0 * fabs(Q) / (g.K * g.K);
  if (p->curv[i] != 0) {
    double rc = 1.0 / p->curv[i];
    double V = Q / fmax(g.A, 1e-6), D = g.A / fmax(g.T, 1e-6), Fr = V / sqrt(G * fmax(D, 1e-6));
    double C = pow(g.R, 1.0 / 6.0) / g.neq, f = 8 * G / (C * C), sq = sqrt(f);
    double num = (2.86 * sq + 2.07 * f) * h * h * Fr * Fr, den = (0.565 + sq) * rc * rc;
    t.Se = Sf + num / den;
    if (fabs(p->curv[i]) > 1e-12) {
      double gD = G * (g.A / g.T);
      double dFrA = -0.5 * (Q / g.A) * pow(gD, -1.5) * G * (1.0 / g.T) + (-Q / (g.A * g.A)) * pow(gD, -0.5);
      double dfA = -(8.0 / 3.0) * G * g.neq * g.neq * pow(g.R, -4.0 / 3.0) * g.dRdA;
      double dnum = (2.86 / (2 * sq) * dfA + 2.07 * dfA) * h * h * Fr * Fr +
                    (2.86 * sq + 2.07 * f) * (2 * h * (1.0 / g.T) * Fr * Fr + h * h * 2 * Fr * dFrA);
      double dden = (1.0 / (2 * sq) * dfA) * rc * rc;
      double dFrQ = (1.0 / g.A) * pow(gD, -0.5);
      double dnumq = (2.86 * sq + 2.07 * f) * h * h * 2 * Fr * dFrQ;
      t.dSeA += (dnum * den - num * dden) / (den * den) * g.T;
      t.dSeQ += (dnumq * den) / (den * den);
    }
  }
  return t;
}

static double blend_q(const double *p, double z) {
  double al, lo, hi;
  if (z >= p[0] + p[1]) al = 1.0; else if (z <= p[0]) al = 0.0;
  else { double s = (z - p[0]) / p[1]; al = 3 * s * s - 2 * s * s * s; }
  lo = p[2] + p[3] * z + p[4] * z * z; hi = p[5] + p[6] * z + p[7] * z * z;
  return (1.0 - al) * lo + al * hi;
}

/* residual and derivatives of one boundary row; returns nonzero on the storage range error */
static int bc_eval(const fso_problem *p, const fso_bc *bc, int node, int k, double h, double Q, double Qold,
                   double Yprev, double *Ynew, double *res, double *dh, double *dq) {
  const double *q = bc->p;
  switch (bc->kind) {
    case BC_FLOW: *res = Q - bc->target[k]; *dh = 0; *dq = 1; break;
    case BC_STAGE: *res = h - (bc->target[k] - q[0]); *dh = 1; *dq = 0; break;
    case BC_FIXED: *res = h - q[0]; *dh = 1; *dq = 0; break;
    case BC_NORMAL: {
      double S0 = q[0], sg = S0 < 0 ? -1.0 : 1.0, rt = pow(fabs(S0), 0.5);
      props_t gr = props(p, node, h), gd = props(p, node, h + q[1] - p->z_bed[node]);
      *res = Q - sg * gr.K * rt; *dh = 0 - sg * gd.dKdA * rt * gd.T; *dq = 1;
    } break;
    case BC_POWER: { double x = q[3] + h + q[2]; *res = Q - q[0] * pow(x, q[1]); *dh = 0 - q[0] * q[1] * pow(x, q[1] - 1); *dq = 1; } break;
    case BC_POLY: { double x = q[4] + h + q[3]; *res = Q - (q[0] * x * x + q[1] * x + q[2]); *dh = 0 - (q[0] * 2 * x + q[1]); *dq = 1; } break;
    case BC_BLEND: {
      double z = q[9] + h, dY = q[8];
      *res = Q - blend_q(q, z); *dh = 0 - (blend_q(q, z + dY) - blend_q(q, z - dY)) / (2 * dY); *dq = 1;
    } break;
    case BC_STORAGE: {
      double vol = 0.5 * (Qold + Q) * p->dt, Yold = k == 1 ? h + q[4] : Yprev, Y = Yold + vol / q[0];
      if (!(Y >= q[2] && Y <= q[3])) return 1;
      if (Y < q[1]) Y = q[1];
      *Ynew = Y;
      *res = h - (Y - q[4]); *dh = 1; *dq = 0 - (Y <= q[1] ? 0.0 : 1.0 / q[0]) * 0.5 * p->dt;
    } break;
    default: return 2;
  }
  return 0;
}

/* banded LU with partial pivoting, kl = ku = 2; ab[(2*kl+ku+1) x n] column-major like dgbsv */
static int band_solve(int n, double *ab, double *b, int *piv) {
  const int kl = 2, ku = 2, ld = 2 * kl + ku + 1, kv = kl + ku;
#define AB(i, j) ab[(size_t)(j) * ld + (kv + (i) - (j))]
  for (int j = 0; j < n; ++j) {
    int km = (kl < n - 1 - j) ? kl : n - 1 - j, jp = 0;
    double mx = fabs(AB(j, j));
    for (int i = 1; i <= km; ++i) if (fabs(AB(j + i, j)) > mx) { mx = fabs(AB(j + i, j)); jp = i; }
    piv[j] = j + jp;
    if (mx == 0.0) return 1;
    int ju = j + ku + jp; if (ju > n - 1) ju = n - 1; if (j + kv < ju) ju = j + kv;
    if (jp) for (int c = j; c <= ju; ++c) { double t = AB(j, c); AB(j, c) = AB(j + jp, c); AB(j + jp, c) = t; }
    for (int i = 1; i <= km; ++i) AB(j + i, j) /= AB(j, j);
    for (int c = j + 1; c <= ju; ++c) { double t = AB(j, c); if (t != 0) for (int i = 1; i <= km; ++i) AB(j + i, c) -= AB(j + i, j) * t; }
  }
  for (int j = 0; j < n; ++j) {
    int km = (kl < n - 1 - j) ? kl : n - 1 - j;
    if (piv[j] != j) { double t = b[j]; b[j] = b[piv[j]]; b[piv[j]] = t; }
    for (int i = 1; i <= km; ++i) b[j + i] -= AB(j + i, j) * b[j];
  }
  for (int j = n - 1; j >= 0; --j) {
    b[j] /= AB(j, j);
    int lo = j - kv > 0 ? j - kv : 0;
    for (int i = lo; i < j; ++i) b[i] -= AB(i, j) * b[j];
  }
#undef AB
  return 0;
}

/* depth, flow: [nt][N] out; iters: [nt] out.  returns 0 ok, 1 max_iter, 2 nan, 3 storage range, 4 singular */
int fso_run(const fso_problem *p, double *depth, double *flow, int *iters, double *storage_stage) {
  const int N = p->N, n2 = 2 * N, ld = 7;
  const double th = p->theta, dt = p->dt, dx = p->dx, cq = th / dx, hth = 0.5 * th;
  double *x = malloc(sizeof(double) * n2), *R = malloc(sizeof(double) * n2), *ab = malloc(sizeof(double) * ld * n2);
  node_t *nw = malloc(sizeof(node_t) * N), *od = malloc(sizeof(node_t) * N);
  int *piv = malloc(sizeof(int) * n2), status = 0;
  double Yprev = 0, Ynew = 0;
  for (int i = 0; i < N; ++i) { depth[i] = p->h0[i]; flow[i] = p->Q0[i]; x[2 * i] = p->h0[i]; x[2 * i + 1] = p->Q0[i]; }
  iters[0] = 0;
  for (int i = 0; i < N; ++i) od[i] = node_terms(p, i, p->h0[i], p->Q0[i]);
  for (int k = 1; k < p->nt && !status; ++k) {
    double *hk = depth + (size_t)k * N, *Qk = flow + (size_t)k * N;
    const double *ho = depth + (size_t)(k - 1) * N, *Qo = flow + (size_t)(k - 1) * N;
    int it = 0;
    for (;;) {
      ++it;
      if (it - 1 >= p->max_iter) { status = 1; --it; break; }
      for (int i = 0; i < N; ++i) { hk[i] = x[2 * i]; Qk[i] = x[2 * i + 1]; nw[i] = node_terms(p, i, hk[i], Qk[i]); }
      memset(ab, 0, sizeof(double) * ld * n2);
#define J(r, c) ab[(size_t)(c) * ld + (4 + (r) - (c))]
      double rU, uh, uq, rD, dh, dq, dummy;
      if (bc_eval(p, &p->us, 0, k, hk[0], Qk[0], 0, 0, &dummy, &rU, &uh, &uq)) { status = 3; break; }
      if (bc_eval(p, &p->ds, N - 1, k, hk[N - 1], Qk[N - 1], Qo[N - 1], Yprev, &Ynew, &rD, &dh, &dq)) { status = 3; break; }
      R[0] = rU; R[n2 - 1] = rD;
      J(0, 0) = uh; J(0, 1) = uq; J(n2 - 1, n2 - 2) = dh; J(n2 - 1, n2 - 1) = dq;
      for (int i = 0; i < N - 1; ++i) {
        const node_t *a = &nw[i], *b = &nw[i + 1], *ao = &od[i], *bo = &od[i + 1];
        double dAdt = (b->A + a->A - bo->A - ao->A) / (2 * dt);
        double dQdx = th * ((Qk[i + 1] - Qk[i]) / dx) + (1 - th) * ((Qo[i + 1] - Qo[i]) / dx);
        double dQdt = (Qk[i + 1] + Qk[i] - Qo[i + 1] - Qo[i]) / (2 * dt);
        double d2 = th * ((Qk[i + 1] * Qk[i + 1] / b->A - Qk[i] * Qk[i] / a->A) / dx) +
                    (1 - th) * ((Qo[i + 1] * Qo[i + 1] / bo->A - Qo[i] * Qo[i] / ao->A) / dx);
        double avgA = 0.5 * th * (b->A + a->A) + 0.5 * (1 - th) * (bo->A + ao->A);
        double dYdx = th * (((p->z_bed[i + 1] + hk[i + 1]) - (p->z_bed[i] + hk[i])) / dx) +
                      (1 - th) * (((p->z_bed[i + 1] + ho[i + 1]) - (p->z_bed[i] + ho[i])) / dx);
        double avgSe = 0.5 * th * (b->Se + a->Se) + 0.5 * (1 - th) * (bo->Se + ao->Se);
        int rc = 1 + 2 * i, rm = 2 + 2 * i, c0 = 2 * i;
        R[rc] = dAdt + dQdx;
        R[rm] = dQdt + d2 + G * avgA * (dYdx + avgSe);
        J(rc, c0) = a->T / (2 * dt); J(rc, c0 + 1) = -cq; J(rc, c0 + 2) = b->T / (2 * dt); J(rc, c0 + 3) = cq;
        {
          double va = Qk[i] / a->A, vb = Qk[i + 1] / b->A;
          J(rm, c0) = cq * va * va * a->T + G * (avgA * (-cq + hth * a->dSeA * a->T) + hth * a->T * (dYdx + avgSe));
          J(rm, c0 + 1) = 1 / (2 * dt) - cq * 2 * va + G * (avgA * (hth * a->dSeQ));
          J(rm, c0 + 2) = -cq * vb * vb * b->T + G * (avgA * (cq + hth * b->dSeA * b->T) + hth * b->T * (dYdx + avgSe));
          J(rm, c0 + 3) = 1 / (2 * dt) + cq * 2 * vb + G * (avgA * (hth * b->dSeQ));
        }
      }
#undef J
      double err = 0;
      for (int i = 0; i < n2; ++i) { err += R[i] * R[i]; R[i] = -R[i]; }
      err = sqrt(err);
      if (band_solve(n2, ab, R, piv)) { status = 4; break; }
      for (int i = 0; i < n2; ++i) x[i] += R[i];
      if (!(err == err) || isinf(err)) { status = 2; break; }
      if (err < p->tol) break;
    }
    iters[k] = it;
    if (!status) {
      memcpy(od, nw, sizeof(node_t) * N);
      if (p->ds.kind == BC_STORAGE) { Yprev = Ynew; if (storage_stage) storage_stage[k] = Ynew; }
    }
  }
  free(x); free(R); free(ab); free(nw); free(od); free(piv);
  return status;
}
