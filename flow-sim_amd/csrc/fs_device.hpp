// fs_device.hpp - device-side hydraulics and the 2x2-block segment algebra for gfx950.
//
// Everything here is per-lane scalar code: the Preissmann system is a banded recurrence, there is
// no dense contraction to hand to MFMA.  The expensive scalar ops (fp64 divide, x^(2/3)) are
// replaced by v_rcp_f64 / v_log_f32+v_exp_f32 seeds refined with FMAs.
//
// Reference formulas (cve-mohd/flow-sim): src/hydromodel/hydraulics.py:4-229,
// cross_section.py:114-175,:623-793, boundary.py:56-242, rating_curve.py:32-63,:132-147,
// lumped_storage.py:24-45.  Bug-compatibility notes refer to SURVEY.md section 0 (F2, F3).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/flowsim_abi.h"

namespace fs {

constexpr double kG = 9.80665;  // scipy.constants.g (hydraulics.py:2)

// ---------------------------------------------------------------------------------------------
// scalar helpers
// ---------------------------------------------------------------------------------------------
#ifndef FS_RCP_F32_NR
#define FS_RCP_F32_NR 0
#endif
#ifndef FS_RCP_NR
#define FS_RCP_NR 1
#endif
// fused multiply-add in the working precision (__builtin_fma on floats is the double one: two conversions in, one out)
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

__device__ __forceinline__ double frcp(double x) {
  // v_rcp_f64 seed (measured on gfx950: 2^-24.4 relative, tools/micro/rcp_prec.hip) + Newton steps:
  // one step leaves <= 2.3e-15 relative (~10 ulp), two steps are correctly rounded.  The kernel's
  // reciprocals feed Jacobian entries and friction terms whose effect on the accepted iterate is
  // orders below the 1e-8 parity bar, so one step is the default.
  double r = __builtin_amdgcn_rcp(x);
#pragma unroll
  for (int i = 0; i < FS_RCP_NR; ++i) {
    const double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
  }
  return r;
}
__device__ __forceinline__ float frcp(float x) {
#if FS_RCP_F32_NR
  float r = __builtin_amdgcn_rcpf(x);
  float e = __builtin_fmaf(-x, r, 1.0f);
  return __builtin_fmaf(r, e, r);
#else
  return __builtin_amdgcn_rcpf(x);          // v_rcp_f32 is accurate to 1 ulp: no Newton step
#endif
}

// 1/sqrt(x): v_rsq_f64 seed r (single-precision accurate, e = 1 - x r^2 ~ 1e-7) and ONE third-order correction
// r (1 - e)^(-1/2) = r (1 + e/2 + 3 e^2/8 + O(e^3)): the neglected term is below double rounding, five instructions where two Newton
// steps (r += r/2 (1 - x r^2), FS_RSQ_THIRD=0) take eight
#ifndef FS_RSQ_THIRD
#define FS_RSQ_THIRD 1
#endif
__device__ __forceinline__ double frsq(double x) {
  double r = __builtin_amdgcn_rsq(x);
#if FS_RSQ_THIRD
  const double e = __builtin_fma(-(x * r), r, 1.0);
  const double p = __builtin_fma(e, 0.375, 0.5);
  return __builtin_fma(r * e, p, r);
#else
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const double e = __builtin_fma(-(x * r), r, 1.0);
    r = __builtin_fma(r * 0.5, e, r);
  }
  return r;
#endif
}
__device__ __forceinline__ float frsq(float x) {
  float r = __builtin_amdgcn_rsqf(x);
  const float e = __builtin_fmaf(-(x * r), r, 1.0f);
  return __builtin_fmaf(r * 0.5f, e, r);
}

// x^(-1/3) for x > 0: single-precision log2/exp2 seed y (relative error ~1e-7, so e = 1 - x y^3 ~ 3e-7),
// then ONE third-order correction y (1 - e)^(-1/3) = y (1 + e/3 + 2 e^2/9 + O(e^3)): the neglected term is
// ~ (14/81) e^3 < 1e-19, below double rounding (two plain Newton steps cost four more instructions)
__device__ __forceinline__ double rcbrt_pos(double x) {
  const float xf = (float)x;
  const double y = (double)__builtin_amdgcn_exp2f(__builtin_amdgcn_logf(xf) * (-1.0f / 3.0f));
  const double e = __builtin_fma(-(x * (y * y)), y, 1.0);
  const double p = __builtin_fma(e, 2.0 / 9.0, 1.0 / 3.0);
  return __builtin_fma(y * e, p, y);
}
__device__ __forceinline__ float rcbrt_pos(float x) {
  float y = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(x) * (-1.0f / 3.0f));
  float t = y * y;
  float e = __builtin_fmaf(-(x * t), y, 1.0f);
  return __builtin_fmaf(y * (1.0f / 3.0f), e, y);
}

// ---------------------------------------------------------------------------------------------
// Explicit AGPR residency.  gfx950 has a unified 512-entry register file per lane but VALU
// operands must be architectural VGPRs (256); the other half is reachable only through
// v_accvgpr_read/write.  Values that are written once per Newton iteration and read once much
// later (the per-node elimination records) are parked there on purpose with the "a" inline-asm
// register class instead of leaving the choice to the spiller (which sends the overflow to
// scratch memory once both halves are full).
// ---------------------------------------------------------------------------------------------
#ifndef FS_PARK
#define FS_PARK 0   // measured: letting the register allocator place these beats forcing AGPRs
#endif
template <typename R> struct Parked;
#if !FS_PARK
template <typename R> struct Parked {
  R v;
  __device__ __forceinline__ void put(R x) { v = x; }
  __device__ __forceinline__ R get() const { return v; }
};
#else
template <> struct Parked<double> {
  uint32_t lo, hi;
  __device__ __forceinline__ void put(double v) {
    const uint32_t l = (uint32_t)__double2loint(v), h = (uint32_t)__double2hiint(v);
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(lo) : "v"(l));
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(hi) : "v"(h));
  }
  __device__ __forceinline__ double get() const {
    uint32_t l, h;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(l) : "a"(lo));
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(h) : "a"(hi));
    return __hiloint2double((int)h, (int)l);
  }
};
template <> struct Parked<float> {
  uint32_t w;
  __device__ __forceinline__ void put(float v) {
    const uint32_t x = __float_as_uint(v);
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(w) : "v"(x));
  }
  __device__ __forceinline__ float get() const {
    uint32_t x;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(w));
    return __uint_as_float(x);
  }
};
#endif

// |x| as the operand modifier of the instruction that consumes it (a compare-and-select costs four instructions per use)
__device__ __forceinline__ double fabs_(double x) { return __builtin_fabs(x); }
__device__ __forceinline__ float fabs_(float x) { return __builtin_fabsf(x); }
template <typename R> __device__ __forceinline__ R fmax_(R a, R b) { return a > b ? a : b; }
// x clamped to [0, 1] in two instructions (v_max / v_min; a NaN comes out as 0 - the callers' other terms keep it)
__device__ __forceinline__ double clamp01_(double x) { return __builtin_fmin(__builtin_fmax(x, 0.0), 1.0); }
__device__ __forceinline__ float clamp01_(float x) { return __builtin_fminf(__builtin_fmaxf(x, 0.0f), 1.0f); }
// x^b as exp(b log x) for x > 0 (|error| ~ b |log x| ulp: 1e-15 .. 1e-14 relative, against a parity bar of 1e-8): two
// libm calls of ~50 instructions instead of pow()'s ~300 with its special cases, once per Newton iteration in the rating row
#ifndef FS_POW_EXPLOG
#define FS_POW_EXPLOG 1
#endif
__device__ __forceinline__ double pow_(double a, double b) { return (FS_POW_EXPLOG && a > 0.0) ? exp(b * log(a)) : pow(a, b); }
// the same without the fallback (no branch): NaN for a < 0, as pow() gives for the non-integer exponents of a rating curve
#ifndef FS_POW_SHORT
#define FS_POW_SHORT 1   // pow_pos(double): log and exp written out below (~65 instructions) instead of the two libm calls (~150 in the rating row of every Newton iteration: a tenth of C5 fp64)
#endif
// log(x), x > 0 finite: x = 2^e m with m in [sqrt(1/2), sqrt(2)), log m = 2 atanh(f), f = (m - 1) / (m + 1), |f| <= 0.1716: ten terms of the
// odd series (f^21 / 21 < 2e-17 relative), the quotient corrected once, e ln 2 in two parts (the high one exact for |e| < 2^20).  Measured
// against long double over x in 1e-4 .. 1e4: 3.2e-16 relative, 3.1e-16 within 2e-4 of 1 (tools/micro/pow_model.c).
__device__ __forceinline__ double log_pos(double x) {
  double m = __builtin_amdgcn_frexp_mant(x);      // [0.5, 1)
  int e = __builtin_amdgcn_frexp_exp(x);
  const bool low = m < 0.70710678118654752;
  m = low ? m + m : m;
  e = low ? e - 1 : e;
  const double n = m - 1.0, d = m + 1.0;
  const double r = frcp(d);
  double f = n * r;
  f = __builtin_fma(__builtin_fma(-d, f, n), r, f);
  const double f2 = f * f;
  double p = 1.0 / 21.0;
  p = __builtin_fma(p, f2, 1.0 / 19.0); p = __builtin_fma(p, f2, 1.0 / 17.0); p = __builtin_fma(p, f2, 1.0 / 15.0);
  p = __builtin_fma(p, f2, 1.0 / 13.0); p = __builtin_fma(p, f2, 1.0 / 11.0); p = __builtin_fma(p, f2, 1.0 / 9.0);
  p = __builtin_fma(p, f2, 1.0 / 7.0); p = __builtin_fma(p, f2, 1.0 / 5.0); p = __builtin_fma(p, f2, 1.0 / 3.0);
  p = p * f2;
  const double tf = f + f, ed = (double)e;
  const double t = __builtin_fma(ed, 0x1.a39ef35793c76p-33, tf * p);
  return __builtin_fma(ed, 0x1.62e42fee00000p-1, tf + t);
}
// exp(y): y = k ln 2 + r, |r| <= 0.3466, Taylor polynomial of degree 12 (r^13 / 13! < 2e-16), scaled by 2^k (v_ldexp: 0 / inf beyond the range)
__device__ __forceinline__ double exp_short(double y) {
  const double k = __builtin_rint(y * 1.4426950408889634);
  double r = __builtin_fma(-k, 0x1.62e42fefa39efp-1, y);
  r = __builtin_fma(-k, 0x1.abc9e3b39803fp-56, r);
  double p = 1.0 / 479001600.0;
  p = __builtin_fma(p, r, 1.0 / 39916800.0); p = __builtin_fma(p, r, 1.0 / 3628800.0); p = __builtin_fma(p, r, 1.0 / 362880.0);
  p = __builtin_fma(p, r, 1.0 / 40320.0); p = __builtin_fma(p, r, 1.0 / 5040.0); p = __builtin_fma(p, r, 1.0 / 720.0);
  p = __builtin_fma(p, r, 1.0 / 120.0); p = __builtin_fma(p, r, 1.0 / 24.0); p = __builtin_fma(p, r, 1.0 / 6.0);
  p = __builtin_fma(p, r, 0.5); p = __builtin_fma(p, r, 1.0); p = __builtin_fma(p, r, 1.0);
  return __builtin_amdgcn_ldexp(p, (int)k);
}
// x^b for the rating rows: 4.1e-15 relative over x in 1e-4 .. 1e4, b in 0.2 .. 5 - what exp(b * log(x)) of libm gives on the same grid (the rounding
// of b log x dominates both); 0 for x = 0 and NaN for x < 0 as before
__device__ __forceinline__ double pow_pos(double a, double b) {
#if FS_POW_SHORT
  const double v = exp_short(b * log_pos(a));
  return a > 0.0 ? v : (a == 0.0 ? 0.0 : __builtin_nan(""));
#else
  return exp(b * log(a));
#endif
}
__device__ __forceinline__ float pow_pos(float a, float b) { return __builtin_amdgcn_exp2f(b * __builtin_amdgcn_logf(a)); }
// fp32 is the throughput mode (tolerance 1e-3, no parity bar): x^b as exp2(b log2 x) on the transcendental unit, ~1e-6 relative,
// instead of libm's powf (165 instructions in the boundary row of every Newton iteration of C5); NaN for x < 0 and 0 for x = 0, b > 0 as powf
__device__ __forceinline__ float pow_(float a, float b) { return __builtin_amdgcn_exp2f(b * __builtin_amdgcn_logf(a)); }
__device__ __forceinline__ double sqrt_(double a) { return sqrt(a); }
__device__ __forceinline__ float sqrt_(float a) { return sqrtf(a); }

// ---------------------------------------------------------------------------------------------
// node terms: everything the residual / Jacobian needs from one node at state (h, Q)
// ---------------------------------------------------------------------------------------------
template <typename R> struct NodeTerms {
  R A;    // wetted area                         (cross_section.py:623-679)
  R T;    // dA/dh = top width                   (cross_section.py:792-793)
  R Se;   // Sf + Sc                             (channel.py:53-69)
  R eAT;  // (dSe/dA in the reference's mixed convention) * dA/dh / T  (channel.py:71-87; the momentum row's dh entries are
          // used divided by the node's top width, below)
  R eQ;   // dSe/dQ                              (channel.py:89-105)
  R v;    // Q / A
  R rT;   // 1 / T: the continuity row's dh coefficient is T/(2dt) on both nodes (preissmann.py:431-447), its reciprocal scales
};        // the node's momentum entries into the characteristic-like unknowns of the solve (below)

// rows the library appends to the caller's FS_GEO_* table on upload (fs_abi.hip: extend_table): geometry-only quantities
enum { FS_GEOX_SM = FS_GEO_NPARAM, FS_GEOX_SFP, FS_GEOX_TB, FS_GEOX_AM, FS_GEOX_PM, FS_GEOX_RNM, FS_GEOX_KM15, FS_GEOX_KL15,
       FS_GEOX_KR15, FS_GEOX_NROWS };

// trapezoidal section parameters at one node (cross_section.py:569-613) ...
template <typename R> struct SecParams {
  R z, b, m, nm, nl, nr, hbf, bl, br, mfp, curv;
  bool compound;
  // ... and what follows from them alone (a lane's nodes never change during a launch; the table modes read these from
  // rows the library appends to the caller's table on upload, sec_derive() computes them for single sections):
  R sm, sfp;          // sqrt(1 + m_main^2), sqrt(1 + m_fp^2): wetted perimeter of the side slopes
  R Tb, Am, Pm;       // bankfull top width, area and wetted perimeter of the main channel (cross_section.py:660-674)
  R rnm;              // 1 / n_main
  R km15, kl15, kr15; // n^-1.5 of the main channel and the two floodplains (Horton-Einstein sum, cross_section.py:741-754)
};

// Rectangular prismatic fast path: A = b h, P = b + 2h, T = b (cross_section.py:636-639).
// dK/dA / K reduces algebraically to (1 + (2/3) b / P) / A for a rectangle (cross_section.py:756-790
// with hydraulics.py:28-40), so one reciprocal of P*h and one x^(-1/3) serve the whole node.
template <typename R>
__device__ __forceinline__ NodeTerms<R> node_terms_rect(R b, R rb, R n, R h, R Q) {
  NodeTerms<R> t;
  const R A = b * h;
  const R P = fma_(R(2), h, b);
  const R r = frcp(P * h);
  const R rP = r * h, rh = r * P;
  const R Rh = A * rP;                   // hydraulic radius
  const R rA = rh * rb;                  // 1/A
  const R y = rcbrt_pos(Rh);             // R^(-1/3);  K = A R^(2/3)/n  ->  1/K^2 = (n y^2 / A)^2
  const R nk = n * rA * (y * y);
  const R iK2 = nk * nk;
  const R aQ = fabs_(Q);
  t.A = A;
  t.T = b;
  const R w = aQ * iK2;
  t.Se = Q * w;                                                    // hydraulics.py:57
  t.eQ = w + w;                                                    // hydraulics.py:92
  t.eAT = t.Se * fma_(R(-4.0 / 3.0) * b, rP, R(-2)) * rA;          // hydraulics.py:75 (times T, over T)
  t.v = Q * rA;
  t.rT = rb;
  return t;
}

// Simple trapezoid fast path (cross_section.py:641-645): T = b + 2 m h, A = (b + m h) h,
// P = b + 2 h sqrt(1 + m^2).  dK/dA / K = (1 + (2/3)(1 - 2 s A / (T P))) / A with s = sqrt(1+m^2)
// (cross_section.py:756-790, hydraulics.py:28-40); one reciprocal of A*P*T serves 1/A, 1/P, 1/T.
template <typename R>
__device__ __forceinline__ NodeTerms<R> node_terms_trap(R b, R m, R sm2, R n, R h, R Q) {
  NodeTerms<R> t;
  const R mh = m * h;
  const R T = fma_(R(2), mh, b);
  const R A = (b + mh) * h;
  const R P = fma_(sm2, h, b);          // sm2 = 2 sqrt(1 + m^2)
  const R AP = A * P;
  const R r3 = frcp(AP * T);
  const R rA = r3 * (P * T), rP = r3 * (A * T), rT = r3 * AP;
  const R Rh = A * rP;
  const R y = rcbrt_pos(Rh);
  const R nk = n * rA * (y * y);
  const R iK2 = nk * nk;
  const R aQ = fabs_(Q);
  t.A = A;
  t.T = T;
  const R w = aQ * iK2;
  t.Se = Q * w;
  t.eQ = w + w;
  const R f2 = fma_(R(-4.0 / 3.0), fma_(-(sm2 * A * rT), rP, R(1)), R(-2));      // -2 (1 + (2/3)(...))
  t.eAT = t.Se * f2 * rA;
  t.v = Q * rA;
  t.rT = rT;
  return t;
}

// x^(2/3), x^(3/2), x^(-1/3) for x >= 0 without pow(): v_log/v_exp seeded x^(-1/3) and v_sqrt
// (each within a few ulp of the reference's libm pow; 0 -> 0 as 0**p gives in the reference)
template <typename R> __device__ __forceinline__ R p23_(R x) { return x > R(0) ? x * rcbrt_pos(x) : R(0); }
// sqrt(x) for x > 0 as x / sqrt(x) from the reciprocal square root (<= 2e-15; an IEEE fp64 sqrt is ~25 instructions)
template <typename R> __device__ __forceinline__ R fsqrt_pos(R x) { return x * frsq(x); }
template <typename R> __device__ __forceinline__ R p32_(R x) { return x > R(0) ? x * fsqrt_pos(x) : R(0); }

// conveyance of a single sub-section, hydraulics.py:15-26
template <typename R> __device__ __forceinline__ R conv_(R A, R n, R Rh) { return A * p23_(Rh) * frcp(n); }

template <typename R> struct GeneralProps { R A, rA, P, Rh, T, rT, K, rK, neq, dRdA, dKdA, dKdA_K, y13; };

// General trapezoid family (rectangle / simple / compound), straight from the reference including
// the over-bank area inconsistency and the frozen-n_eq dK/dA.  Kept out of line: it is
// pow()-heavy and only the boundary rows and the TABLE geometry mode use it.
// general boundary rows (bc_eval) in line or out of line
#ifndef FS_BC_INLINE
#define FS_BC_INLINE 1   // (0, like FS_GENERAL_INLINE 0: experiment switches, to be built with -mllvm -enable-ipra=0, profiles/round3/polyline_calls.txt)
                         // measured on the one-wave-per-reach kernels: C4 +8 %, C5 +4 % (fp64) / +15 % (fp32); a call inside the Newton loop spills the caller around it
#endif
#if FS_BC_INLINE
#define FS_BC_ATTR __forceinline__
#else
#define FS_BC_ATTR __noinline__
#endif
#ifndef FS_GENERAL_INLINE
#define FS_GENERAL_INLINE 1   // inline the general section evaluation into the fold (measured on C4: 4.5e6 -> 7.1e6)
#endif
#if FS_GENERAL_INLINE
#define FS_GEN_ATTR __forceinline__
#else
#define FS_GEN_ATTR __noinline__
#endif
// x^-1.5 for x > 0
template <typename R> __device__ __forceinline__ R pm15_(R x) { const R r = frsq(x); return r * r * r; }

// the geometry-only members of SecParams from the others (single sections: uniform-geometry modes, post-processing)
template <typename R> __device__ __forceinline__ void sec_derive(SecParams<R> &s) {
  s.sm = sqrt_(R(1) + s.m * s.m); s.sfp = sqrt_(R(1) + s.mfp * s.mfp);
  s.Tb = s.b + R(2) * s.m * s.hbf;
  s.Am = (s.b + s.Tb) / R(2) * s.hbf;                             // :660 (column above omitted)
  s.Pm = s.b + R(2) * s.hbf * s.sm;
  s.rnm = R(1) / s.nm;
  s.km15 = pm15_(s.nm); s.kl15 = s.nl > R(0) ? pm15_(s.nl) : R(0); s.kr15 = s.nr > R(0) ? pm15_(s.nr) : R(0);
}

// K_i^1.5 of one sub-section, K_i = A_i R_i^(2/3) / n_i (hydraulics.py:15-26):  A_i^2.5 / (P_i n_i^1.5) - one reciprocal
// root and one reciprocal instead of an x^(2/3), a division by n and an x^1.5
#ifndef FS_K15_ONE_RSQ
#define FS_K15_ONE_RSQ 1   // A^2.5 / P = A^3 / sqrt(A P^2): one reciprocal root in place of a reciprocal root and a reciprocal
#endif
template <typename R> __device__ __forceinline__ R k15_(R A, R P, R n15) {
#if FS_K15_ONE_RSQ
  return (A > R(0) && P > R(0)) ? (A * A) * (A * frsq(A * (P * P))) * n15 : R(0);
#else
  return (A > R(0) && P > R(0)) ? A * A * fsqrt_pos(A) * frcp(P) * n15 : R(0);
#endif
}

#ifndef FS_GEN_RCP3
#define FS_GEN_RCP3 1
#endif
#ifndef FS_GEN_RK_ALG
#define FS_GEN_RK_ALG 1
#endif
template <typename R>
__device__ FS_GEN_ATTR GeneralProps<R> general_props(const SecParams<R> s, R h) {
  GeneralProps<R> g;
  const R d = fmax_(R(0), h);
  R T = fma_(R(2) * s.m, d, s.b);
  R A = R(0.5) * (s.b + T) * d;
  R P = fma_(R(2) * s.sm, d, s.b);
  R dPdh = R(2) * s.sm;
  const bool over = s.compound && d > s.hbf;
  R K = R(0);
  R cK = R(0);                                                     // over bank: (sum of K_i^1.5)^(-1/3)
  if (over) {
    const R dfp = d - s.hbf;
    const R hm = R(0.5) * s.mfp * dfp;
    const R A_l = (s.bl + hm) * dfp, P_l = fma_(dfp, s.sfp, s.bl);
    const R A_r = (s.br + hm) * dfp, P_r = fma_(dfp, s.sfp, s.br);
    A = s.Am + A_l + A_r;                                           // :660-674 (column above the main channel omitted)
    P = s.Pm + P_l + P_r;
    T = (s.bl + s.Tb + s.br) + R(2) * s.mfp * dfp;
    dPdh = R(2) * s.sfp;
    const R A_m = fma_(s.Tb, dfp, s.Am);                            // :694 (column included)
#if FS_GEN_RK_ALG
    const R S15 = k15_(A_l, P_l, s.kl15) + k15_(A_m, s.Pm, s.km15) + k15_(A_r, P_r, s.kr15);
    cK = S15 > R(0) ? rcbrt_pos(S15) : R(0);                       // K = S^(2/3) = S S^(-1/3), 1/K = (S^(-1/3))^2
    K = S15 * cK;                                                  // :741-754
#else
    K = p23_(k15_(A_l, P_l, s.kl15) + k15_(A_m, s.Pm, s.km15) + k15_(A_r, P_r, s.kr15));      // :741-754
#endif
  }
  // divisions are reciprocal (v_rcp_f64 + one Newton step, 2e-15) times multiply: an IEEE fp64 divide is ~14 instructions
#if FS_GEN_RCP3
  // 1/P, 1/T and 1/A from ONE reciprocal of their product: two v_rcp_f64 and their refinements less per node (C4 +1.8 %).  As before a dry node
  // (A = 0) gets rA = 0 and keeps 1/P, 1/T; a section without width at its bed (P = T = 0 when dry) gets zeros
  const bool wet = A > R(0);
  const R As = wet ? A : R(1);
  const R PT = P * T;
  const R r3 = PT > R(0) ? frcp(PT * As) : R(0);
  const R rP = r3 * (T * As), rT = r3 * (P * As);
#else
  const R rP = P > R(0) ? frcp(P) : R(0), rT = T > R(0) ? frcp(T) : R(0);
#endif
  g.Rh = A * rP;
  const R y13 = g.Rh > R(0) ? rcbrt_pos(g.Rh) : R(0);            // R^(-1/3)
  const R R23 = g.Rh * y13;
  // in bank K = A R^(2/3) / n_main; the reference's round trip (K^1.5)^(2/3) for a compound section in bank (:747-754)
  // is the identity, and its equivalent n = A R^(2/3) / K (:710-739) is n_main there
  if (!over) K = A * R23 * s.rnm;
#if FS_GEN_RK_ALG && FS_GEN_RCP3
  // 1/K without a reciprocal: in bank n / (A R^(2/3)) = n (1/A) (R^(-1/3))^2, over bank the square of the sum's reciprocal cube root
  // (0 where K is 0: a dry node has 1/A = 0, an empty sum has cK = 0)
  const R rA_ = wet ? r3 * PT : R(0);
  g.rK = over ? cK * cK : s.nm * rA_ * (y13 * y13);
#else
  g.rK = K > R(0) ? frcp(K) : R(0);
#endif
  g.neq = over ? A * R23 * g.rK : s.nm;
  g.dRdA = (P <= R(0) || T <= R(0)) ? R(0) : (P - A * (dPdh * rT)) * (rP * rP);   // :766-790
  // dK/dA = (R^(2/3) + (2/3) A R^(-1/3) dR/dA) / n_eq with the frozen n_eq = A R^(2/3) / K (:756-764), so
  // dK/dA / K = 1/A + (2/3) (P/A) dR/dA: neither K nor n_eq has to be divided by
#if FS_GEN_RCP3
  g.rA = wet ? r3 * PT : R(0);
#else
  g.rA = A > R(0) ? frcp(A) : R(0);
#endif
  g.dKdA_K = g.rA * fma_(R(2.0 / 3.0) * P, g.dRdA, R(1));
  g.dKdA = K * g.dKdA_K;
  g.A = A; g.P = P; g.T = T; g.rT = rT; g.K = K; g.y13 = y13;
  return g;
}

// out-of-line copy for the boundary rows (executed by two lanes per reach: keep it out of the hot code)
template <typename R>
__device__ FS_BC_ATTR GeneralProps<R> general_props_call(const SecParams<R> s, R h) { return general_props(s, h); }

// Curvature slope Sc and its derivatives (cross_section.py:145-175 over hydraulics.py:94-153) added to
// (Se, dSe/dA, dSe/dQ).  T = geometric top width, dAdh = what the section reports as dA/dh (the same
// number for the trapezoid family, a finite difference for polylines), y13 = Rh^(-1/3).
template <typename R>
__device__ __forceinline__ void add_curvature(R curv, R A, R rA, R T, R rT, R dAdh, R neq, R y13, R dRdA, R h, R Q, R &Se,
                                              R &dSeA, R &eQ) {
  if (curv == R(0)) return;               // ==0 guard for Sc, <=1e-12 guard for its derivatives
  // every division below is a reciprocal (2e-15) times a multiplication; rA = 1/A, rT = 1/T come from the caller
  const R V = A > R(1e-6) ? Q * rA : Q * R(1e6);                // Q / max(A, 1e-6), hydraulics.py:155-168
  const R D = A * rT;
  const bool deep = T > R(1e-6) && D > R(1e-6);                  // the clamps of froude_num leave D as it is
  const R rs = frsq(R(kG) * D);                                  // (gD)^-0.5
  const R Fr = deep ? V * rs : V * frsq(R(kG) * fmax_(T > R(1e-6) ? D : A * R(1e6), R(1e-6)));
  const R f = R(8) * R(kG) * neq * neq * y13;                   // 8 g / C^2 with C = R^(1/6)/n, :217-229
  const R rsq_f = frsq(f), sq = f * rsq_f;
  const R hF = h * Fr, hF2 = hF * hF;                            // h^2 Fr^2
  const R poly = fma_(R(2.86), sq, R(2.07) * f);
  const R num = poly * hF2;
  const R c2 = curv * curv;
  const R r0 = frcp(R(0.565) + sq);
  const R rden = c2 * r0;                                        // 1 / ((0.565 + sqrt f) rc^2), rc = 1 / curvature
  Se += num * rden;                                             // :94-117
  if (fabs_(curv) > R(1e-12)) {
    const R rs3 = rs * rs * rs;                                   // (gD)^-1.5
    const R Vr = Q * rA;
    const R dFrA = R(-0.5) * Vr * rs3 * R(kG) * rT + (-Vr * rA) * rs;
    const R y2 = y13 * y13;
    const R dfA = -(R(8.0 / 3.0)) * R(kG) * neq * neq * (y2 * y2) * dRdA;
    const R half_rsq = R(0.5) * rsq_f;                          // 1 / (2 sqrt f)
    const R dnum = fma_(R(2.86) * half_rsq, dfA, R(2.07) * dfA) * hF2 +
                   poly * (R(2) * h * rT * Fr * Fr + h * h * R(2) * Fr * dFrA);
    // (dnum den - num dden) / den^2 = (dnum - num dden / den) / den with dden / den = (dfA / (2 sqrt f)) / (0.565 + sqrt f)
    const R dden_den = half_rsq * dfA * r0;
    dSeA += fma_(-num, dden_den, dnum) * rden * dAdh;           // :119-137, x dA_dh (cross_section.py:164)
    const R dFrQ = rA * rs;
    const R dnumq = poly * h * h * R(2) * Fr * dFrQ;
    eQ += dnumq * rden;                                         // :139-153
  }
}

template <typename R>
__device__ FS_GEN_ATTR NodeTerms<R> node_terms_general(const SecParams<R> s, R h, R Q) {
  const GeneralProps<R> g = general_props(s, h);
  NodeTerms<R> t;
  const R iK2 = g.rK * g.rK;
  const R aQ = fabs_(Q);
  const R Sf = Q * aQ * iK2;
  R dSeA = R(-2) * Sf * g.dKdA_K;        // per unit area
  R Se = Sf, eQ = R(2) * aQ * iK2;
  add_curvature(s.curv, g.A, g.rA, g.T, g.rT, g.T, g.neq, g.y13, g.dRdA, h, Q, Se, dSeA, eQ);
  t.A = g.A; t.T = g.T; t.Se = Se; t.eAT = dSeA; t.eQ = eQ; t.v = Q * g.rA;    // (dSeA T) / T
  t.rT = g.rT;
  return t;
}

// ---------------------------------------------------------------------------------------------
// boundary rows  (boundary.py:56-242)
// ---------------------------------------------------------------------------------------------
template <typename R> struct BCDesc {
  int32_t kind;
  int32_t stride;          // 0: params shared by all reaches, 1: params[i*B + reach]
  const R *params;
  const R *target;         // [levels][B] or nullptr
  R tgt;                   // device side: target[level][reach] of the level being solved (loaded once per level)
};
template <typename R> struct BCRow { R dh, dq, res; };   // dh*d(h) + dq*d(Q) = -res

template <typename R>
__device__ __forceinline__ R bc_param(const BCDesc<R> &bc, int i, int reach, int B) {
  return bc.stride ? bc.params[(size_t)i * B + reach] : bc.params[i];
}

// RCP: reciprocal instead of an IEEE division (the general rows, where the row is on the critical path of a one-wave
// reach: C4 +1.5 %; the rectangular rows keep the division - the flagship kernel, which never executes this row, loses
// 0.5 % to the different register allocation around it otherwise)
template <bool RCP, typename R>
__device__ __forceinline__ R rating_blend(R z, R s0, R buf, R l0, R l1, R l2, R h0, R h1, R h2) {
  R al;
  if (RCP) {
    // branch-free: the smoothstep of the clamped argument is exactly 0 / 1 outside the buffer (three evaluations per row,
    // each with two divergent branches otherwise: 12 % of a C4 iteration went into this row)
    const R s = clamp01_((z - s0) * frcp(buf));
    al = R(3) * s * s - R(2) * s * s * s;
  } else if (z >= s0 + buf) al = R(1);
  else if (z <= s0) al = R(0);
  else { const R s = (z - s0) / buf; al = R(3) * s * s - R(2) * s * s * s; }
  const R lo = l0 + l1 * z + l2 * z * z;
  const R hi = h0 + h1 * z + h2 * z * z;
  return (R(1) - al) * lo + al * hi;
}

// ---- general LumpedStorage behind a fixed_depth boundary (FS_BC_STORAGE_CURVE) ----
// what the entrance-loss terms need from the boundary node's section at one stage
template <typename R> struct EntryProps { R A, Rh, neq, dRdA, dAdh; };

template <typename R> struct StorageCurve {
  const BCDesc<R> &bc; int reach, B, nc;
  __device__ __forceinline__ R p(int i) const { return bc_param(bc, i, reach, B); }
  __device__ __forceinline__ R xs(int j) const { return p(FS_SC_NFIXED + j); }
  __device__ __forceinline__ R ys(int j) const { return p(FS_SC_NFIXED + nc + j); }
  // lumped_storage.py:152-157 (np.interp: clamped outside the table)
  __device__ R area_at(R Y) const {
    if (nc == 0) return p(FS_SC_SURFACE_AREA);
    const R x = Y + p(FS_SC_BETA);
    R a;
    if (x <= xs(0)) a = ys(0);
    else if (x >= xs(nc - 1)) a = ys(nc - 1);
    else {
      int lo = 0, hi = nc - 1;
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (xs(mid) <= x) lo = mid; else hi = mid; }
      const R slope = (ys(lo + 1) - ys(lo)) / (xs(lo + 1) - xs(lo));
      a = slope * (x - xs(lo)) + ys(lo);
    }
    return p(FS_SC_ALPHA) * a;
  }
  // lumped_storage.py:166-179: n-point trapezoid over np.linspace(Y1, Y2, n) when n = int(|Y2-Y1|/step) > 2
  __device__ R net_vol(R Y1, R Y2, R step) const {
    if (nc == 0) return (Y2 - Y1) * p(FS_SC_SURFACE_AREA);
    const int n = (int)(fabs_(Y2 - Y1) / step);
    if (n > 2) {
      const R d = (Y2 - Y1) / R(n - 1);
      R sum = R(0), y0 = Y1, a0 = area_at(Y1);
      for (int i = 1; i < n; ++i) {
        const R y1 = i == n - 1 ? Y2 : Y1 + R(i) * d;
        const R a1 = area_at(y1);
        sum += (y1 - y0) * (a1 + a0) / R(2);
        y0 = y1; a0 = a1;
      }
      return sum;
    }
    return R(0.5) * (area_at(Y2) + area_at(Y1)) * (Y2 - Y1);
  }
  __device__ R outflow(R Y) const {                                     // rating_curve.py:50-61
    const int type = (int)p(FS_SC_RC_TYPE);
    if (type == 0) return R(0);
    const R x = Y + p(FS_SC_RC_SHIFT);
    return type == 2 ? p(FS_SC_RC_A) * x * x + p(FS_SC_RC_B) * x + p(FS_SC_RC_C) : p(FS_SC_RC_A) * pow_(x, p(FS_SC_RC_B));
  }
};

// Root of the reservoir mass balance on [Y_min, Y_max] (lumped_storage.py:24-31).  The reference calls
// scipy.optimize.brentq with its defaults (xtol 2e-12, rtol 4 eps, 100 iterations); this is that
// published algorithm (Brent 1973 as arranged in scipy/optimize/Zeros/brentq.c: bisection guarded
// secant / inverse quadratic extrapolation on a bracketing triple), so the iterates - and with them
// where the n-point trapezoid of net_vol switches its n - follow the reference's.
template <typename R>
__device__ R storage_root(const StorageCurve<R> &sc, R Yold, R vol_in, R dt, R step, int *flag) {
  const R qold = sc.outflow(Yold);
  const bool has_rc = (int)sc.p(FS_SC_RC_TYPE) != 0;
  auto f = [&](R Y) {
    const R qout = has_rc ? R(0.5) * (qold + sc.outflow(Y)) : R(0);
    return sc.net_vol(Yold, Y, step) - (vol_in - qout * dt);
  };
  const R xtol = R(2e-12), rtol = R(8.881784197001252e-16);
  R xpre = sc.p(FS_SC_Y_MIN), xcur = sc.p(FS_SC_Y_MAX), xblk = R(0);
  R fpre = f(xpre), fcur = f(xcur), fblk = R(0), spre = R(0), scur = R(0);
  if (fpre == R(0)) return xpre;
  if (fcur == R(0)) return xcur;
  if ((fpre < R(0)) == (fcur < R(0)) || !(fpre == fpre) || !(fcur == fcur)) {   // brentq: "f(a) and f(b) must have different signs"
    *flag = FS_STORAGE_RANGE;
    return xcur;
  }
  for (int it = 0; it < 100; ++it) {
    if (fpre != R(0) && fcur != R(0) && ((fpre < R(0)) != (fcur < R(0)))) {
      xblk = xpre; fblk = fpre;
      spre = scur = xcur - xpre;
    }
    if (fabs_(fblk) < fabs_(fcur)) {
      xpre = xcur; xcur = xblk; xblk = xpre;
      fpre = fcur; fcur = fblk; fblk = fpre;
    }
    const R delta = (xtol + rtol * fabs_(xcur)) / R(2);
    const R sbis = (xblk - xcur) / R(2);
    if (fcur == R(0) || fabs_(sbis) < delta) return xcur;
    if (fabs_(spre) > delta && fabs_(fcur) < fabs_(fpre)) {
      R stry;
      if (xpre == xblk) {
        stry = -fcur * (xcur - xpre) / (fcur - fpre);                    // secant
      } else {
        const R dpre = (fpre - fcur) / (xpre - xcur), dblk = (fblk - fcur) / (xblk - xcur);
        stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre));   // inverse quadratic
      }
      const R lim = fmin(fabs_(spre), R(3) * fabs_(sbis) - delta);
      if (R(2) * fabs_(stry) < lim) { spre = scur; scur = stry; }
      else { spre = sbis; scur = sbis; }
    } else {
      spre = sbis; scur = sbis;
    }
    xpre = xcur; fpre = fcur;
    if (fabs_(scur) > delta) xcur += scur;
    else xcur += (sbis > R(0) ? delta : -delta);
    fcur = f(xcur);
  }
  return xcur;
}

// fixed_depth + general LumpedStorage: residual and derivatives of boundary.py:97-133, :152-164, :213-237.
// pr: section at hw = z_min + depth (residual), pd: at hw = depth + bed_level (derivatives).
template <typename R>
__device__ __forceinline__ BCRow<R> bc_storage_curve(const BCDesc<R> bc, int reach, int B, int level, const EntryProps<R> pr,
                                                  const EntryProps<R> pd, R h, R Q, R Qold, R dt, R Yprev, R *Ynew,
                                                  int *flag) {
  StorageCurve<R> sc{bc, reach, B, 0};
  sc.nc = (int)sc.p(FS_SC_N_CURVE);
  const R bed = sc.p(FS_SC_BED_LEVEL), ymin = sc.p(FS_SC_MIN_STAGE);
  R step = R(1);
  for (int j = 0; j + 1 < sc.nc; ++j) {
    const R d = fabs_(sc.xs(j + 1) - sc.xs(j));
    step = j == 0 ? d : (d < step ? d : step);
  }
  const R vol = R(0.5) * (Qold + Q) * dt;                               // preissmann.py:314
  const R Yold = level == 1 ? h + bed : Yprev;                          // boundary.py:104-108 (k==1 quirk)
  R Y = storage_root(sc, Yold, vol, dt, step, flag);
  if (Y < ymin) Y = ymin;                                               // lumped_storage.py:32-33
  *Ynew = Y;
  const R dY = Y <= ymin ? R(0) : R(1) / sc.area_at(Y);                 // lumped_storage.py:37-45
  R hl = R(0), dhlA = R(0), dhlQ = R(0);
  if (sc.p(FS_SC_CAPTURE_LOSSES) > R(0.5)) {                            // lumped_storage.py:47-56, :95-147
    const R Lr = sc.p(FS_SC_RESERVOIR_LENGTH), Kq = sc.p(FS_SC_K_Q);
    const R aQ = fabs_(Q);
    {
      const R K = conv_(pr.A, pr.neq, pr.Rh), V = Q / pr.A;
      hl = Q * aQ / (K * K) * Lr + Kq * V * V / (R(2) * R(kG));
    }
    const R y13 = pd.Rh > R(0) ? rcbrt_pos(pd.Rh) : R(0);
    const R R23 = pd.Rh * y13;
    const R K = pd.A * R23 / pd.neq;
    const R dK = (R23 + pd.A * R(2.0 / 3.0) * y13 * pd.dRdA) / pd.neq;   // hydraulics.py:28-40
    const R V = Q / pd.A;
    dhlA = R(-2) * (Q * aQ / (K * K)) * (dK / K) * Lr + Kq * R(2) * V * (-Q / (pd.A * pd.A)) / (R(2) * R(kG));
    dhlQ = R(2) * aQ / (K * K) * Lr + Kq * R(2) * V * (R(1) / pd.A) / (R(2) * R(kG));
  }
  BCRow<R> r;
  r.res = h - (Y + hl - bed);
  r.dh = R(1) - dhlA * pd.dAdh;
  r.dq = R(0) - (dY * R(0.5) * dt + dhlQ);
  return r;
}

template <typename R> __device__ __forceinline__ EntryProps<R> entry_props(const GeneralProps<R> &g) {
  EntryProps<R> e;
  e.A = g.A; e.Rh = g.Rh; e.neq = g.neq; e.dRdA = g.dRdA; e.dAdh = g.T;
  return e;
}

__host__ __device__ constexpr bool bc_is_storage(int kind) { return kind == FS_BC_STORAGE || kind == FS_BC_STORAGE_CURVE; }

// sec: section of the boundary node; Qold: flow[k-1] at that node; Yprev: storage stage of level
// k-1; level: k.  Ynew returns the storage stage implied by this evaluation (boundary.py:126-131).
// WITH_SC = false compiles the general storage row out (kernels of boundary class 0, fs_kernel.hpp): it is the one
// row with an out-of-line call (the Brent iteration), which costs every caller 90-130 registers and its scratch.
// The parameters of the closed-form kinds (up to FS_BC_STORAGE) are read from the kernel's LDS copy (the prologue of
// preissmann_step_kernel points bc.params there for exactly these kinds): ds_read with immediate offsets.  Through the
// generic pointer every parameter was a flat load behind an address the compiler kept in (spilled) scalar registers -
// 4 instructions and a full memory wait per parameter, in a row one lane evaluates while the others wait (C4: +7.7 %).
template <typename R> using LdsParams = const __attribute__((address_space(3))) R *;
template <bool WITH_SC = true, typename R>
__device__ FS_BC_ATTR BCRow<R> bc_eval(const BCDesc<R> bc, int reach, int B, int level, const SecParams<R> sec,
                                         R h, R Q, R Qold, R dt, R Yprev, R *Ynew, int *flag) {
  BCRow<R> r;
  auto p = [&](int i) { return ((LdsParams<R>)bc.params)[i]; };      // kinds <= FS_BC_STORAGE only (bc_storage_curve reads its own)
  switch (bc.kind) {
    case FS_BC_FLOW_HYDROGRAPH:
      r.res = Q - bc.tgt; r.dh = R(0); r.dq = R(1); break;
    case FS_BC_STAGE_HYDROGRAPH:
      r.res = h - (bc.tgt - p(0)); r.dh = R(1); r.dq = R(0); break;
    case FS_BC_FIXED_DEPTH:
      r.res = h - p(0); r.dh = R(1); r.dq = R(0); break;
    case FS_BC_NORMAL_DEPTH: {
      const R S0 = p(0), bed = p(1);
      const R sg = S0 < R(0) ? R(-1) : R(1);
      const R rt = sqrt_(fabs_(S0));
      const GeneralProps<R> gr = general_props_call(sec, h);                 // residual: hw = z_min + h
      const R hd = h + bed - sec.z;                                           // df_dh: hw = h + bed_level
      GeneralProps<R> gd = gr;                                                // usually the same depth: one evaluation
      if (hd != h) gd = general_props_call(sec, hd);
      r.res = Q - sg * gr.K * rt;                                       // hydraulics.py:4-13
      r.dh = R(0) - sg * gd.dKdA * rt * gd.T;                           // hydraulics.py:206-215
      r.dq = R(1);
    } break;
    case FS_BC_RATING_POWER: {
      const R x = p(3) + h + p(2);
      const R q = p(0) * pow_pos(x, p(1));                              // rating_curve.py:59
      r.res = Q - q;
      r.dh = R(0) - p(1) * q * frcp(x);                                 // a b x^(b-1) (:143) from the one pow()
      r.dq = R(1);
    } break;
    case FS_BC_RATING_POLY: {
      const R x = p(4) + h + p(3);
      r.res = Q - (p(0) * x * x + p(1) * x + p(2));
      r.dh = R(0) - (p(0) * R(2) * x + p(1));
      r.dq = R(1);
    } break;
    case FS_BC_RATING_BLEND: {
      const R z = p(9) + h, dY = p(8);
      const R q0 = rating_blend<true>(z, p(0), p(1), p(2), p(3), p(4), p(5), p(6), p(7));
      const R qp = rating_blend<true>(z + dY, p(0), p(1), p(2), p(3), p(4), p(5), p(6), p(7));
      const R qm = rating_blend<true>(z - dY, p(0), p(1), p(2), p(3), p(4), p(5), p(6), p(7));
      r.res = Q - q0;
      r.dh = R(0) - (qp - qm) * frcp(R(2) * dY);                            // roseires_rating_curve.py:202-208
      r.dq = R(1);
    } break;
    case FS_BC_STORAGE: {
      const R area = p(0), ymin = p(1), bed = p(4);
      const R vol = R(0.5) * (Qold + Q) * dt;                           // preissmann.py:314
      const R Yold = level == 1 ? h + bed : Yprev;                      // boundary.py:104-108 (k==1 quirk)
      R Y = Yold + vol / area;                                          // root of lumped_storage.py:25-28
      if (!(Y >= p(2) && Y <= p(3))) *flag = FS_STORAGE_RANGE;          // brentq would raise
      if (Y < ymin) Y = ymin;
      *Ynew = Y;
      r.res = h - (Y - bed);
      r.dh = R(1);
      r.dq = R(0) - (Y <= ymin ? R(0) : R(1) / area) * R(0.5) * dt;     // boundary.py:213-237
    } break;
    case FS_BC_STORAGE_CURVE: {
      if (!WITH_SC) { r.res = R(0); r.dh = R(1); r.dq = R(0); break; }      // never launched (fs_abi.hip: pick_kernel)
      const R bed = bc_param(bc, FS_SC_BED_LEVEL, reach, B);             // (this kind's parameters stay in global memory)
      return bc_storage_curve(bc, reach, B, level, entry_props(general_props_call(sec, h)),
                              entry_props(general_props_call(sec, h + bed - sec.z)), h, Q, Qold, dt, Yprev, Ynew, flag);
    }
    case FS_BC_HOST_ROW:
      // the row was evaluated by the caller at this Newton vector (fs_batch_set_host_rows; kernels of class -1 only)
      if (WITH_SC) { r.dh = bc_param(bc, 0, reach, B); r.dq = bc_param(bc, 1, reach, B); r.res = bc_param(bc, 2, reach, B); }
      else { r.res = R(0); r.dh = R(1); r.dq = R(0); }
      break;
    default:
      r.res = R(0); r.dh = R(1); r.dq = R(0); break;
  }
  return r;
}

// Rectangular prismatic reaches: same rows with the closed-form conveyance of node_terms_rect and
// no out-of-line call (a call inside the Newton loop makes the caller spill its register-resident
// state around it).  Only the kinds that need no pow() are inlined (bc_is_light): the power rating
// curve drags ~50 SGPR constants and ~300 instructions of pow() into the hot loop otherwise.
// zsec = bed level of the boundary node's section.
__host__ __device__ constexpr bool bc_is_light(int kind) { return kind != FS_BC_RATING_POWER && kind < FS_BC_STORAGE_CURVE; }

// lp: this reach's parameters in LDS (typed pointer: ds_read, not a flat load through a generic one)
template <typename R>
__device__ __forceinline__ BCRow<R> bc_eval_rect(const BCDesc<R> &bc, LdsParams<R> lp, int level, R b, R n, R zsec,
                                                 R h, R Q, R Qold, R dt, R Yprev, R *Ynew, int *flag) {
  BCRow<R> r;
  auto p = [&](int i) { return lp[i]; };
  switch (bc.kind) {
    case FS_BC_FLOW_HYDROGRAPH:
      r.res = Q - bc.tgt; r.dh = R(0); r.dq = R(1); break;
    case FS_BC_STAGE_HYDROGRAPH:
      r.res = h - (bc.tgt - p(0)); r.dh = R(1); r.dq = R(0); break;
    case FS_BC_FIXED_DEPTH:
      r.res = h - p(0); r.dh = R(1); r.dq = R(0); break;
    case FS_BC_NORMAL_DEPTH: {
      // p(2) = sign(S0) sqrt|S0|, prepared once per launch next to the cached parameters (fs_kernel.hpp)
      const R bed = p(1), srt = p(2);
      const R rn = frcp(n);
      // K = A R^(2/3) / n ; dK/dA * T = K (1 + (2/3) b/P) / h        (rectangle)
      const R hd = h + bed - zsec;                                      // df_dh uses hw = h + bed_level
      const R P = fma_(R(2), h, b);
      if (hd == h) {
        // the usual case (boundary bed level == section bed): one reciprocal serves both evaluations,
        // 1/h = P/(P h) and b/P = b h/(P h)
        const R q = frcp(P * h);
        const R Rh = b * h * h * q;
        const R K = b * h * Rh * rcbrt_pos(Rh) * rn;
        r.res = Q - srt * K;
        r.dh = R(0) - srt * K * fma_(R(2.0 / 3.0) * b * h, q, R(1)) * (P * q);
      } else {
        const R Rh = b * h * frcp(P);
        const R K = b * h * Rh * rcbrt_pos(Rh) * rn;
        const R Pd = fma_(R(2), hd, b);
        const R rPd = frcp(Pd);
        const R Rd = b * hd * rPd;
        const R Kd = b * hd * Rd * rcbrt_pos(Rd) * rn;
        r.res = Q - srt * K;
        r.dh = R(0) - srt * Kd * fma_(R(2.0 / 3.0) * b, rPd, R(1)) * frcp(hd);
      }
      r.dq = R(1);
    } break;
    case FS_BC_RATING_POLY: {
      const R x = p(4) + h + p(3);
      r.res = Q - (p(0) * x * x + p(1) * x + p(2));
      r.dh = R(0) - (p(0) * R(2) * x + p(1));
      r.dq = R(1);
    } break;
    case FS_BC_RATING_BLEND: {
      const R z = p(9) + h, dY = p(8);
      const R q0 = rating_blend<false>(z, p(0), p(1), p(2), p(3), p(4), p(5), p(6), p(7));
      const R qp = rating_blend<false>(z + dY, p(0), p(1), p(2), p(3), p(4), p(5), p(6), p(7));
      const R qm = rating_blend<false>(z - dY, p(0), p(1), p(2), p(3), p(4), p(5), p(6), p(7));
      r.res = Q - q0;
      r.dh = R(0) - (qp - qm) / (R(2) * dY);
      r.dq = R(1);
    } break;
    case FS_BC_STORAGE: {
      const R area = p(0), ymin = p(1), bed = p(4);
      const R vol = R(0.5) * (Qold + Q) * dt;
      const R Yold = level == 1 ? h + bed : Yprev;
      R Y = Yold + vol / area;
      if (!(Y >= p(2) && Y <= p(3))) *flag = FS_STORAGE_RANGE;
      if (Y < ymin) Y = ymin;
      *Ynew = Y;
      r.res = h - (Y - bed);
      r.dh = R(1);
      r.dq = R(0) - (Y <= ymin ? R(0) : R(1) / area) * R(0.5) * dt;
    } break;
    default:
      r.res = R(0); r.dh = R(1); r.dq = R(0); break;
  }
  return r;
}

// ---------------------------------------------------------------------------------------------
// The linear solve (reference: scipy.sparse.linalg.spsolve on the 2N x 2N banded Jacobian, preissmann.py:146).
//
// Characteristic-like unknowns.  The continuity row of cell i (preissmann.py:431-491) carries the same two numbers on
// both of its nodes: t = T/(2dt) on dh and -+cq = theta/dx on dQ.  With
//       p_i = t_i dh_i + cq dQ_i ,   m_i = t_i dh_i - cq dQ_i
// it reads  m_i + p_{i+1} = rc_i : every p but the first is an m and a number - no division, no pivot.  What remains is
// ONE scalar tridiagonal system in (p_0, m_0, ..., m_{N-1}); row i is the momentum row of cell i,
//       al_i p_i + D_i m_i + de_i m_{i+1} = rho0_i ,     p_i = rc_{i-1} - m_{i-1}   (i > 0)
//       al, be = pm0/(2t_i) +- pm1/(2cq) ,  ga, de = sm0/(2t_{i+1}) +- sm1/(2cq) ,  D = be - ga ,  rho0 = qm - ga rc
// followed by the downstream boundary row (on p_{N-1}, m_{N-1}) and identity rows up to the lane grid; the upstream
// boundary row closes the system on the left.  Frictionless and subcritical the rows are diagonally dominant
// (|D| = 2(a + k) against |a - k + v| + |a - k - v|, a = theta dt (c^2 - v^2)/dx, k = dx/(4 theta dt)), friction adds to
// the diagonal: the elimination needs no pivoting, and it has none of the singular 2x2 pivot blocks the block order of
// the classical double sweep meets on steep shallow reaches (tests/partition_model.py is the numpy model of all this,
// tests/test_partition_model.py compares it with SuperLU).
//
// A run of rows [a, b) is a "segment": what is left between its four boundary unknowns once the interior is eliminated,
//       up  :  u1 p_a +    m_a + u3 m_{b-1}           = ru        (the first row, its far unknown substituted)
//       down:  d1 p_a        + d2 m_{b-1} + d3 m_b    = rd        (forward elimination, fill-in column of p_a)
// plus rc of its last row, which links the next segment: p_b = rc - m_{b-1}.  The coefficient of m_a in the up row is 1
// by construction and stays 1 through every merge, so a segment is 8 numbers.
// ---------------------------------------------------------------------------------------------
template <typename R> struct Row { R al, D, de, rho0, rc; };                 // one row of the scalar system
template <typename R> struct Seg { R u1, u3, ru, d1, d2, d3, rd, rc; };
// what the way down needs to recover the separator m_{b-1} = -A1 p_a + A2 m_{c-1} + A3 of a merge and the p of the
// right half's first row, p_b = rc - m_{b-1}
template <typename R> struct Elim { R A1, A2, A3, rc; };

__device__ __forceinline__ float abs_mod(float x) { return __builtin_fabsf(x); }     // source modifier, no instruction
__device__ __forceinline__ double abs_mod(double x) { return __builtin_fabs(x); }

// merges X = [a, b) and Y = [b, c): m_{b-1} is eliminated from {X.down, Y.up}; m_b appears in no outer row and drops out
template <typename R>
__device__ __forceinline__ void merge(const Seg<R> &X, const Seg<R> &Y, Seg<R> &Z, Elim<R> &e) {
  const R ru2 = fma_(-Y.u1, X.rc, Y.ru);          // Y's rows on m_{b-1} instead of p_b
  const R rd2 = fma_(-Y.d1, X.rc, Y.rd);
  const R det = fma_(X.d3, Y.u1, X.d2);           // | d2 d3 ; -u1' 1 |
#ifdef FS_FAKE_TREE_RCP      // timing experiment only (wrong numbers): the merge without its reciprocal - what a reciprocal-free tree could gain at most
  const R r = det;
#else
  const R r = frcp(det);
#endif
  const R g2 = X.d3 * r;
  const R A1 = r * X.d1, A2 = g2 * Y.u3, A3 = fma_(r, X.rd, -(g2 * ru2));
  e.A1 = A1; e.A2 = A2; e.A3 = A3; e.rc = X.rc;
  Seg<R> z;
  z.u1 = fma_(-X.u3, A1, X.u1); z.u3 = X.u3 * A2; z.ru = fma_(-X.u3, A3, X.ru);
  z.d1 = Y.d1 * A1; z.d2 = fma_(-Y.d1, A2, Y.d2); z.d3 = Y.d3; z.rd = fma_(Y.d1, A3, rd2);
  z.rc = Y.rc;
  Z = z;
}

// the separator of a merge from the two numbers its group carries on the way down
template <typename R> __device__ __forceinline__ R separator(const Elim<R> &e, R pL, R mR) {
  return fma_(-e.A1, pL, fma_(e.A2, mR, e.A3));
}

// Root segment S (all rows; nothing right of its last row, so d3 multiplies no unknown) and the upstream boundary row
// aU p_0 + bU m_0 = rU  ->  p_0, m_0 and the m of the last row.
template <typename R>
__device__ __forceinline__ void close_root(const Seg<R> &S, R aU, R bU, R rU, R &p0, R &m0, R &mlast) {
  const R r = frcp(S.d2);
  const R f = S.u3 * r;
  const R e1 = fma_(-f, S.d1, S.u1), e3 = fma_(-f, S.rd, S.ru);       // e1 p_0 + m_0 = e3
  const R rdet = frcp(fma_(-bU, e1, aU));
  p0 = fma_(-bU, e3, rU) * rdet;
  m0 = fma_(-e1, p0, e3);
  mlast = fma_(-S.d1, p0, S.rd) * r;
}

// ---------------------------------------------------------------------------------------------
// cross-lane plumbing (wave = 64 lanes)
// ---------------------------------------------------------------------------------------------

// Cross-lane moves without the LDS crossbar: DPP modifiers on v_mov_b32 (gfx9 family).
//   row_shr:n / row_shl:n  move within a row of 16 lanes; row_bcast15 / row_bcast31 hand lane 15 of
//   each row to the next row / lane 31 to rows 2-3.  The tree only ever needs lane-d for a lane
//   whose low bits are all ones, which these patterns cover for every stride (see fs_kernel.hpp).
#ifndef FS_DPP
#define FS_DPP 1
#endif
#ifndef FS_DPP_TIED
#define FS_DPP_TIED 0   // 0: bound_ctrl moves without a tied destination (no register copy per move); 1: copy + in-place DPP
#endif
// lanes without a source read 0 (bound_ctrl): their value is unspecified for every caller below, and an
// "old" operand that never shows through spares the register copy that keeps v alive next to its shifted copy
template <int CTRL> __device__ __forceinline__ int dpp_mov(int v) {
  return FS_DPP_TIED ? __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false) : __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v) {
  const int lo = dpp_mov<CTRL>(__double2loint(v)), hi = dpp_mov<CTRL>(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(dpp_mov<CTRL>(__float_as_int(v)));
}
// the same move into the lanes of the banks (groups of 4 lanes within a row of 16) in BANKS only; the others keep `old`
template <int CTRL, int BANKS> __device__ __forceinline__ int dpp_mov_banks(int old, int v) {
  return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xF, BANKS, false);
}
template <int CTRL, int BANKS> __device__ __forceinline__ double dpp_mov_banks(double old, double v) {
  const int lo = dpp_mov_banks<CTRL, BANKS>(__double2loint(old), __double2loint(v));
  const int hi = dpp_mov_banks<CTRL, BANKS>(__double2hiint(old), __double2hiint(v));
  return __hiloint2double(hi, lo);
}
template <int CTRL, int BANKS> __device__ __forceinline__ float dpp_mov_banks(float old, float v) {
  return __int_as_float(dpp_mov_banks<CTRL, BANKS>(__float_as_int(old), __float_as_int(v)));
}
// value of lane - D for lanes whose low log2(2D) bits are all ones (others: unspecified)
template <int D, typename R> __device__ __forceinline__ R tree_from_below(R v) {
#if FS_DPP
  if (D < 16) return dpp_mov<0x110 + (D < 16 ? D : 1)>(v);      // row_shr:D
  if (D == 16) return dpp_mov<0x142>(v);                          // row_bcast15
  return dpp_mov<0x143>(v);                                       // row_bcast31
#else
  return __shfl_up(v, D, 64);
#endif
}
template <int D, typename R> __device__ __forceinline__ Seg<R> seg_from_below(const Seg<R> &s) {
  Seg<R> o;
  o.u1 = tree_from_below<D>(s.u1); o.u3 = tree_from_below<D>(s.u3); o.ru = tree_from_below<D>(s.ru);
  o.d1 = tree_from_below<D>(s.d1); o.d2 = tree_from_below<D>(s.d2); o.d3 = tree_from_below<D>(s.d3);
  o.rd = tree_from_below<D>(s.rd); o.rc = tree_from_below<D>(s.rc);
  return o;
}
// the value lane `src` holds, in every lane (src: wave-uniform - a constant or a scalar register)
__device__ __forceinline__ double read_lane(double v, int src) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
__device__ __forceinline__ float read_lane(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
template <int CTRL, int ROWS, int BANKS> __device__ __forceinline__ int dpp_zero(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROWS, BANKS, true);     // lanes without a source (or masked off) get 0
}
template <int CTRL, int ROWS, int BANKS> __device__ __forceinline__ double dpp_zero(double v) {
  return __hiloint2double(dpp_zero<CTRL, ROWS, BANKS>(__double2hiint(v)), dpp_zero<CTRL, ROWS, BANKS>(__double2loint(v)));
}
template <int CTRL, int ROWS, int BANKS> __device__ __forceinline__ float dpp_zero(float v) {
  return __int_as_float(dpp_zero<CTRL, ROWS, BANKS>(__float_as_int(v)));
}
// sum over the 64 lanes, the same value in every lane (fixed order: prefix sums inside each row of 16,
// then across rows; lane 63 holds the total and is broadcast through a scalar register)
template <typename R> __device__ __forceinline__ R wave_sum(R v) {
#if FS_DPP
  v += dpp_zero<0x111, 0xF, 0xF>(v);        // row_shr:1
  v += dpp_zero<0x112, 0xF, 0xF>(v);        // row_shr:2
  v += dpp_zero<0x114, 0xF, 0xF>(v);        // row_shr:4
  v += dpp_zero<0x118, 0xF, 0xF>(v);        // row_shr:8   -> lane 15 of a row: the row's sum
  v += dpp_zero<0x142, 0xA, 0xF>(v);        // row_bcast15 into rows 1 and 3
  v += dpp_zero<0x143, 0xC, 0xF>(v);        // row_bcast31 into rows 2 and 3 -> lane 63: total
  if constexpr (sizeof(R) == 8) {
    const double d = v;
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(d), 63), __builtin_amdgcn_readlane(__double2loint(d), 63));
  } else {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
  }
#else
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
#endif
}

}  // namespace fs
