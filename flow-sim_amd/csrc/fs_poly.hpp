// fs_poly.hpp - polyline ("irregular") cross-sections on the device, fp64.
//
// What the reference's IrregularSection does per call (src/hydromodel/cross_section.py:207-543):
// wetted area / perimeter / top width by walking the polyline, composite roughness over left / main
// / right strips, central differences with dh = 1e-6 for dR/dA and dA/dh, and - when the water
// surface splits the section into two or more wetted runs of >= 2 vertices - the conveyance sum over
// temporary sub-sections.  Here one pass over the vertices accumulates, edge by edge, everything a
// (sub-)section evaluation needs: (A, P) at the three stages hw, hw -+ dh, the top width at hw and
// the (A, P) of the three roughness strips at hw.
//
// Reference behaviour that is kept (tests/golden/irr_*.npz pin it):
//   * a vertex exactly on the water surface is neither wet nor above: its edge contributes nothing
//     (:262, :292, :300);
//   * the left water's-edge point of a temporary sub-section sits at the x of the first wet vertex
//     (np.interp over a decreasing abscissa, :357); the right one is interpolated (:361);
//   * evaluated at its own creation stage a sub-section loses the edge triangles, at +dh it has
//     them: that is what its finite-difference dR/dA sees (:523-531).
#pragma once
#include "fs_device.hpp"

#ifndef FS_POLY_INLINE
#define FS_POLY_INLINE 1   // polyline walk inlined into the fold (+37 % on the polyline ensemble): a call inside the Newton loop spills its caller
                           // (0 is an experiment switch: build it with -mllvm -enable-ipra=0, profiles/round3/polyline_calls.txt)
#endif
#if FS_POLY_INLINE
#define FS_POLY_ATTR __forceinline__
#else
#define FS_POLY_ATTR __noinline__
#endif

#ifndef FS_POLY_PREFETCH
#define FS_POLY_PREFETCH 1
#endif

namespace fs {

// vertex j of a node lives at x[j * stride], z[j * stride] (vertex-major [P][N] tables: the lanes
// of a wave walk different nodes in step)
// -DFS_BOUNDS (audit builds only): every index of the polyline path against its allocation; a violation is printed, the index clamped to 0
#ifdef FS_BOUNDS
__device__ __forceinline__ long fs_chk(int site, long idx, long lim) {
  if (idx < 0 || idx >= lim) { printf("FS_BOUNDS site %d idx %ld lim %ld\n", site, idx, lim); return 0; }
  return idx;
}
#define FS_CHK(site, idx, lim) fs_chk(site, (long)(idx), (long)(lim))
#else
#define FS_CHK(site, idx, lim) (idx)
#endif

template <typename R> struct PolyNode {
  const R *x, *z;
  int stride, n;
  R nl, nm, nr, liml, limr, curv, zmin;
  // Stage table (fs_abi.hip: build_stage_table; tz == nullptr: none, walk the edges).  Two parts per channel:
  //   breakpoints  [N][KP]: the node's distinct vertex elevations, ascending, padded with +inf (KP a multiple of 16);
  //   intervals    [K][FS_PT_BLOCK / 2][N] 16-byte pairs: for interval k = (z_k, z_k+1] of a node FS_PT_NCOEF coefficients in
  //                u = stage - z_k, then what FS_PT_* lists.  NODE-MINOR: the lanes of a wave sit on consecutive nodes, and
  //                neighbouring nodes are wetted to the same interval nearly everywhere, so one load instruction of the
  //                wave touches a few cache lines.  (A first layout kept each node's intervals in one 256-byte block: a
  //                single fetch per lane, but every one of its 16 load instructions then touched 64 different lines - the
  //                texture addresser takes a cycle per line, TA_TA_BUSY 78 % of the kernel, profiles/round3.)
  const R *tz;             // the node's breakpoints
  const R *tco;            // the node's slot in the interval part (pair 0 of interval 0); cstride pairs from pair to pair
  int cstride;
  int K, KP;
};

// Stage table.  Between two consecutive vertex elevations the set of wet vertices is fixed, and what properties() /
// get_equivalent_n() sum edge by edge (cross_section.py:248-328, :449-500) are low-order polynomials of the stage:
//   an edge wet at both ends adds   dx (s - zmid)            to A,  its length to P,  dx to T;
//   a water's-edge edge adds        (dx / 2|dz|) (s - zw)^2  to A,  (len / |dz|) (s - zw) to P,  (dx / |dz|) (s - zw) to T
// (zw: elevation of its wet end).  Expanded in u = s - tz[k] >= 0 every coefficient is a sum of non-negative terms - no
// cancellation, so the 1e-6 finite differences of dR_dA / dA_dh (:523-538) survive.  Per interval: whole section A (3), P (2),
// T (2), then A (3) and P (2) of the left, main and right roughness strips (an edge belongs to a strip by its two stations, :459).
// An interval's block then carries what an evaluation inside it needs and nothing else has to be fetched: its own bounds (the
// next evaluation of the node starts from the interval of the last one, node_terms_poly_hinted), the node's three Manning
// values, its curvature and z_min.  32 doubles = two 128-byte lines, fetched with sixteen 16-byte loads off one address.
enum { FS_PT_A0 = 0, FS_PT_A1, FS_PT_A2, FS_PT_P0, FS_PT_P1, FS_PT_T0, FS_PT_T1, FS_PT_STRIP = 7, FS_PT_NCOEF = 22, FS_PT_NSUB = 22,
       FS_PT_ZLO = 23, FS_PT_ZHI = 24, FS_PT_NL = 25, FS_PT_NM = 26, FS_PT_NR = 27, FS_PT_CURV = 28, FS_PT_ZMIN = 29, FS_PT_USED = 30,
       FS_PT_BLOCK = 32 };
// doubles of one node's stage table for polylines of up to P vertices
// (the breakpoints padded with +inf to a multiple of 16: the scan fetches them 16 at a time, eight 16-byte loads in flight)
__host__ __device__ constexpr int poly_table_bp(int P) { return (P + 16) & ~15; }
__host__ __device__ constexpr int poly_table_stride(int P) { return poly_table_bp(P) + P * FS_PT_BLOCK; }   // per node, both parts

// vertices [lo, hi] of a node, optionally extended by a water's-edge point at elevation zc on
// either side (the x_seg / z_seg of cross_section.py:352-365)
template <typename R> struct PolyView { int lo, hi; bool vl, vr; R xl, xr, zc; };

template <typename R> struct PolyEval { R A, P, Rh, T, neq, K, dRdA, dKdA, dAdh, y13; };
// (K, dK/dA, dA/dh of the WHOLE section ride along: a normal-depth boundary at the node is a row in these three,
// boundary.py:80, :161-181, and the kernel takes it from the fold's own evaluation of the node instead of evaluating twice)
template <typename R> struct PolyBC { R K, dKdA, dAdh; };

// sqrt that tolerates 0 (a vertical or degenerate edge): x / sqrt(x) via the reciprocal square root
template <typename R> __device__ __forceinline__ R fsqrt_len(R x) { return x > R(0) ? fsqrt_pos(x) : R(0); }

// contribution of the edge (x0,z0)-(x1,z1) below stage hw; cross_section.py:286-322 edge by edge.
// dx, dz, len: the edge's extents and full length (an edge that is wet at both ends contributes its
// whole length whatever the stage, so the square root is shared by the three stages of poly_eval).
template <typename R>
__device__ __forceinline__ void poly_edge(R x0, R z0, R x1, R z1, R dx, R dz, R len, R hw, R &A, R &P, R &T) {
  const R d0 = hw - z0, d1 = hw - z1;
  const bool w0 = d0 > R(0), w1 = d1 > R(0);
  if (w0 && w1) {
    A += R(0.5) * (d0 + d1) * dx;
    P += len;
    T += dx;
  } else if (w1 && z0 > hw) {                 // left water's edge, :289-296
    const R t = (hw - z0) * frcp(dz);
    const R xl = x0 + t * dx;
    const R cx = x1 - xl, cz = z1 - hw;
    A += R(0.5) * d1 * cx;
    P += fsqrt_len(cx * cx + cz * cz);
    T += cx;
  } else if (w0 && z1 > hw) {                 // right water's edge, :298-305
    const R t = (hw - z0) * frcp(dz);
    const R xr = x0 + t * dx;
    const R cx = xr - x0, cz = hw - z0;
    A += R(0.5) * d0 * cx;
    P += fsqrt_len(cx * cx + cz * cz);
    T += cx;
  }
}

template <typename R> __device__ __forceinline__ R strip_K(R A, R P, R n) {   // :457-481
  return (A <= R(0) || P <= R(0)) ? R(0) : conv_(A, n, A * frcp(P));
}

// what one pass over the edges of a (sub-)section accumulates: (A, P) at the stages hw, hw - dh, hw + dh, the top width at hw
// and (A, P) of the three roughness strips at hw
template <typename R> struct PolySums { R A0, P0, T0, A1, P1, A2, P2, Al, Pl, Am, Pm, Ar, Pr; };

template <typename R>
__device__ FS_POLY_ATTR PolySums<R> poly_sums_walk(const PolyNode<R> nd, const PolyView<R> v, R hw) {
  const R dh = R(1e-6);
  const int j0 = v.lo - (v.vl ? 1 : 0), j1 = v.hi + (v.vr ? 1 : 0);
  typedef const __attribute__((address_space(1))) R *GlobalR;          // (device memory: global loads, not flat ones)
  const GlobalR gx = (GlobalR)nd.x, gz = (GlobalR)nd.z;
  auto X = [&](int j) { return j < v.lo ? v.xl : (j > v.hi ? v.xr : gx[(size_t)FS_CHK(1, j, nd.n) * nd.stride]); };
  auto Z = [&](int j) { return (j < v.lo || j > v.hi) ? v.zc : gz[(size_t)FS_CHK(2, j, nd.n) * nd.stride]; };
  const R xa = X(j0), xb = X(j1);             // self.x[0], self.x[-1] of this (sub-)section
  R A0 = 0, P0 = 0, T0 = 0, A1 = 0, P1 = 0, A2 = 0, P2 = 0, Td = 0;
  R Al = 0, Pl = 0, Am = 0, Pm = 0, Ar = 0, Pr = 0;
  R x0 = xa, z0 = Z(j0);
#if FS_POLY_PREFETCH
  // the next vertex is requested one edge ahead: an edge is ~300 instructions, a vertex load an L2 round trip
  R xn = X(min(j0 + 1, j1)), zn = Z(min(j0 + 1, j1));
#endif
  for (int j = j0; j < j1; ++j) {
#if FS_POLY_PREFETCH
    const R x1 = xn, z1 = zn;
    xn = X(min(j + 2, j1)); zn = Z(min(j + 2, j1));
#else
    const R x1 = X(j + 1), z1 = Z(j + 1);
#endif
    const R dx = x1 - x0, dz = z1 - z0;
    // full length: needed as soon as both ends are wet at the highest of the three stages
    const R len = (hw + dh > z0 && hw + dh > z1) ? fsqrt_len(dx * dx + dz * dz) : R(0);
    R eA = 0, eP = 0, eT = 0;
    poly_edge(x0, z0, x1, z1, dx, dz, len, hw, eA, eP, eT);
    A0 += eA; P0 += eP; T0 += eT;
    poly_edge(x0, z0, x1, z1, dx, dz, len, hw - dh, A1, P1, Td);
    poly_edge(x0, z0, x1, z1, dx, dz, len, hw + dh, A2, P2, Td);
    // roughness strips: an edge belongs to a strip when both of its stations pass the strip's mask (:459)
    if (x0 >= xa && x1 <= nd.liml) { Al += eA; Pl += eP; }
    if (x0 >= nd.liml && x1 <= nd.limr) { Am += eA; Pm += eP; }
    if (x0 >= nd.limr && x1 <= xb) { Ar += eA; Pr += eP; }
    x0 = x1; z0 = z1;
  }
  PolySums<R> q;
  q.A0 = A0; q.P0 = P0; q.T0 = T0; q.A1 = A1; q.P1 = P1; q.A2 = A2; q.P2 = P2;
  q.Al = Al; q.Pl = Pl; q.Am = Am; q.Pm = Pm; q.Ar = Ar; q.Pr = Pr;
  return q;
}

// properties / get_equivalent_n / conveyance / dR_dA / dK_dA / dA_dh from the sums of a (sub-)section
template <typename R>
__device__ __forceinline__ PolyEval<R> poly_finish(const PolyNode<R> &nd, const PolySums<R> &q) {
  const R A0 = q.A0, P0 = q.P0, A1 = q.A1, P1 = q.P1, A2 = q.A2, P2 = q.P2;
  PolyEval<R> e;
  e.A = A0; e.P = P0; e.T = q.T0;
  e.Rh = P0 > R(0) ? A0 * frcp(P0) : R(0);
  e.y13 = e.Rh > R(0) ? rcbrt_pos(e.Rh) : R(0);
  const R R23 = e.Rh * e.y13;
  e.neq = nd.nm;                                                        // :486-487 fallback
  if (A0 > R(0) && P0 > R(0)) {
    const R Kt = p23_(p32_(strip_K(q.Al, q.Pl, nd.nl)) + p32_(strip_K(q.Am, q.Pm, nd.nm)) + p32_(strip_K(q.Ar, q.Pr, nd.nr)));
    if (Kt > R(0)) e.neq = A0 * R23 * frcp(Kt);                         // :492-498
  }
  const R rneq = frcp(e.neq);
  e.K = A0 <= R(0) ? R(0) : A0 * R23 * rneq;                            // :505-513
  // the two radii differ by ~1e-6 of themselves: these quotients stay IEEE divisions
  const R R1 = P1 > R(0) ? A1 / P1 : R(0), R2 = P2 > R(0) ? A2 / P2 : R(0);
  e.dRdA = (A2 - A1) == R(0) ? R(0) : (R2 - R1) / (A2 - A1);            // :523-531
  e.dKdA = A0 <= R(0) ? R(0) : (R23 + A0 * R(2.0 / 3.0) * e.y13 * e.dRdA) * rneq;    // :515-521
  e.dAdh = (A2 - A1) * R(5.0e5);                                        // / (2 dh), :533-538
  return e;
}

template <typename R>
__device__ FS_POLY_ATTR PolyEval<R> poly_eval(const PolyNode<R> nd, const PolyView<R> v, R hw) {
  return poly_finish(nd, poly_sums_walk(nd, v, hw));
}

template <typename R> __device__ __forceinline__ PolyView<R> poly_whole(const PolyNode<R> &nd) {
  PolyView<R> v;
  v.lo = 0; v.hi = nd.n - 1; v.vl = false; v.vr = false; v.xl = R(0); v.xr = R(0); v.zc = R(0);
  return v;
}

// The whole section at stage hw from the node's stage table: one scan over the breakpoints places the three stages, then
// ~25 coefficient loads and as many fmas stand in for the walk over all edges.  *nsub returns the number of wetted runs of
// >= 2 vertices (1 when there is no table: the caller then counts them itself).  A stage that coincides with a vertex
// elevation to the last bit goes back to the edge walk (there the reference drops the two edges at that vertex, :262).
// The edge walk over the whole section: with stage tables it only runs when a stage hits a vertex elevation to within 1e-6 (or
// when the tables are switched off, FS_POLY_WALK=1).  Inline like everything else (fs_kernel.hpp, Geometry<R, FS_SEC_IRREGULAR>: a
// call here was what the compiler's interprocedural register allocation tripped over).
template <typename R>
__device__ __forceinline__ PolyEval<R> poly_eval_whole_walk(const PolyNode<R> nd, R hw) {
  return poly_finish(nd, poly_sums_walk(nd, poly_whole(nd), hw));
}

// (A, P) at the three stages, T and the strips' (A, P) at hw from the coefficients of the interval that holds them (zk: its
// lower breakpoint)
template <typename R>
__device__ __forceinline__ PolySums<R> poly_sums_table(const R (&co)[FS_PT_USED], R zk, R hw) {
  const R dh = R(1e-6);
  const R u = hw - zk, u1 = (hw - dh) - zk, u2 = (hw + dh) - zk;
  PolySums<R> q;
  q.A0 = fma_(fma_(co[FS_PT_A2], u, co[FS_PT_A1]), u, co[FS_PT_A0]);
  q.P0 = fma_(co[FS_PT_P1], u, co[FS_PT_P0]);
  q.T0 = fma_(co[FS_PT_T1], u, co[FS_PT_T0]);
  q.A1 = fma_(fma_(co[FS_PT_A2], u1, co[FS_PT_A1]), u1, co[FS_PT_A0]); q.P1 = fma_(co[FS_PT_P1], u1, co[FS_PT_P0]);
  q.A2 = fma_(fma_(co[FS_PT_A2], u2, co[FS_PT_A1]), u2, co[FS_PT_A0]); q.P2 = fma_(co[FS_PT_P1], u2, co[FS_PT_P0]);
  auto strip = [&](int sidx, R &A, R &P) {
    const int o = FS_PT_STRIP + 5 * sidx;
    A = fma_(fma_(co[o + 2], u, co[o + 1]), u, co[o]);
    P = fma_(co[o + 4], u, co[o + 3]);
  };
  strip(0, q.Al, q.Pl); strip(1, q.Am, q.Pm); strip(2, q.Ar, q.Pr);
  return q;
}

// interval k of a node's stage table (tco: the node's slot in the interval part, cstride: 16-byte pairs between two pairs of it)
template <typename R>
__device__ __forceinline__ void poly_load_block(const R *tco, int cstride, int k, R (&co)[FS_PT_USED]) {
  typedef R R2 __attribute__((ext_vector_type(2)));
  typedef const __attribute__((address_space(1))) R2 *GlobalR2;
  const GlobalR2 cb = (GlobalR2)tco + (size_t)FS_CHK(3, k, 4096) * (FS_PT_BLOCK / 2) * cstride;
#pragma unroll
  for (int i = 0; i < FS_PT_USED / 2; ++i) { const R2 v = cb[(size_t)i * cstride]; co[2 * i] = v.x; co[2 * i + 1] = v.y; }
}

// *kout (optional): the interval the stage was found in, -1 when the evaluation went back to the edge walk
template <typename R>
__device__ FS_POLY_ATTR PolyEval<R> poly_eval_whole(const PolyNode<R> nd, R hw, int *nsub, int *kout = nullptr) {
  *nsub = -1;                                   // unknown: the caller counts the runs
  if (kout) *kout = -1;
  if (nd.tz == nullptr) return poly_eval_whole_walk(nd, hw);
  typedef R R2 __attribute__((ext_vector_type(2)));
  // the tables live in device memory: say so (the pointer came through a struct and a call boundary, the compiler no longer
  // knows, and a flat load waits on the LDS counter as well as on the memory one)
  typedef const __attribute__((address_space(1))) R2 *GlobalR2;
  const R dh = R(1e-6);
  const R s1 = hw - dh, s2 = hw + dh;
  // One scan places all three stages: c1 = breakpoints below hw - dh, c2 = breakpoints at or below hw + dh.  Equal counts: the
  // three stages lie strictly inside one interval.  Otherwise a vertex elevation lies within 1e-6 of the stage - or coincides
  // with a stage to the last bit, where the reference drops the two edges at that vertex (:262) - and the evaluation goes back
  // to the edge walk (a chance of ~1e-5 per evaluation).
  int c1 = 0, c2 = 0;
  const GlobalR2 bp = (GlobalR2)nd.tz;
  for (int j0 = 0; j0 < nd.KP / 2; j0 += 8) {          // 16 breakpoints per round: all eight loads issued before the first compare
    R2 z[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = bp[FS_CHK(4, j0 + i, nd.KP / 2)];
#pragma unroll
    for (int i = 0; i < 8; ++i) { c1 += (z[i].x < s1) + (z[i].y < s1); c2 += (z[i].x <= s2) + (z[i].y <= s2); }
  }
  if (c1 != c2) return poly_eval_whole_walk(nd, hw);
  PolySums<R> q;
  q.A0 = q.P0 = q.T0 = q.A1 = q.P1 = q.A2 = q.P2 = q.Al = q.Pl = q.Am = q.Pm = q.Ar = q.Pr = R(0);
  *nsub = 0;
  if (c1 > 0) {
    R co[FS_PT_USED];
    poly_load_block(nd.tco, nd.cstride, (int)FS_CHK(5, c1 - 1, nd.K), co);
    q = poly_sums_table(co, co[FS_PT_ZLO], hw);
    *nsub = (int)co[FS_PT_NSUB];
    if (kout) *kout = c1 - 1;
  }
  return poly_finish(nd, q);
}

// Se, dSe/dA, dSe/dQ, A, dA/dh of a polyline node: friction_slope / dSf_dA / dSf_dQ of
// cross_section.py:372-447 (sub-channel sum when >= 2 wetted runs) plus the base-class curvature terms.
// the node terms from the section's properties and its conveyance (K, dK/dA: the whole section's or the sub-channel sums)
template <typename R>
__device__ __forceinline__ NodeTerms<R> poly_terms_tail(const PolyEval<R> &e, R K, R dK, R curv, R h, R Q) {
  NodeTerms<R> t;
  const R rK = frcp(K), iK2 = rK * rK;
  const R aQ = fabs_(Q);
  const R Sf = Q * aQ * iK2;
  R dSeA = R(-2) * Sf * (dK * rK);
  R Se = Sf, eQ = R(2) * aQ * iK2;
  add_curvature(curv, e.A, frcp(e.A), e.T, frcp(e.T), e.dAdh, e.neq, e.y13, e.dRdA, h, Q, Se, dSeA, eQ);
  t.A = e.A; t.T = e.dAdh; t.Se = Se; t.eAT = dSeA; t.eQ = eQ; t.v = Q * frcp(e.A);
  t.rT = frcp(e.dAdh);
  return t;
}

// *kout (optional): the table interval the evaluation used when a later one may start from it (one wetted run, no edge walk),
// else -1
template <typename R>
__device__ FS_POLY_ATTR NodeTerms<R> node_terms_poly(const PolyNode<R> nd, R h, R Q, int *kout = nullptr, PolyBC<R> *bc = nullptr) {
  const R hw = h + nd.zmin;
  int nsub = 0;
  const PolyEval<R> e = poly_eval_whole(nd, hw, &nsub, kout);
  if (kout && nsub >= 2) *kout = -1;
  if (bc) { bc->K = e.K; bc->dKdA = e.dKdA; bc->dAdh = e.dAdh; }
  R K = e.K, dK = e.dKdA;
  // wetted runs of >= 2 vertices (get_subchannels, :330-370): from the stage table, else counted here
  typedef const __attribute__((address_space(1))) R *GlobalR;
  const GlobalR gx = (GlobalR)nd.x, gz = (GlobalR)nd.z;
  if (nsub < 0) {
    nsub = 0;
    int run = 0;
    for (int j = 0; j < nd.n; ++j) {
      const bool wet = gz[(size_t)FS_CHK(6, j, nd.n) * nd.stride] < hw;
      if (wet) ++run;
      if (!wet || j == nd.n - 1) { nsub += run >= 2; run = 0; }
    }
  }
  if (nsub >= 2) {
    R Ks = 0, dKs = 0;
    int s = -1;
    for (int j = 0; j <= nd.n; ++j) {
      const bool wet = j < nd.n && gz[(size_t)FS_CHK(7, j, nd.n) * nd.stride] < hw;
      if (wet && s < 0) s = j;
      if (!wet && s >= 0) {
        const int en = j;                                   // one past the last wet vertex
        if (en - s >= 2) {
          PolyView<R> v;
          v.lo = s; v.hi = en - 1; v.zc = hw;
          v.vl = s > 0 && gz[(size_t)FS_CHK(8, s - 1, nd.n) * nd.stride] > hw;
          v.xl = gx[(size_t)FS_CHK(9, s, nd.n) * nd.stride];               // :357 (np.interp, decreasing abscissa)
          v.vr = false; v.xr = R(0);
          if (en < nd.n) {
            const R za = gz[(size_t)FS_CHK(10, en - 1, nd.n) * nd.stride], zb = gz[(size_t)FS_CHK(11, en, nd.n) * nd.stride];
            if (za < hw && zb > hw) {                       // :360-363
              const R xa_ = gx[(size_t)FS_CHK(12, en - 1, nd.n) * nd.stride], xb_ = gx[(size_t)FS_CHK(13, en, nd.n) * nd.stride];
              v.vr = true;
              v.xr = (xb_ - xa_) / (zb - za) * (hw - za) + xa_;
            }
          }
          const PolyEval<R> sub = poly_eval(nd, v, hw);
          Ks += p32_(sub.K);                                // :389-390
          dKs += R(1.5) * fsqrt_len(sub.K) * sub.dKdA;      // :412-413
        }
        s = -1;
      }
    }
    K = p23_(Ks);                                           // :392, :415
    dK = R(2.0 / 3.0) * rcbrt_pos(Ks) * dKs;                // :416
  }
  return poly_terms_tail(e, K, dK, nd.curv, h, Q);
}

// The same from the interval of the node's last evaluation.  A Newton iterate moves the stage by far less than the distance
// between two vertex elevations, so nearly every evaluation finds its stage in the interval the last one used: ONE fetch of
// that interval's entry (it brings the interval's bounds and the node's constants with it, see FS_PT_*), no breakpoint scan, no
// per-node parameter loads - one memory round trip where the scan path has three in a row (node parameters, breakpoints,
// coefficients).  Same coefficients, same arithmetic: the result is bitwise that of node_terms_poly.  `slow` is called (with
// nothing fetched) when there is no interval to start from, when the stage has left it, or when the interval has two or more
// wetted runs (sub-channels: node_terms_poly walks them); it returns the terms and the next hint.
template <typename R> struct TermsHint { NodeTerms<R> t; PolyBC<R> bc; int k; };

template <typename R, typename Slow>
__device__ __forceinline__ NodeTerms<R> node_terms_poly_hinted(const R *tco, int cstride, bool has_over, R n_over, int &kh, R h, R Q,
                                                               PolyBC<R> &bc, Slow slow) {
  if (kh >= 0) {
    R co[FS_PT_USED];
    poly_load_block(tco, cstride, kh, co);
    const R hw = h + co[FS_PT_ZMIN], dh = R(1e-6);
    if (co[FS_PT_ZLO] < hw - dh && co[FS_PT_ZHI] > hw + dh && co[FS_PT_NSUB] < R(2)) {
      PolyNode<R> nd;                                      // (poly_finish reads the three Manning values only)
      nd.nl = co[FS_PT_NL]; nd.nm = has_over ? n_over : co[FS_PT_NM]; nd.nr = co[FS_PT_NR];
      const PolyEval<R> e = poly_finish(nd, poly_sums_table(co, co[FS_PT_ZLO], hw));
      bc.K = e.K; bc.dKdA = e.dKdA; bc.dAdh = e.dAdh;
      return poly_terms_tail(e, e.K, e.dKdA, co[FS_PT_CURV], h, Q);
    }
  }
  const TermsHint<R> r = slow();
  kh = r.k; bc = r.bc;
  return r.t;
}

// normal-depth boundary row at a polyline node: conveyance at hw = z_min + h for the residual,
// dK/dA * dA/dh at hw = h + bed_level for the derivative (boundary.py:80, :91, :161, :180)
template <typename R> __device__ __forceinline__ BCRow<R> normal_depth_row_poly(R sg, R rt, R K, R dKdA, R dAdh, R Q) {
  BCRow<R> r;
  r.res = Q - sg * K * rt;
  r.dh = R(0) - sg * dKdA * rt * dAdh;
  r.dq = R(1);
  return r;
}

template <typename R>
__device__ FS_POLY_ATTR BCRow<R> bc_normal_depth_poly(const PolyNode<R> nd, R S0, R bed, R h, R Q) {
  const R sg = S0 < R(0) ? R(-1) : R(1);
  const R rt = sqrt_(fabs_(S0));
  int ns_;
  const PolyEval<R> gr = poly_eval_whole(nd, nd.zmin + h, &ns_);
  // boundary.py:165-181 takes the residual at the section's own datum and the derivative at the boundary's bed level; they
  // are usually the same stage, and a section walk here costs the reach's only wave as much as a node of the fold
  PolyEval<R> gd = gr;
  if (h + bed != nd.zmin + h) gd = poly_eval_whole(nd, h + bed, &ns_);
  return normal_depth_row_poly(sg, rt, gr.K, gd.dKdA, gd.dAdh, Q);
}

// area and geometric top width only (Solver.prepare_results, solver.py:65-127)
template <typename R>
__device__ __forceinline__ void poly_area_top(const PolyNode<R> &nd, R hw, R &A, R &T) {
  R P = 0;
  A = 0; T = 0;
  R x0 = nd.x[0], z0 = nd.z[0];
  for (int j = 1; j < nd.n; ++j) {
    const R x1 = nd.x[(size_t)j * nd.stride], z1 = nd.z[(size_t)j * nd.stride];
    const R dx = x1 - x0, dz = z1 - z0;
    poly_edge(x0, z0, x1, z1, dx, dz, R(0), hw, A, P, T);        // the perimeter is not used here
    x0 = x1; z0 = z1;
  }
}

}  // namespace fs
