// fs_kernel.hpp - the fused Preissmann time-step kernel for gfx950 (MI355X).
//
// One workgroup of W wavefronts advances ONE reach through n_steps time levels; the whole Newton
// loop of a level (residual, Jacobian, linear solve, norm, update - reference
// src/hydromodel/preissmann.py:101-163) runs on chip:
//
//   * lane t owns the M consecutive cells [t*M, (t+1)*M) and keeps their unknowns (h, Q) in
//     registers for the whole launch (M+1 nodes, the last one shared with lane t+1);
//   * the level-(k) halves of the 4-point Preissmann stencil (preissmann.py:899-910) are reduced
//     to 4 constants per cell, kept in LDS, written once per level, read once per iteration;
//   * the 2x2-block banded Jacobian is never materialised: each cell's 8 entries
//     (preissmann.py:407-733) are folded straight into the lane's running segment
//     (fs_device.hpp); the 64*W lane segments are reduced by a log-depth tree - DPP moves inside a
//     wave, one LDS mailbox and one barrier across waves - the two boundary rows close the system,
//     the separators come back down (every lane tracks both ends of its group and reads the group's
//     record from LDS: no cross-lane traffic) and each lane back-substitutes its chunk;
//   * HBM is read once per launch and written at its last level (every level only into an optional
//     history): the accepted iterate of level k (preissmann.py:166-177; SURVEY F2: the pre-update
//     iterate) seeds the level constants of k+1 straight from registers.  Boundary hydrographs and
//     iteration counts go to small per-level tables.
//
// Template parameters: R = float|double, SEC = FS_SEC_*, M = cells per lane, W = waves per reach,
// RAGGED = false promises N-1 in {64*W*M - 1, 64*W*M}: then only the very last cell of a lane can be
// padding and the per-cell padding selects (and their 64-bit lane masks) disappear;
// BCK (boundary-kind class the kernel is compiled for): -1 = any kinds, 0 = any but FS_BC_STORAGE_CURVE (the row
// evaluation switches at run time),
// 1 = RECT_UNIFORM with bc_is_light() kinds on both ends (closed-form rows, parameters in LDS), 2 + k = flow
// hydrograph upstream and kind k downstream, known at compile time: the switch over the kinds folds away (rows the
// reach never evaluates still cost registers and shape the code around them: flagship +1.5 %, C5 +2-3.5 %).
// fs_abi.hip picks the most specific instantiation that matches the batch.
#pragma once
#include <type_traits>
#include "fs_device.hpp"
#include "fs_poly.hpp"

// build-time experiment switches (defaults = the configuration that measured fastest)
#ifndef FS_CELL_FENCE
#define FS_CELL_FENCE 1
#endif
#ifndef FS_WPE_W1
#define FS_WPE_W1 1        // min waves/SIMD the one-wave-per-reach kernels are compiled for (2..4 measured: scratch spills, 0.25-0.8x)
#endif
#ifndef FS_SAVE_TERMS_MAXM_F64
#define FS_SAVE_TERMS_MAXM_F64 2   // 4 and 8 cells per lane measured: trapezoid -13 % / -3 %, table +1 % (the LDS traffic of every fold outweighs one pass per level)
#endif
#ifndef FS_SAVE_TERMS
#define FS_SAVE_TERMS 1
#endif
#ifndef FS_SHARE_NODE
#define FS_SHARE_NODE 1    // one-wave-per-reach kernels with general sections: a lane's last node is its right neighbour's first -
#endif                     // take the neighbour's node terms (12 DPP moves) instead of evaluating the node a second time
#ifndef FS_WPE_TRAP4
#define FS_WPE_TRAP4 2          // trapezoid kernels with 4 cells per lane: 348 registers capped at 256 (+29 % at N = 200; the table kernel of that shape, 406 registers, loses 20 % when capped)
#endif
#ifndef FS_WPE_RECT8
#define FS_WPE_RECT8 2          // rectangular fast-path kernels with <= 8 cells per lane: the ragged (8,1) one needs 310 registers, capped at 256 it runs two waves per SIMD (+21 % at N = 300)
#endif
#ifndef FS_WPE_LEAN_POLY
#define FS_WPE_LEAN_POLY 2      // the same for the polyline kernels (332 registers capped at 256: +79 % on the polyline ensemble)
#endif
#ifndef FS_WPE_LEAN_SHORT
#define FS_WPE_LEAN_SHORT 2     // fp64 table kernels of class 0 with <= 2 cells per lane: 314 registers capped at 256, two waves per SIMD (C4 +43 %)
#endif
#ifndef FS_WPE_PINNED_SHORT
#define FS_WPE_PINNED_SHORT 2   // the same for kernels with the boundary kinds fixed at compile time
#endif
#ifndef FS_WPE_W1_F32_UNIFORM
#define FS_WPE_W1_F32_UNIFORM 3   // fp32, uniform geometry, <= 8 cells per lane: 207 registers capped at 168, three waves per SIMD (C5 +14 %; four: -24 %)
#endif
#ifndef FS_WPE_W1_F32
#define FS_WPE_W1_F32 2    // the same for fp32: two waves per SIMD fit (<= 256 registers) and hide the tree's latency (C5 fp32 +18 %)
#endif
#ifndef FS_LEVEL_FENCE
#define FS_LEVEL_FENCE 0     // scheduling fence every k cells of the level-constant pass (0 = none)
#endif
#ifndef FS_PIN_SEG
#define FS_PIN_SEG 1     // keep each merge in its cell's scheduling region (+3 % at M >= 8)
#endif
#ifndef FS_PHASE_FENCE
#define FS_PHASE_FENCE 8   // bit 3: pin the back-substituted updates and fence them off from the acceptance block (+0.9 %, 450 instead of 508 registers); bits 0-2 (other phase boundaries): no gain
#endif
#ifndef FS_LAUNDER_BACK
#define FS_LAUNDER_BACK 1
#endif

namespace fs {

// In-kernel phase stamps for diagnostic builds only (never in the shipped library): cycle sums per
// phase of the Newton iteration, one row per wave, written once at the end of the launch.
#ifdef FS_STAMP
#define FS_T(i)                                                         \
  do {                                                                  \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();       \
    stamp_[i] += now_ - tprev_;                                         \
    tprev_ = now_;                                                      \
  } while (0)
#else
#define FS_T(i) do { } while (0)
#endif

template <typename R> __host__ __device__ constexpr R huge_norm() { return sizeof(R) == 8 ? R(1e300) : R(3.0e38); }

template <typename R> struct KernelArgs {
  int32_t B, N, n_steps, level0, max_iter;
  int32_t iter_budget;     // kernels of boundary class -1 only: > 0 = Newton iterations this launch may spend on a reach (fs_batch_iterate)
  int32_t *it_done;        // [B] with iter_budget: iterations already spent on the open level, -1 once the reach has closed it
  R theta, dt, dx, tol;
  R *hk, *Qk;              // [B][N] accepted state of the current level (in: level0, out: level0+n_steps)
  R *hg, *Qg;              // [B][N] Newton start vector for the next level
  const R *geo_uniform;    // RECT_UNIFORM: [FS_RU_NPARAM][B]
  const R *geo_table;      // TABLE: [FS_GEO_NPARAM][N]
  const R *n_override;     // TABLE: [B] or nullptr
  const R *poly_x, *poly_z;      // IRREGULAR: [P][N] polyline stations / elevations (vertex-major)
  const R *poly_lim;             // IRREGULAR: [2][N] roughness strip limits
  const int32_t *poly_n;         // IRREGULAR: [N] vertex counts (0 = trapezoid-family node of the table)
  BCDesc<R> us, ds;
  R *Yprev;                // [B] storage stage of the current level
  R *stage_hist;           // [levels][B] storage stage per level (boundary.py:126-131)
  R *hydro;                // [levels][4][B]
  int32_t *iters;          // [levels][B]
  int32_t *status;         // [B]
  R *hist_h, *hist_Q;      // [levels][B][N] or nullptr
  R *trace;                // [levels][FS_TRACE_CAP][B] residual norms or nullptr
  unsigned long long *dbg; // diagnostic builds (-DFS_STAMP): [B][16][12] cycle sums per phase, else nullptr
};

template <typename R, int SEC> struct Geometry;

// the descriptor with its kind pinned to what the instantiation was compiled for (BCK >= 2)
template <int BCK, int SIDE, typename R>
__device__ __forceinline__ BCDesc<R> pinned(const BCDesc<R> &bc) {
  BCDesc<R> d = bc;
  if (BCK >= 2) d.kind = SIDE == 0 ? (int)FS_BC_FLOW_HYDROGRAPH : BCK - 2;
  return d;
}

template <typename R> struct Geometry<R, FS_SEC_RECT_UNIFORM> {
  static constexpr bool kConstT = true;      // dA/dh = b everywhere: no per-node top width to keep
  R b, rb, n, z_us, z_ds, inv_nm1, dz;
  __device__ __forceinline__ void init(const KernelArgs<R> &a, int reach) {
    b = a.geo_uniform[(size_t)FS_RU_WIDTH * a.B + reach];
    n = a.geo_uniform[(size_t)FS_RU_MANNING * a.B + reach];
    z_us = a.geo_uniform[(size_t)FS_RU_Z_US * a.B + reach];
    z_ds = a.geo_uniform[(size_t)FS_RU_Z_DS * a.B + reach];
    rb = R(1) / b;
    inv_nm1 = R(1) / R(a.N - 1);
    dz = (z_ds - z_us) * inv_nm1;
  }
  __device__ __forceinline__ R bed_step(int) const { return dz; }   // bed(node+1) - bed(node)
  __device__ __forceinline__ R terms_T() const { return b; }
  // distance-weighted interpolation between the two end sections (cross_section.py:887-900)
  __device__ __forceinline__ R bed(int node) const {
    const R w2 = R(node) * inv_nm1;
    return z_us * (R(1) - w2) + z_ds * w2;
  }
  __device__ __forceinline__ NodeTerms<R> terms(int, R h, R Q) const { return node_terms_rect(b, rb, n, h, Q); }
  __device__ __forceinline__ SecParams<R> section(int node) const {
    SecParams<R> s;
    s.z = bed(node); s.b = b; s.m = R(0); s.nm = n; s.nl = n; s.nr = n; s.hbf = R(0);
    s.bl = R(0); s.br = R(0); s.mfp = R(0); s.curv = R(0); s.compound = false;
    return s;
  }
  template <int BCK, int SIDE>
  __device__ __forceinline__ BCRow<R> boundary(const BCDesc<R> &bc, int reach, int B, int level, int node, R h, R Q,
                                               R Qold, R dt, R Yprev, R *Ynew, int *flag) const {
    // closed-form rows, bc.params points at the LDS copy made in the kernel prologue (fixed-size kinds only); with the kinds
    // fixed at compile time the rows this reach never evaluates are not compiled in (the flagship kernel: +1.5 %)
    if (BCK == 1 || (BCK >= 2 && bc_is_light(BCK - 2)))
      return bc_eval_rect(pinned<BCK, SIDE>(bc), (LdsParams<R>)bc.params, level, b, n, node == 0 ? z_us : z_ds, h, Q, Qold, dt, Yprev, Ynew, flag);
    return bc_eval<BCK != 0>(pinned<BCK, SIDE>(bc), reach, B, level, section(node), h, Q, Qold, dt, Yprev, Ynew, flag);
  }
};

template <typename R> struct Geometry<R, FS_SEC_TRAP_UNIFORM> {
  static constexpr bool kConstT = false;
  R b, m, sm2, n, z_us, z_ds, inv_nm1, dz;
  __device__ __forceinline__ void init(const KernelArgs<R> &a, int reach) {
    b = a.geo_uniform[(size_t)FS_RU_WIDTH * a.B + reach];
    n = a.geo_uniform[(size_t)FS_RU_MANNING * a.B + reach];
    z_us = a.geo_uniform[(size_t)FS_RU_Z_US * a.B + reach];
    z_ds = a.geo_uniform[(size_t)FS_RU_Z_DS * a.B + reach];
    m = a.geo_uniform[(size_t)FS_TU_SIDE_SLOPE * a.B + reach];
    sm2 = R(2) * sqrt_(R(1) + m * m);
    inv_nm1 = R(1) / R(a.N - 1);
    dz = (z_ds - z_us) * inv_nm1;
  }
  __device__ __forceinline__ R bed_step(int) const { return dz; }
  __device__ __forceinline__ R terms_T() const { return R(0); }      // unused (kConstT == false)
  __device__ __forceinline__ R bed(int node) const {
    const R w2 = R(node) * inv_nm1;
    return z_us * (R(1) - w2) + z_ds * w2;
  }
  __device__ __forceinline__ NodeTerms<R> terms(int, R h, R Q) const { return node_terms_trap(b, m, sm2, n, h, Q); }
  __device__ __forceinline__ SecParams<R> section(int node) const {
    SecParams<R> s;
    s.z = bed(node); s.b = b; s.m = m; s.nm = n; s.nl = n; s.nr = n; s.hbf = R(0);
    s.bl = R(0); s.br = R(0); s.mfp = R(0); s.curv = R(0); s.compound = false;
    return s;
  }
  template <int BCK, int SIDE>
  __device__ __forceinline__ BCRow<R> boundary(const BCDesc<R> &bc, int reach, int B, int level, int node, R h, R Q,
                                               R Qold, R dt, R Yprev, R *Ynew, int *flag) const {
    return bc_eval<BCK != 0>(pinned<BCK, SIDE>(bc), reach, B, level, section(node), h, Q, Qold, dt, Yprev, Ynew, flag);
  }
};

template <typename R> struct Geometry<R, FS_SEC_TABLE> {
  static constexpr bool kConstT = false;
  const R *tab;
  int N;
  R n_over;
  bool has_over;
  __device__ __forceinline__ void init(const KernelArgs<R> &a, int reach) {
    tab = a.geo_table; N = a.N;
    has_over = a.n_override != nullptr;
    n_over = has_over ? a.n_override[reach] : R(0);
  }
  __device__ __forceinline__ R terms_T() const { return R(0); }      // unused (kConstT == false)
  __device__ __forceinline__ R bed_step(int node) const {
    return tab[(size_t)FS_GEO_Z_BED * N + min(node + 1, N - 1)] - tab[(size_t)FS_GEO_Z_BED * N + min(node, N - 1)];
  }
  __device__ __forceinline__ R bed(int node) const { return tab[(size_t)FS_GEO_Z_BED * N + node]; }
  __device__ __forceinline__ SecParams<R> section(int node) const {
    SecParams<R> s;
    auto g = [&](int row) { return tab[(size_t)row * N + node]; };
    s.z = g(FS_GEO_Z_BED); s.b = g(FS_GEO_B_MAIN); s.m = g(FS_GEO_M_MAIN);
    s.nm = has_over ? n_over : g(FS_GEO_N_MAIN);
    s.nl = g(FS_GEO_N_LEFT); s.nr = g(FS_GEO_N_RIGHT);
    s.compound = g(FS_GEO_IS_COMPOUND) > R(0.5);
    s.hbf = g(FS_GEO_H_BANKFULL); s.bl = g(FS_GEO_B_FP_LEFT); s.br = g(FS_GEO_B_FP_RIGHT);
    s.mfp = g(FS_GEO_M_FP); s.curv = g(FS_GEO_CURVATURE);
    return s;
  }
  __device__ __forceinline__ NodeTerms<R> terms(int node, R h, R Q) const {
    return node_terms_general(section(node), h, Q);
  }
  template <int BCK, int SIDE>
  __device__ __forceinline__ BCRow<R> boundary(const BCDesc<R> &bc, int reach, int B, int level, int node, R h, R Q,
                                               R Qold, R dt, R Yprev, R *Ynew, int *flag) const {
    return bc_eval<BCK != 0>(pinned<BCK, SIDE>(bc), reach, B, level, section(node), h, Q, Qold, dt, Yprev, Ynew, flag);
  }
};

// TABLE plus polyline nodes (IrregularSection): per node either a row of the trapezoid table or a
// polyline; both evaluations are inlined (FS_POLY_INLINE: out of line they spill the caller around every call).
template <typename R>
__device__ FS_POLY_ATTR NodeTerms<R> node_terms_general_call(const SecParams<R> s, R h, R Q) { return node_terms_general(s, h, Q); }

template <typename R> struct Geometry<R, FS_SEC_IRREGULAR> {
  static constexpr bool kConstT = false;
  Geometry<R, FS_SEC_TABLE> tb;
  const R *px, *pz, *plim;
  const int32_t *pn;
  __device__ __forceinline__ void init(const KernelArgs<R> &a, int reach) {
    tb.init(a, reach);
    px = a.poly_x; pz = a.poly_z; plim = a.poly_lim; pn = a.poly_n;
  }
  __device__ __forceinline__ R terms_T() const { return R(0); }
  __device__ __forceinline__ R bed_step(int node) const { return tb.bed_step(node); }
  __device__ __forceinline__ R bed(int node) const { return tb.bed(node); }
  __device__ __forceinline__ SecParams<R> section(int node) const { return tb.section(node); }
  __device__ __forceinline__ PolyNode<R> poly(int node) const {
    PolyNode<R> p;
    const int N = tb.N;
    auto g = [&](int row) { return tb.tab[(size_t)row * N + node]; };
    p.x = px + node; p.z = pz + node; p.stride = N; p.n = pn[node];
    p.nl = g(FS_GEO_N_LEFT); p.nm = tb.has_over ? tb.n_over : g(FS_GEO_N_MAIN); p.nr = g(FS_GEO_N_RIGHT);
    p.liml = plim[node]; p.limr = plim[N + node];
    p.curv = g(FS_GEO_CURVATURE); p.zmin = g(FS_GEO_Z_BED);
    return p;
  }
  __device__ __forceinline__ NodeTerms<R> terms(int node, R h, R Q) const {
    if (pn[node] > 0) return node_terms_poly(poly(node), h, Q);
    return node_terms_general_call(section(node), h, Q);
  }
  template <int BCK, int SIDE>
  __device__ __forceinline__ BCRow<R> boundary(const BCDesc<R> &bc_, int reach, int B, int level, int node, R h, R Q,
                                               R Qold, R dt, R Yprev, R *Ynew, int *flag) const {
    const BCDesc<R> bc = pinned<BCK, SIDE>(bc_);
    if (pn[node] > 0 && bc.kind == FS_BC_NORMAL_DEPTH)
      return bc_normal_depth_poly(poly(node), bc_param(bc, 0, reach, B), bc_param(bc, 1, reach, B), h, Q);
    if (BCK != 0 && pn[node] > 0 && bc.kind == FS_BC_STORAGE_CURVE) {
      const PolyNode<R> nd = poly(node);
      const PolyEval<R> er = poly_eval(nd, poly_whole(nd), nd.zmin + h);
      const PolyEval<R> ed = poly_eval(nd, poly_whole(nd), h + bc_param(bc, FS_SC_BED_LEVEL, reach, B));
      EntryProps<R> pr{er.A, er.Rh, er.neq, er.dRdA, er.dAdh}, pd{ed.A, ed.Rh, ed.neq, ed.dRdA, ed.dAdh};
      return bc_storage_curve(bc, reach, B, level, pr, pd, h, Q, Qold, dt, Yprev, Ynew, flag);
    }
    return bc_eval<BCK != 0>(bc, reach, B, level, section(node), h, Q, Qold, dt, Yprev, Ynew, flag);
  }
};

// LDS carve-up for one reach
// W == 8 (512 threads, N up to 4097 at M = 8): the level-0 records of the in-wave tree stay in
// registers (20 per lane) so that the LDS slots of levels 1..5 (31 per wave) fit next to kc.
template <int W> struct TreeCfg { static constexpr bool kL0Regs = (W >= 8); static constexpr int kSlots = kL0Regs ? 32 : 64; };

// (A, Se, Q/A) of the nodes as the last fold saw them (kSaveTerms); an empty base otherwise
template <typename R, int M, int T, bool SAVE> struct SavedTerms { R nt[3][M + 1][T]; };
template <typename R, int M, int T> struct SavedTerms<R, M, T, false> {};

template <typename R, int M, int W, bool SAVE> struct Smem : SavedTerms<R, M, 64 * W, SAVE> {
  static constexpr int T = 64 * W;
  R kc[4][M][T];           // per-cell level-k constants, lane-minor (conflict-free ds_read_b64)
  R tree[W][10][TreeCfg<W>::kSlots];   // per-wave spill slots of the in-wave tree
  R xseg[2][W][10];        // wave segments, double-buffered by iteration parity
  R xbc[2][8];             // boundary rows: U(dh,dq,res) D(dh,dq,res)
  R bcp[2][FS_BC_MAX_PARAMS];   // this reach's boundary parameters (fixed-size kinds), read every Newton iteration
  R xnorm[2][W];
  int32_t xflag[2];
};

// What back-substitution needs for an interior node j of a lane's chunk: the M-like row of the
// running segment [first node .. j] (sm, pm, qm), pre-scaled by 1/det of the pivot block.  The
// continuity row of cell j that completes the block (T_j/(2dt), -+theta/dx and its residual) is
// recomputed from the still un-updated state when the top width is constant (kConstT), else its
// residual is kept in qc.
template <typename R> struct LocalElim { Parked<R> rs0, rs1, rk, rq, qc; };   // lives in AGPRs

// Minimum number of waves per SIMD a kernel is compiled for: caps its registers at 512 / n.  One wave per SIMD cannot
// hide the latency of the in-wave tree, a second one is worth 20-80 % wherever the kernel fits 256 registers or nearly
// does; forcing it on the larger kernels sends them to scratch (measured 0.25-0.8x).
template <typename R, int SEC, int M, int W, int BCK> constexpr int min_waves() {
  if (W > 1) return 1;          // multi-wave table kernels with 2 cells per lane at two waves per SIMD: no better than the 4- and 8-cell ones
  if (sizeof(R) == 4) return (M <= 8 && (SEC == FS_SEC_RECT_UNIFORM || SEC == FS_SEC_TRAP_UNIFORM)) ? FS_WPE_W1_F32_UNIFORM : FS_WPE_W1_F32;
  if (SEC == FS_SEC_RECT_UNIFORM && BCK >= 1 && M <= 8) return FS_WPE_RECT8;
  if (BCK >= 2 && M <= 2) return FS_WPE_PINNED_SHORT;
  if (BCK == 0 && M <= 2 && SEC == FS_SEC_TABLE) return FS_WPE_LEAN_SHORT;
  if (BCK == 0 && M == 4 && SEC == FS_SEC_TRAP_UNIFORM) return FS_WPE_TRAP4;
  if (BCK == 0 && M <= 2 && SEC == FS_SEC_IRREGULAR) return FS_WPE_LEAN_POLY;
  return FS_WPE_W1;
}

// DIAG = false: no per-level history and no residual trace (batches created without FS_FLAG_HISTORY / FS_FLAG_TRACE): the
// stores are never executed there, but compiled in they cost the flagship kernel 1.1 %
template <typename R, int SEC, int M, int W, bool RAGGED = true, int BCK = 0, bool DIAG = true>
__global__ __launch_bounds__(64 * W, (min_waves<R, SEC, M, W, BCK>())) void preissmann_step_kernel(const KernelArgs<R> a) {
  constexpr int T = 64 * W;
  using Geo = Geometry<R, SEC>;
  // Short general-section kernels keep (A, Se, Q/A) of every node of the current fold in LDS: if the iterate is accepted they
  // are the node terms of level k, and the level constants of level k+1 come from them instead of from another pass over the
  // sections (1 of 6 section passes of the polyline ensemble, 1 of 11 of C4)
  constexpr bool kSaveTerms = FS_SAVE_TERMS && !Geometry<R, SEC>::kConstT && (M <= 2 || (sizeof(R) == 8 && M <= FS_SAVE_TERMS_MAXM_F64));
  __shared__ Smem<R, M, W, kSaveTerms> sm;

  const int reach = blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // One Newton iteration per launch for batches with host-evaluated boundary rows (FS_BC_HOST_ROW, fs_batch_iterate): the
  // kernels of boundary class -1 carry an iteration budget and the count of the open level across launches.
  constexpr bool kBudget = (BCK == -1);
  int it_entry = 0;
  if (kBudget && a.iter_budget > 0) {
    it_entry = a.it_done[reach];
    if (it_entry < 0) return;                 // this reach has closed the level: it waits for the others (whole workgroup)
  }
  const int N = a.N, NC = N - 1;
  const int s0 = t * M;                       // first node / cell of this lane
  const int tD = (NC - 1) / M;                // lane that owns the last real cell
  const int jD = NC - tD * M;                 // local index (1..M) of node N-1 in that lane
  constexpr int kJD0 = RAGGED ? 1 : (M > 1 ? M - 1 : 1);   // full chunks: jD is M-1 or M
  const size_t base = (size_t)reach * N;

  Geo geo;
  geo.init(a, reach);

  const R th = a.theta, dt = a.dt;
  R r2dt = R(1) / (R(2) * dt);
  R cq = th / a.dx;                           // theta/dx
  const R cqk = (R(1) - th) / a.dx;           // (1-theta)/dx
  R hth = R(0.5) * th;
  const R hthk = R(0.5) * (R(1) - th);
  R g = R(kG);

  // ---- unknowns of this lane: nodes s0 .. s0+M (clamped copies beyond the last node) ----
  R h[M + 1], Q[M + 1];
  R QoldD = R(0);                              // flow[k] at the last node (vol_in of the storage BC)

  // level-k constants of the 4-point stencil from the accepted state (h, Q) of level k:
  //   C = [sumA]/(2dt) + cq*dQ                 + kc0
  //   M = [sumQ]/(2dt) + cq*d(Q^2/A)           + kc1 + g*(hth*sumA + kc2)*(cq*dY + hth*sumSe + kc3)
  // Node s0 + M is lane + 1's node s0 (both lanes hold bitwise equal copies of its unknowns): with one wave per reach the
  // neighbour's terms arrive by a wave rotate.  Lane 63 receives lane 0's, which only a padding cell ever looks at
  // (and discards) unless the reach fills the wave exactly - then lane 63 evaluates its last node itself.
  constexpr bool kShareNode = FS_SHARE_NODE && W == 1 && !Geo::kConstT;
  auto last_node_terms = [&](const NodeTerms<R> &first, R hM, R QM) {
    auto rol = [](R v) { return dpp_mov<0x134>(v); };     // wave_rol:1
    NodeTerms<R> r;
    r.A = rol(first.A); r.T = rol(first.T); r.Se = rol(first.Se); r.eA = rol(first.eA); r.eQ = rol(first.eQ); r.v = rol(first.v);
    if (NC == 64 * M && lane == 63) r = geo.terms(N - 1, hM, QM);
    return r;
  };
  auto write_level_constants = [&](const R(&hh)[M + 1], const R(&QQ)[M + 1]) {
    NodeTerms<R> L = geo.terms(min(s0, N - 1), hh[0], QQ[0]);
    NodeTerms<R> Rlast;
    if (kShareNode) Rlast = last_node_terms(L, hh[M], QQ[M]);
#pragma unroll
    for (int c = 0; c < M; ++c) {
      const NodeTerms<R> Rn = (kShareNode && c == M - 1) ? Rlast : geo.terms(min(s0 + c + 1, N - 1), hh[c + 1], QQ[c + 1]);
      const R sumA = L.A + Rn.A;
      sm.kc[0][c][t] = -sumA * r2dt + cqk * (QQ[c + 1] - QQ[c]);
      sm.kc[1][c][t] = -(QQ[c + 1] + QQ[c]) * r2dt + cqk * (QQ[c + 1] * Rn.v - QQ[c] * L.v);
      sm.kc[2][c][t] = hthk * sumA;
      sm.kc[3][c][t] = cqk * (geo.bed_step(s0 + c) + (hh[c + 1] - hh[c])) + hthk * (L.Se + Rn.Se);
      L = Rn;
#if FS_LEVEL_FENCE
      if ((c % FS_LEVEL_FENCE) == FS_LEVEL_FENCE - 1) __builtin_amdgcn_sched_barrier(0);
#endif
    }
  };
  auto save_terms = [&](int j, const NodeTerms<R> &nt) {
    if constexpr (kSaveTerms) { sm.nt[0][j][t] = nt.A; sm.nt[1][j][t] = nt.Se; sm.nt[2][j][t] = nt.v; }
  };
  auto level_constants_from_saved = [&](const R(&hh)[M + 1], const R(&QQ)[M + 1]) {
    if constexpr (kSaveTerms) {
      R A0 = sm.nt[0][0][t], Se0 = sm.nt[1][0][t], v0 = sm.nt[2][0][t];
#pragma unroll
      for (int c = 0; c < M; ++c) {
        const R A1 = sm.nt[0][c + 1][t], Se1 = sm.nt[1][c + 1][t], v1 = sm.nt[2][c + 1][t];
        const R sumA = A0 + A1;
        sm.kc[0][c][t] = -sumA * r2dt + cqk * (QQ[c + 1] - QQ[c]);
        sm.kc[1][c][t] = -(QQ[c + 1] + QQ[c]) * r2dt + cqk * (QQ[c + 1] * v1 - QQ[c] * v0);
        sm.kc[2][c][t] = hthk * sumA;
        sm.kc[3][c][t] = cqk * (geo.bed_step(s0 + c) + (hh[c + 1] - hh[c])) + hthk * (Se0 + Se1);
        A0 = A1; Se0 = Se1; v0 = v1;
      }
    }
  };

#pragma unroll
  for (int j = 0; j <= M; ++j) {
    const int node = min(s0 + j, N - 1);
    h[j] = a.hg[base + node];  Q[j] = a.Qg[base + node];
  }
  // per-lane base pointers: every later access is base + immediate offset
  R *const hk_p = a.hk + base + s0, *const Qk_p = a.Qk + base + s0;
  R *const hg_p = a.hg + base + s0, *const Qg_p = a.Qg + base + s0;

  // Boundary descriptors of this reach: the parameters of the fixed-size kinds are copied to LDS once
  // (the boundary rows are evaluated every Newton iteration on the critical path of the first and the
  // last wave; a global load there costs more than the row itself), the hydrograph value of a level
  // is fetched when the level starts.
  BCDesc<R> usd = a.us, dsd = a.ds;
  if (t < 2 * FS_BC_MAX_PARAMS) {
    const int side = t / FS_BC_MAX_PARAMS, i = t - side * FS_BC_MAX_PARAMS;
    const BCDesc<R> &src = side ? a.ds : a.us;
    static constexpr int kCount[] = {0, 1, 1, 2, 4, 5, 10, 5};
    if (src.kind <= FS_BC_STORAGE && i < kCount[src.kind]) sm.bcp[side][i] = bc_param(src, i, reach, a.B);
    if (src.kind == FS_BC_NORMAL_DEPTH && i == 2) {        // derived: sign(S0) sqrt|S0| (hydraulics.py:4-13)
      const R S0 = bc_param(src, 0, reach, a.B);
      sm.bcp[side][2] = (S0 < R(0) ? R(-1) : R(1)) * sqrt_(fabs_(S0));
    }
  }
  if (usd.kind <= FS_BC_STORAGE) { usd.params = &sm.bcp[0][0]; usd.stride = 0; }
  if (dsd.kind <= FS_BC_STORAGE) { dsd.params = &sm.bcp[1][0]; dsd.stride = 0; }

  // a compile-time constant in the kernels compiled for a boundary pair
  const bool ds_storage = BCK >= 2 ? bc_is_storage(BCK - 2) : bc_is_storage(a.ds.kind);
  R Yprev = (ds_storage && t == tD) ? a.Yprev[reach] : R(0);
  int status = a.status[reach];
  int parity = 0;
  if (t == 0) { sm.xflag[0] = 0; sm.xflag[1] = 0; }
  __syncthreads();

  {   // accepted state of the entry level -> 4 constants per cell in LDS (later levels: from registers, below)
    R hk[M + 1], Qk[M + 1];
#pragma unroll
    for (int j = 0; j <= M; ++j) {
      const int node = min(s0 + j, N - 1);
      hk[j] = a.hk[base + node]; Qk[j] = a.Qk[base + node];
    }
    if (t == tD) {
#pragma unroll
      for (int j = kJD0; j <= M; ++j) if (j == jD) QoldD = Qk[j];
    }
    write_level_constants(hk, Qk);
  }
#ifdef FS_STAMP
  unsigned long long stamp_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev_ = __builtin_amdgcn_s_memtime();
#endif
  for (int step = 0; step < a.n_steps && status == FS_OK; ++step) {
    const int level = a.level0 + step + 1;
    if (usd.target) usd.tgt = usd.target[(size_t)level * a.B + reach];
    if (dsd.target) dsd.tgt = dsd.target[(size_t)level * a.B + reach];
    int it = kBudget ? it_entry : 0;
    int budget = (kBudget && a.iter_budget > 0) ? a.iter_budget : 0x7fffffff;
    bool converged = false;
    R Ynew = Yprev;
    while (!converged && status == FS_OK) {
      if (kBudget && budget-- <= 0) break;
      ++it;
      if (it - 1 >= a.max_iter) { status = FS_MAX_ITER; break; }     // preissmann.py:124-126
      parity ^= 1;
      FS_T(7);

      // opaque lane offset (an integer, so the accesses stay LDS ds_read, not flat): the 4*M level
      // constants are Newton-loop invariants and would otherwise be hoisted into registers
      int kco = t;
      asm volatile("" : "+v"(kco));
      // the same for the lane number the tree-slot addresses derive from: as loop invariants the 12 addresses
      // are hoisted and spilled to scratch, and every reload is a full memory round trip inside the tree
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const R *kcb = &sm.kc[0][0][0] + kco;

      // ================= 1. local assembly + fold (registers only) =================
      LocalElim<R> el[M > 1 ? M - 1 : 1];
      R Tn[Geo::kConstT ? 1 : M + 1];           // top widths, only when they vary
      Seg<R> seg;
      R pf0 = R(0), pf1 = R(0);                // M-like row of the lane's first cell (left part)
      R nrm2 = R(0);
      {
        NodeTerms<R> L = geo.terms(min(s0, N - 1), h[0], Q[0]);
        NodeTerms<R> Rlast;
        if (kShareNode) Rlast = last_node_terms(L, h[M], Q[M]);
        save_terms(0, L);
        if (!Geo::kConstT) Tn[0] = L.T;
        R kap = R(1);                      // the M-like row keeps the direction of cell 0's: pm = kap * (pf0, pf1)
#pragma unroll
        for (int c = 0; c < M; ++c) {
          const R k0 = kcb[(0 * M + c) * T], k1 = kcb[(1 * M + c) * T], k2 = kcb[(2 * M + c) * T], k3 = kcb[(3 * M + c) * T];
          const NodeTerms<R> Rn = (kShareNode && c == M - 1) ? Rlast : geo.terms(min(s0 + c + 1, N - 1), h[c + 1], Q[c + 1]);
          save_terms(c + 1, Rn);
          if (!Geo::kConstT) Tn[c + 1] = Rn.T;
          Seg<R> cell;
          {
            // identity padding d_{i+1} = d_i beyond the last cell
            const bool real = (RAGGED || c == M - 1) ? (s0 + c < NC) : true;
            const R sumA = L.A + Rn.A;
            const R Cres = sumA * r2dt + cq * (Q[c + 1] - Q[c]) + k0;                            // :220-249
            const R avgA = hth * sumA + k2;
            const R S = cq * (geo.bed_step(s0 + c) + (h[c + 1] - h[c])) + hth * (L.Se + Rn.Se) + k3;
            const R Mres = (Q[c + 1] + Q[c]) * r2dt + cq * (Q[c + 1] * Rn.v - Q[c] * L.v) + k1 +
                           g * avgA * S;                                                     // :251-301
            nrm2 += real ? Cres * Cres + Mres * Mres : R(0);
            const R gA = g * avgA, gS = g * hth * S;
            cell.pc0 = real ? L.T * r2dt : R(1);   cell.pc1 = real ? -cq : R(0);             // :431-447, :476-491
            cell.sc0 = real ? Rn.T * r2dt : R(-1); cell.sc1 = real ? cq : R(0);              // :407-422, :456-471
            cell.qc = real ? -Cres : R(0);
            cell.pm0 = real ? cq * L.v * L.v * L.T + gA * (hth * L.eA - cq) + gS * L.T : R(0);        // :558-612
            cell.pm1 = real ? r2dt - cq * R(2) * L.v + gA * hth * L.eQ : R(1);                         // :677-733
            cell.sm0 = real ? -cq * Rn.v * Rn.v * Rn.T + gA * (hth * Rn.eA + cq) + gS * Rn.T : R(0);  // :496-550
            cell.sm1 = real ? r2dt + cq * R(2) * Rn.v + gA * hth * Rn.eQ : R(-1);                      // :619-675
            cell.qm = real ? -Mres : R(0);
          }
          if (c == 0) {
            seg = cell;
            pf0 = cell.pm0; pf1 = cell.pm1;
          } else {
            // merge(seg, cell) keeping only what the local back-substitution reads.  The merged M-like row's
            // left part is always a multiple of cell 0's (o.pm = -ga * seg.pm): only the factor kap is carried
            // and recorded, which is one value per node less to park and one multiplication less per merge.
            const R det = pivot_det(seg.sm0, seg.sm1, cell.pc0, cell.pc1);
            const R r = frcp(det);
            LocalElim<R> &e = el[c - 1];
            const R w21 = r * seg.sm0, rs1 = r * seg.sm1;
            e.rs0.put(w21); e.rs1.put(rs1); e.rk.put(r * kap); e.rq.put(r * seg.qm);
            if (!Geo::kConstT) e.qc.put(cell.qc);
            const R w10 = cell.pc1 * r, w11 = -cell.pc0 * r, w20 = -rs1;
            const R al = seg.sc0 * w10 + seg.sc1 * w11, be = seg.sc0 * w20 + seg.sc1 * w21;
            const R ga = cell.pm0 * w10 + cell.pm1 * w11, ep = cell.pm0 * w20 + cell.pm1 * w21;
            const R ak = al * kap;
            Seg<R> o;
            o.pc0 = seg.pc0 - ak * pf0;     o.pc1 = seg.pc1 - ak * pf1;
            o.sc0 = -be * cell.sc0;         o.sc1 = -be * cell.sc1;
            o.qc = seg.qc - al * seg.qm - be * cell.qc;
            kap = -ga * kap;
            o.pm0 = R(0);                   o.pm1 = R(0);          // materialised after the last cell
            o.sm0 = cell.sm0 - ep * cell.sc0; o.sm1 = cell.sm1 - ep * cell.sc1;
            o.qm = cell.qm - ga * seg.qm - ep * cell.qc;
            seg = o;
          }
#if FS_PIN_SEG
          // the running segment must exist here: keeps cell c's merge inside cell c's scheduling region
          // (otherwise the 15 merges sink below the last fence and every cell's coefficients are parked)
          // (long chunks only: with M <= 4 the cells' node terms interleave profitably, measured on C4)
          if (M >= 8)
            asm volatile("" :: "v"(seg.pc0), "v"(seg.pc1), "v"(seg.sc0), "v"(seg.sc1), "v"(seg.qc),
                               "v"(kap), "v"(seg.sm0), "v"(seg.sm1), "v"(seg.qm));
#endif
          L = Rn;
#if FS_CELL_FENCE
          if ((c % FS_CELL_FENCE) == FS_CELL_FENCE - 1) __builtin_amdgcn_sched_barrier(0);
#endif
        }
        seg.pm0 = kap * pf0; seg.pm1 = kap * pf1;
      }

      FS_T(0);
      // ================= 2. boundary rows =================
      if (t == 0) {
        R dummy; int flag = 0;
        const BCRow<R> U = geo.template boundary<BCK, 0>(usd, reach, a.B, level, 0, h[0], Q[0], R(0), dt, R(0), &dummy, &flag);
        sm.xbc[parity][0] = U.dh; sm.xbc[parity][1] = U.dq; sm.xbc[parity][2] = U.res;
        nrm2 += U.res * U.res;
      }
      if (t == tD) {
        R hD = h[0], QD = Q[0];
        int flag = 0;
#pragma unroll
        for (int j = kJD0; j <= M; ++j) if (j == jD) { hD = h[j]; QD = Q[j]; }
        const BCRow<R> Dn = geo.template boundary<BCK, 1>(dsd, reach, a.B, level, N - 1, hD, QD, QoldD, dt, Yprev, &Ynew, &flag);
        sm.xbc[parity][3] = Dn.dh; sm.xbc[parity][4] = Dn.dq; sm.xbc[parity][5] = Dn.res;
        nrm2 += Dn.res * Dn.res;
        if (flag) sm.xflag[parity] = flag;
      }

      FS_T(1);
#if FS_PHASE_FENCE & 1
      __builtin_amdgcn_sched_barrier(0);   // phases are not interleaved: it only costs registers (measured around the down-sweep: +5 %)
#endif
      // ================= 3. in-wave tree (up-sweep) =================
      constexpr bool L0R = TreeCfg<W>::kL0Regs;
      constexpr int TS = TreeCfg<W>::kSlots;
      Elim<R> e_l0;                     // level-0 record (odd lanes), only when L0R
      auto up_level = [&](auto lc) {
        constexpr int l = decltype(lc)::value;
        constexpr int d = 1 << l;
        const Seg<R> left = seg_from_below<d>(seg);
        Seg<R> mg; Elim<R> e;
        merge(left, seg, mg, e);
        if ((lane & (2 * d - 1)) == (2 * d - 1)) {
          if (L0R && l == 0) {
            e_l0 = e;
          } else {
            const int slot = (L0R ? (32 - (64 >> l)) : (64 - (64 >> l))) + (ln >> (l + 1));
            R *p = &sm.tree[wave][0][slot];
            p[0 * TS] = e.w10; p[1 * TS] = e.w11; p[2 * TS] = e.w20; p[3 * TS] = e.w21; p[4 * TS] = e.pm0;
            p[5 * TS] = e.pm1; p[6 * TS] = e.qm;  p[7 * TS] = e.sc0; p[8 * TS] = e.sc1; p[9 * TS] = e.qc;
          }
        }
        seg = mg;      // in every lane: a lane that does not survive this level is not read again (no select, no branch around the merge)
      };
      up_level(std::integral_constant<int, 0>{}); up_level(std::integral_constant<int, 1>{});
      up_level(std::integral_constant<int, 2>{}); up_level(std::integral_constant<int, 3>{});
      up_level(std::integral_constant<int, 4>{}); up_level(std::integral_constant<int, 5>{});
#if FS_PHASE_FENCE & 1
      __builtin_amdgcn_sched_barrier(0);
#endif
#if FS_PHASE_FENCE & 4
      asm volatile("" : "+v"(seg.pc0), "+v"(seg.pc1), "+v"(seg.sc0), "+v"(seg.sc1), "+v"(seg.qc));
      asm volatile("" : "+v"(seg.pm0), "+v"(seg.pm1), "+v"(seg.sm0), "+v"(seg.sm1), "+v"(seg.qm));
      __builtin_amdgcn_sched_barrier(0);
#endif
      nrm2 = wave_sum(nrm2);
      if (lane == 63) {
        R *p = sm.xseg[parity][wave];
        p[0] = seg.pc0; p[1] = seg.pc1; p[2] = seg.sc0; p[3] = seg.sc1; p[4] = seg.qc;
        p[5] = seg.pm0; p[6] = seg.pm1; p[7] = seg.sm0; p[8] = seg.sm1; p[9] = seg.qm;
        sm.xnorm[parity][wave] = nrm2;
      }
      FS_T(2);
      __syncthreads();
      FS_T(3);

      // ================= 4. across waves: fold, close with the boundary rows, unfold =================
      R tot = R(0);
      R bL0, bL1, bR0, bR1;          // updates at the first / last node of this wave's span
      {
        // pairwise tree over the W wave segments (depth log2 W instead of a serial chain of W-1 merges;
        // the two merges of a level are independent and overlap)
        Seg<R> sw[W];
        Elim<R> we[W > 1 ? W - 1 : 1];
#pragma unroll
        for (int w = 0; w < W; ++w) {
          const R *p = sm.xseg[parity][w];
          sw[w].pc0 = p[0]; sw[w].pc1 = p[1]; sw[w].sc0 = p[2]; sw[w].sc1 = p[3]; sw[w].qc = p[4];
          sw[w].pm0 = p[5]; sw[w].pm1 = p[6]; sw[w].sm0 = p[7]; sw[w].sm1 = p[8]; sw[w].qm = p[9];
          tot += sm.xnorm[parity][w];
        }
#pragma unroll
        for (int st = 1; st < W; st *= 2)
#pragma unroll
          for (int i = 0; i + st < W; i += 2 * st) merge(sw[i], sw[i + st], sw[i], we[i + st - 1]);
        BCRow<R> U, Dn;
        U.dh = sm.xbc[parity][0]; U.dq = sm.xbc[parity][1]; U.res = sm.xbc[parity][2];
        Dn.dh = sm.xbc[parity][3]; Dn.dq = sm.xbc[parity][4]; Dn.res = sm.xbc[parity][5];
        R bnd[W + 1][2];
        close_system(sw[0], U, Dn, bnd[0][0], bnd[0][1], bnd[W][0], bnd[W][1]);
#pragma unroll
        for (int st = W / 2; st >= 1; st /= 2)
#pragma unroll
          for (int i = 0; i + st < W; i += 2 * st)
            back(we[i + st - 1], bnd[i][0], bnd[i][1], bnd[i + 2 * st][0], bnd[i + 2 * st][1], bnd[i + st][0], bnd[i + st][1]);
        bL0 = bnd[0][0]; bL1 = bnd[0][1]; bR0 = bnd[1][0]; bR1 = bnd[1][1];
#pragma unroll
        for (int w = 1; w < W; ++w)
          if (wave == w) { bL0 = bnd[w][0]; bL1 = bnd[w][1]; bR0 = bnd[w + 1][0]; bR1 = bnd[w + 1][1]; }
      }
#if FS_PHASE_FENCE & 2
      asm volatile("" : "+v"(bL0), "+v"(bL1), "+v"(bR0), "+v"(bR1), "+v"(tot));
      __builtin_amdgcn_sched_barrier(0);
#endif
      FS_T(4);
      const R err = sqrt_(tot);                                        // utility.py:20-22
      if ((BCK < 2 || ds_storage) && sm.xflag[parity] != 0) status = sm.xflag[parity];     // only the storage rows raise a flag
      if (!(err == err) || !(err <= huge_norm<R>())) status = FS_NAN;        // NaN or overflow (preissmann.py:135-137)
      converged = status == FS_OK && err < a.tol;                      // preissmann.py:153
      if (DIAG && a.trace && t == 0 && it <= FS_TRACE_CAP)
        a.trace[((size_t)level * FS_TRACE_CAP + (it - 1)) * a.B + reach] = err;

      FS_T(5);
      // ================= 6. separators down the tree, local back-substitution, update ============
      // Every lane carries the updates (aL, aR) at the two ends of the group of 2^(l+1) lanes it belongs
      // to at the current level.  The group's elimination record is one LDS slot that all its lanes read
      // (a broadcast read), each lane recovers the group's middle separator itself and keeps it as its
      // new right end (lower half of the group) or left end (upper half): no cross-lane traffic and a
      // dependent chain of four fp64 operations per level.
      R aL0 = bL0, aL1 = bL1, aR0 = bR0, aR1 = bR1;
      auto load_rec = [&](auto lc) {
        constexpr int l = decltype(lc)::value;
        Elim<R> e;
        if (L0R && l == 0) {
          // level-0 records live in the odd lanes' registers: both lanes of a pair take the odd lane's copy
          auto pair = [](R v) { return dpp_mov<0xF5>(v); };     // quad_perm:[1,1,3,3]
          e.w10 = pair(e_l0.w10); e.w11 = pair(e_l0.w11); e.w20 = pair(e_l0.w20); e.w21 = pair(e_l0.w21);
          e.pm0 = pair(e_l0.pm0); e.pm1 = pair(e_l0.pm1); e.qm = pair(e_l0.qm);
          e.sc0 = pair(e_l0.sc0); e.sc1 = pair(e_l0.sc1); e.qc = pair(e_l0.qc);
        } else {
          const int slot = (L0R ? (32 - (64 >> l)) : (64 - (64 >> l))) + (ln >> (l + 1));
          const R *p = &sm.tree[wave][0][slot];
          e.w10 = p[0 * TS]; e.w11 = p[1 * TS]; e.w20 = p[2 * TS]; e.w21 = p[3 * TS]; e.pm0 = p[4 * TS];
          e.pm1 = p[5 * TS]; e.qm = p[6 * TS];  e.sc0 = p[7 * TS]; e.sc1 = p[8 * TS]; e.qc = p[9 * TS];
        }
        return e;
      };
      auto down_level = [&](auto lc, const Elim<R> &e) {
        constexpr int l = decltype(lc)::value;
        R m0, m1;
        back(e, aL0, aL1, aR0, aR1, m0, m1);
        const bool upper = ((lane >> l) & 1) != 0;
        aL0 = upper ? m0 : aL0; aL1 = upper ? m1 : aL1;
        aR0 = upper ? aR0 : m0; aR1 = upper ? aR1 : m1;
      };
      {
        // records are requested exactly one level ahead (the memory clobbers keep the compiler from
        // requesting all six up front, which costs 120 registers and sends the kernel to scratch)
        using I5 = std::integral_constant<int, 5>; using I4 = std::integral_constant<int, 4>;
        using I3 = std::integral_constant<int, 3>; using I2 = std::integral_constant<int, 2>;
        using I1 = std::integral_constant<int, 1>; using I0 = std::integral_constant<int, 0>;
        __builtin_amdgcn_sched_barrier(0);
        const Elim<R> r5 = load_rec(I5{}), r4 = load_rec(I4{});
        asm volatile("" ::: "memory");
        down_level(I5{}, r5);
        const Elim<R> r3 = load_rec(I3{});
        asm volatile("" ::: "memory");
        down_level(I4{}, r4);
        const Elim<R> r2 = load_rec(I2{});
        asm volatile("" ::: "memory");
        down_level(I3{}, r3);
        const Elim<R> r1 = load_rec(I1{});
        asm volatile("" ::: "memory");
        down_level(I2{}, r2);
        const Elim<R> r0 = load_rec(I0{});
        asm volatile("" ::: "memory");
        down_level(I1{}, r1);
        down_level(I0{}, r0);
        asm volatile("" : "+v"(aL0), "+v"(aL1), "+v"(aR0), "+v"(aR1));
        __builtin_amdgcn_sched_barrier(0);
      }
      const R dR0 = aR0, dR1 = aR1, dL0 = aL0, dL1 = aL1;

      // The update is kept pending in dh/dQ (they take the registers the elimination records free up):
      // the accepted iterate must still be intact for the level-constant pass below.
      R dh[M + 1], dQ[M + 1];
      dh[M] = dR0; dQ[M] = dR1; dh[0] = dL0; dQ[0] = dL1;
      if (Geo::kConstT && FS_LAUNDER_BACK) {
        // The continuity residuals are recomputed below on purpose (one value per node less to keep
        // across the solve).  Hide the operands so that common-subexpression elimination does not
        // resurrect the fold's copies of dQ / kc0 and keep 2 values per node alive instead.
        asm volatile("" : "+v"(kco));
        kcb = &sm.kc[0][0][0] + kco;
#pragma unroll
        for (int j = 0; j <= M; ++j) asm volatile("" : "+v"(h[j]), "+v"(Q[j]));
      }
      {
        R n0 = dR0, n1 = dR1;
        const R pfL = pf0 * dL0 + pf1 * dL1;          // (first cell's M-like row) . (update at the lane's first node)
#pragma unroll
        for (int j = M - 1; j >= 1; --j) {
          const LocalElim<R> &e = el[j - 1];
          // pivot block rows: (sm0, sm1) and the continuity row of cell j: (T_j/(2dt), -cq | T_{j+1}/(2dt), cq)
          const bool real = (RAGGED || j == M - 1) ? (s0 + j < NC) : true;   // else identity padding
          R c0, b0, qc;
          if (Geo::kConstT) {
            c0 = geo.terms_T() * r2dt; b0 = c0;
            qc = -(geo.terms_T() * (h[j] + h[j + 1]) * r2dt + cq * (Q[j + 1] - Q[j]) + kcb[(0 * M + j) * T]);
          } else {
            c0 = Tn[j] * r2dt; b0 = Tn[j + 1] * r2dt; qc = e.qc.get();
          }
          const R c1 = real ? -cq : R(0), b1 = real ? cq : R(0);
          if (!real) { c0 = R(1); b0 = R(-1); qc = R(0); }
          const R rsig = e.rq.get() - e.rk.get() * pfL;
          const R tau = qc - (b0 * n0 + b1 * n1);
          n0 = c1 * rsig - e.rs1.get() * tau;
          n1 = e.rs0.get() * tau - c0 * rsig;
          dh[j] = n0; dQ[j] = n1;
        }
      }
      FS_T(6);
#if FS_PHASE_FENCE & 8
#pragma unroll
      for (int j = 1; j < M; ++j) asm volatile("" : "+v"(dh[j]), "+v"(dQ[j]));
      __builtin_amdgcn_sched_barrier(0);
#elif FS_PHASE_FENCE & 1
      __builtin_amdgcn_sched_barrier(0);
#endif

      // ================= 5. accepted iterate -> level k =================
      if (converged) {
        if (t == 0) {
          a.hydro[((size_t)level * 4 + 0) * a.B + reach] = h[0];
          a.hydro[((size_t)level * 4 + 1) * a.B + reach] = Q[0];
          a.iters[(size_t)level * a.B + reach] = it;
        }
        {
          // Level k+1 is rebuilt from registers, so the accepted state only has to reach HBM when
          // somebody can look at it: at the last level of this launch (fs_batch_get_state, next launch)
          // and, if a history is kept, at every level.  (A transposed, fully coalesced write-back
          // through LDS was measured too: no gain, three extra barriers.)
          const bool last = (step == a.n_steps - 1);
          R *const hh_p = (DIAG && a.hist_h) ? a.hist_h + ((size_t)level * a.B + reach) * N + s0 : nullptr;
          R *const hQ_p = (DIAG && a.hist_h) ? a.hist_Q + ((size_t)level * a.B + reach) * N + s0 : nullptr;
          if (last || hh_p) {
#pragma unroll
            for (int j = 0; j <= M; ++j) {
              const int node = s0 + j;
              if ((j < M || node == N - 1) && node < N) {
                if (last) { hk_p[j] = h[j]; Qk_p[j] = Q[j]; }
                if (hh_p) { hh_p[j] = h[j]; hQ_p[j] = Q[j]; }
              }
            }
          }
        }
        FS_T(8);
        if (t == tD) {
#pragma unroll
          for (int j = kJD0; j <= M; ++j)
            if (j == jD) {
              a.hydro[((size_t)level * 4 + 2) * a.B + reach] = h[j];
              a.hydro[((size_t)level * 4 + 3) * a.B + reach] = Q[j];
            }
          Yprev = Ynew;
          if (ds_storage) a.stage_hist[(size_t)level * a.B + reach] = Ynew;
        }
        if (t == tD) {
#pragma unroll
          for (int j = kJD0; j <= M; ++j) if (j == jD) QoldD = Q[j];     // flow[k] of the next level's storage row
        }
        FS_T(9);
        if (kSaveTerms) level_constants_from_saved(h, Q);             // level constants of the next level
        else write_level_constants(h, Q);
        FS_T(10);
      }

      FS_T(5);
#pragma unroll
      for (int j = 0; j <= M; ++j) { h[j] += dh[j]; Q[j] += dQ[j]; }     // preissmann.py:146-147
    }
    if (status != FS_OK && t == 0) a.iters[(size_t)level * a.B + reach] = it - (status == FS_MAX_ITER ? 1 : 0);
    if (kBudget && a.iter_budget > 0) {
      if (t == 0) a.it_done[reach] = (converged || status != FS_OK) ? -1 : it;
      if (!converged) break;                  // budget spent: the Newton vector goes back to hg / Qg below
    }
  }

  // ---- Newton start vector of the next level + per-reach bookkeeping ----
#pragma unroll
  for (int j = 0; j <= M; ++j) {
    const int node = s0 + j;
    if ((j < M || node == N - 1) && node < N) { hg_p[j] = h[j]; Qg_p[j] = Q[j]; }
  }
  if (t == 0) a.status[reach] = status;
  if (ds_storage && t == tD) a.Yprev[reach] = Yprev;
#ifdef FS_STAMP
  if (a.dbg && lane == 0)
    for (int i = 0; i < 12; ++i) a.dbg[((size_t)reach * 16 + wave) * 12 + i] = stamp_[i];
#endif
}


// ---------------------------------------------------------------------------------------------
// Post-processing (reference: Solver.prepare_results, solver.py:65-127).  One thread per
// (reach, node) walks the stored levels: consecutive threads touch consecutive nodes, every
// load / store is coalesced, 16 B in and up to 56 B out per element - a plain HBM-bound stream.
// ---------------------------------------------------------------------------------------------
template <typename R> struct DeriveArgs {
  int32_t B, N, first, n, section_mode;
  const R *hist_h, *hist_Q;       // [levels][B][N]
  const R *geo_uniform, *geo_table;
  const R *poly_x, *poly_z;       // IRREGULAR (see KernelArgs)
  const int32_t *poly_n;
  R *level, *area, *top, *froude, *vel, *cel, *amp, *peak;   // [n][B][N] (peak: [B][N]) or nullptr
};

template <typename R> __global__ void derive_fields_kernel(const DeriveArgs<R> a) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t BN = (size_t)a.B * a.N;
  if (i >= BN) return;
  const int reach = (int)(i / a.N), node = (int)(i - (size_t)reach * a.N);
  SecParams<R> s;
  PolyNode<R> pnode;
  pnode.n = 0;
  if (a.section_mode == FS_SEC_TABLE || a.section_mode == FS_SEC_IRREGULAR) {
    auto g = [&](int row) { return a.geo_table[(size_t)row * a.N + node]; };
    s.z = g(FS_GEO_Z_BED); s.b = g(FS_GEO_B_MAIN); s.m = g(FS_GEO_M_MAIN);
    s.compound = g(FS_GEO_IS_COMPOUND) > R(0.5);
    s.hbf = g(FS_GEO_H_BANKFULL); s.bl = g(FS_GEO_B_FP_LEFT); s.br = g(FS_GEO_B_FP_RIGHT); s.mfp = g(FS_GEO_M_FP);
    if (a.section_mode == FS_SEC_IRREGULAR && a.poly_n[node] > 0) {
      pnode.x = a.poly_x + node; pnode.z = a.poly_z + node; pnode.stride = a.N; pnode.n = a.poly_n[node];
    }
  } else {
    const R z_us = a.geo_uniform[(size_t)FS_RU_Z_US * a.B + reach], z_ds = a.geo_uniform[(size_t)FS_RU_Z_DS * a.B + reach];
    const R w2 = R(node) / R(a.N - 1);
    s.z = z_us * (R(1) - w2) + z_ds * w2;
    s.b = a.geo_uniform[(size_t)FS_RU_WIDTH * a.B + reach];
    s.m = a.section_mode == FS_SEC_TRAP_UNIFORM ? a.geo_uniform[(size_t)FS_TU_SIDE_SLOPE * a.B + reach] : R(0);
    s.compound = false; s.hbf = s.bl = s.br = s.mfp = R(0);
  }
  const R h0 = a.hist_h[i];                 // depth[0] (amplitude reference, solver.py:96-97)
  R peak = R(-3.0e38);
  for (int k = 0; k < a.n; ++k) {
    const size_t src = (size_t)(a.first + k) * BN + i, dst = (size_t)k * BN + i;
    const R h = a.hist_h[src], Q = a.hist_Q[src];
    // area and top width: cross_section.py:623-679 (incl. the over-bank convention)
    const R d = fmax_(R(0), h);
    R T = s.b + R(2) * s.m * d;
    R A = (s.b + T) / R(2) * d;
    if (s.compound && d > s.hbf) {
      const R dfp = d - s.hbf, Tb = s.b + R(2) * s.m * s.hbf;
      A = (s.b + Tb) / R(2) * s.hbf + (s.bl + R(0.5) * s.mfp * dfp) * dfp + (s.br + R(0.5) * s.mfp * dfp) * dfp;
      T = (s.bl + Tb + s.br) + R(2) * s.mfp * dfp;
    }
    if (d <= R(0)) { A = R(0); T = R(0); }
    if (pnode.n > 0) poly_area_top(pnode, h + s.z, A, T);     // cross_section.py:248-328
    const R V = Q / A;
    if (a.level) a.level[dst] = h + s.z;
    if (a.area) a.area[dst] = A;
    if (a.top) a.top[dst] = T;
    if (a.froude) {                          // hydraulics.py:155-168 with its clamps
      const R Vc = Q / fmax_(A, R(1e-6)), D = A / fmax_(T, R(1e-6));
      a.froude[dst] = Vc / sqrt_(R(kG) * fmax_(D, R(1e-6)));
    }
    if (a.vel) a.vel[dst] = V;
    if (a.cel) a.cel[dst] = V + sqrt_(R(kG) * A / T);
    const R am = h - h0;
    if (a.amp) a.amp[dst] = am;
    peak = fmax_(peak, am);
  }
  if (a.peak) a.peak[i] = peak;
}

}  // namespace fs
