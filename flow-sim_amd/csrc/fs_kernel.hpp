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
//   * the 2x2-block banded Jacobian is never materialised: in characteristic-like unknowns the continuity rows
//     are substitutions and the momentum rows one scalar tridiagonal system (fs_device.hpp); each cell's entries
//     (preissmann.py:407-733) are folded straight into the lane's running segment; the 64*W lane segments are
//     reduced by a log-depth tree - DPP moves inside a wave, one LDS mailbox and one barrier across waves - the
//     upstream boundary row closes the system, the separators come back down (every lane carries two numbers of
//     its group and reads the group's record from LDS: no cross-lane traffic) and each lane back-substitutes;
//   * HBM is read once per launch and written at its last level (every level only into an optional
//     history): the accepted iterate of level k (preissmann.py:166-177; SURVEY F2: the pre-update
//     iterate) seeds the level constants of k+1 straight from registers.  Boundary hydrographs and
//     iteration counts go to small per-level tables.
//
// Template parameters: R = float|double, SEC = FS_SEC_*, M = rows per lane (>= 2), W = waves per reach,
// RAGGED = false promises N = 64*W*M: then every row is a cell except the very last one (the downstream
// boundary row) and the per-row selects (and their 64-bit lane masks) disappear;
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
#ifndef FS_LONG_RECOMPUTE
#define FS_LONG_RECOMPUTE 1   // multi-pass kernel, uniform sections: level constants recomputed per sweep instead of stored (fs_long.hpp)
#endif
#ifndef FS_LONG_PREFETCH
#define FS_LONG_PREFETCH 0    // multi-pass kernel: the next pass's lines requested into L2 a pass ahead
#endif
#ifndef FS_LONG_COALESCE
#define FS_LONG_COALESCE 1    // multi-pass kernel: state loads and stores with consecutive lanes on consecutive nodes, transposed through LDS
#endif
#ifndef FS_LONG_WPE
#define FS_LONG_WPE 2
#endif
#ifndef FS_POLY_HINT_MAXM
#define FS_POLY_HINT_MAXM 8
#endif
#ifndef FS_POLY_HINT
#define FS_POLY_HINT 1     // polyline nodes start from the stage-table interval of their last evaluation (fs_poly.hpp)
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
#ifndef FS_PIN_SEG_F32
#define FS_PIN_SEG_F32 1 // the same switch for the fp32 instantiations (C5 fp32: +5 % with the pin)
#endif
#ifndef FS_PIN_SEG
#define FS_PIN_SEG 0     // pin the running rows at the end of each cell's scheduling region (round 1's block fold: +3 % at M >= 8; the
#endif                   // scalar fold has 8 running numbers instead of 10 + a factor and schedules better without: +1.2 %)
#ifndef FS_PHASE_FENCE
#define FS_PHASE_FENCE 0   // bit 3: pin the back-substituted updates and fence them off from the acceptance block (round 1: +0.9 %; with the
#endif                     // round-2 solve: fp64 flagship -2.3 %, C5 fp32 +4.5 %: see FS_PHASE_FENCE_F32); bits 0-1 (other phase boundaries): no gain
#ifndef FS_PHASE_FENCE_F32
#define FS_PHASE_FENCE_F32 8   // the same switch for the fp32 instantiations (three waves per SIMD at 168 registers: the fence keeps them there)
#endif
// table kernels with 2 rows per lane: the section parameters of the lane's three nodes are loaded once per launch instead of
// at every node evaluation (46 vector loads per Newton iteration, their latency only half hidden by the second wave of the
// SIMD): C4 +8.5 %, the general table kernel +11.6 %, still within the 256-register cap (20 spilled registers)
#ifndef FS_REG_GEO
#define FS_REG_GEO 1
#endif
#ifndef FS_FLAT_BC
#define FS_FLAT_BC 1     // kernels compiled for a boundary pair evaluate the two rows in every lane, without a branch (below)
#endif
#ifndef FS_LAUNDER_BACK
#define FS_LAUNDER_BACK 1
#endif
#ifndef FS_MONITOR_ALL
#define FS_MONITOR_ALL 0   // 1: the conditioning monitor (below) also in the kernels compiled without diagnostics (DIAG = false); measured
#endif                     // cost on the flagship: profiles/round3/README.md
#ifndef FS_XLANES_MINW
#define FS_XLANES_MINW 8 // waves per reach from which the cross-wave step runs as a second, small DPP tree (lane w of every wave carries
#endif                   // the segment of wave w: 8 live numbers) instead of every thread folding all W segments in its own registers
                         // (16 W live doubles).  Measured on 65 536 x 4 096 (profiles/round3/second_wave_per_simd.md): W = 8 (the (8, 8)
                         // shape, two waves per SIMD) 7.9e6 -> 9.8e6 (the per-thread fold spills at 256 registers); W = 4 (the flagship
                         // (16, 4) shape) 1.072e7 -> 1.058e7: with four segments the per-thread fold has the shorter dependent chain
#ifndef FS_XWAVE_CONT
#define FS_XWAVE_CONT 0      // 1: multi-wave kernels without the conditioning monitor (the no-diagnostics benchmark shapes) solve the W + 1
#endif                       // unknowns of the cross-wave step (p of the first row, m of each wave's last row) as ONE small tridiagonal system by
                             // forward and backward continuants - two independent chains of W + 1 fmas and one reciprocal - instead of folding the
                             // W wave segments pairwise (two dependent merge levels, the root closure, two unfolding levels: four reciprocal
                             // chains).  Built and measured in round 4 (profiles/round4/cross_wave_continuants.txt): the phase itself gets
                             // shorter (1 330 -> 1 170 ticks of 15 500 in the stamped build), the shipped flagship does not get faster
                             // (121.1 -> 122.2 ms; 121.0 with the top tree record no longer requested ahead of the barrier): off
#ifndef FS_XWAVE_FENCE
#define FS_XWAVE_FENCE 0     // scheduling fences around the cross-wave step (bit 0: before, bit 1: after)
#endif
#ifndef FS_TREE_REGS
#define FS_TREE_REGS 0       // flagship shapes: the in-wave tree's records stay in registers (6 levels x 4 numbers, valid in the lane that survives
#endif                       // its level) and come back on the way down by DPP broadcasts / readlanes instead of through LDS slots
#ifndef FS_RC_EARLY
#define FS_RC_EARLY 0        // flagship shapes: the continuity residuals the back-substitution needs are recomputed inside the cross-wave step
#endif
#ifndef FS_PREFETCH_DOWN
#define FS_PREFETCH_DOWN 1   // multi-wave kernels without diagnostics: this many of the wave's top tree records (levels 5, 4, 3) are requested
#endif                       // ahead of the cross-wave barrier instead of after the cross-wave step
#ifndef FS_PRIME
#define FS_PRIME 0      // 1: the level constants of a launch's first level come from the acceptance block of the loop (a priming pass
#endif                   // through it) instead of from a second instance of that code ahead of the loop.  Not needed: with
                         // -ffp-contract=on (Makefile) the two instances compile to the same arithmetic and chunked stepping equals
                         // one launch bit for bit (tests/test_gpu_parity.py, test_gpu_dropin.py); flagship -5.5 %; kept as a switch

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

template <typename R> __host__ __device__ constexpr R finite_max() { return sizeof(R) == 8 ? R(1.7976931348623157e308) : R(3.4028234e38); }
template <typename R> __host__ __device__ constexpr R eps_of() { return sizeof(R) == 8 ? R(2.220446049250313e-16) : R(1.1920929e-7); }

// Conditioning monitor.  The up row of a segment [a, b) reads  u1 p_a + m_a + u3 m_{b-1} = ru : u3 is the product of
// -(super-diagonal / pivot) over the segment's rows, i.e. how strongly the first unknown of the segment depends on its last
// one.  Subcritical, the rows are diagonally dominant and |u3| decays along the reach (typically 0.2 ... 0.6 over a whole
// reach); with v > c the upstream-travelling characteristic turns round, |u3| grows geometrically with the length of the
// segment, the two-point boundary value problem (one condition per end, boundary.py) is ill-posed and the Jacobian the
// reference hands to SuperLU is ill-conditioned (cond 1e13 ... 1e16 on the draws of profiles/round2/supercritical_scan.txt
// that part from the pivoted CPU solvers).  The kernel keeps the largest |u3| of any segment the tree forms - one integer
// (the high word of the magnitude) carried along with the segment - and raises FS_ILL_CONDITIONED above 2^10 (calibration:
// DESIGN.md section 4.1; every reference-generated near-critical fixture the kernel misses by more than 1e-8 lies above): the reference's
// `diagnos` check (preissmann.py:139-144, rcond < 1e-12 -> ValueError "Jacobian is ill-conditioned") stands behind it.
__device__ __forceinline__ int hi_abs(double v) { return __double2hiint(v) & 0x7fffffff; }
__device__ __forceinline__ int hi_abs(float v) { return __float_as_int(v) & 0x7fffffff; }
template <typename R> __host__ __device__ constexpr int growth_limit_bits() { return sizeof(R) == 8 ? ((1023 + 10) << 20) : ((127 + 10) << 23); }
__device__ __forceinline__ int max_(int a, int b) { return a > b ? a : b; }

template <typename R> struct KernelArgs {
  int32_t B, N, n_steps, level0, max_iter;
  int32_t iter_budget;     // kernels of boundary class -1 only: > 0 = Newton iterations this launch may spend on a reach (fs_batch_iterate)
  int32_t *it_done;        // [B] with iter_budget: iterations already spent on the open level, -1 once the reach has closed it
  R theta, dt, dx, tol;
  R *hk, *Qk;              // [B][N] accepted state of the current level (in: level0, out: level0+n_steps)
  R *hg, *Qg;              // [B][N] Newton start vector for the next level
  const R *geo_uniform;    // RECT_UNIFORM: [FS_RU_NPARAM][B]
  const R *geo_table;      // TABLE: [FS_GEOX_NROWS][N], shared by the batch, or one such table per reach (geo_reach_stride)
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
  // ---- round 3 (kept behind the fields above: the benchmark kernels read none of these, and their code - down to the scalar
  // loads of the arguments - stays what it was) ----
  int64_t geo_reach_stride;      // elements between the tables of consecutive reaches (0: one table for the whole batch)
  int64_t poly_reach_stride;     // IRREGULAR: the same for poly_x / poly_z (P * N); poly_lim and poly_n follow with 2 N and N
  // heterogeneous batches (fs_batch_set_reach_*): each reach its own channel length, grid and scheme - what the reference
  // builds per Channel / Solver (channel.py:213-241, solver.py:34-38,53-55)
  const int32_t *reach_nodes;    // [B] or nullptr: nodes of each reach (<= N; N stays the row stride of every [B][N] array)
  const R *reach_scheme;         // [5][B] or nullptr: theta, dt, dx, tolerance, max_iter of each reach
  const int32_t *reach_kinds;    // [2][B] or nullptr: boundary kind of each reach, upstream row then downstream row (kinds <= FS_BC_STORAGE, or FS_BC_HOST_ROW)
  const R *poly_tz;              // IRREGULAR: stage tables (fs_poly.hpp): breakpoints [N][KP], then intervals [P][16][N] pairs
  int32_t poly_K;                // IRREGULAR: P, intervals per node in the stage tables (0: no tables, walk the edges)
  R *kc_scratch;           // long reaches (fs_long.hpp): [B][4][passes * 64 W M] level constants, owned by the batch
  int32_t passes;          // long reaches: passes of 64 W M rows a workgroup makes over its reach
  // ---- round 4: a reach longer than one lane grid as a TEAM of workgroups (kTeam below) ----
  int32_t team_size;             // G: workgroups per reach, member g owns rows [g C, (g + 1) C), C = 64 W M
  R *team_mail;                  // [B][2][G W + 1][kTeamWords] mailboxes: one slot per (member, wave) + one for the upstream row, per iteration parity
  unsigned long long *team_sync; // [1 + B]: [0] the ticket counter of the launch, [1 + reach] posts made for that reach so far (zeroed before every launch)
  uint32_t team_epoch;           // tagged mailbox (FS_TEAM_TAGGED): the number of this launch among the handle's team launches (>= 1): the high half of every tag
};
constexpr int kTeamSlots = 64;   // (member, wave) segments of a team: the top tree is one wave wide
// how long a member waits for its team before it gives the reach up (FS_TEAM_STALL): s_memtime ticks, 2.0e9 per second on gfx950 (tools/micro/rates.hip:
// 158 795 ticks in 0.079 ms of HIP-event time).  Eight seconds: a team's members start in ticket order as CUs become free, so the wait is bounded by the
// longest kernel of a co-tenant that holds the CUs meanwhile, not by anything of this library's own
constexpr unsigned long long kTeamPatience = 16000000000ull;
constexpr int kTeamWords = 12;   // per slot: the segment (8), the wave's residual sum (1), two ints (monitor word, boundary flag), pad

template <typename R, int SEC> struct Geometry;

// A team's mailbox (kTeam in the step kernel): every word is written and read with agent-scope atomic accesses - device-coherent
// stores and loads (sc1), no cache involved that another XCD could not see - so that the exchange needs neither the write-back of
// the L2 that an agent-scope release costs nor the invalidation of an acquire, both whole-cache operations and both on the critical
// path of every Newton iteration of every team (measured: the exchange took 22 600 ticks of a 43 000-tick iteration with them).
// Order is kept the plain way: the poster waits for its stores to be acknowledged (vmcnt) before the workgroup's barrier, thread 0
// bumps the reach's counter after it; a reader polls the counter, passes a barrier, and only then issues its loads.
template <typename T> __device__ __forceinline__ void team_put(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void team_posted() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Tagged form (FS_TEAM_TAGGED): every mailbox word travels as ONE 16-byte device-coherent store of (value, tag), tag = (launch, exchange);
// a reader polls the words themselves until every tag is the one it waits for.  No counter, no acknowledged stores, no barrier between
// posting and counting: the exchange is one store flight and the reader's polls.  (A lane's aligned 16-byte access is one request.)
#ifndef FS_TEAM_TAGGED
#define FS_TEAM_TAGGED 1
#endif
#ifndef FS_TEAM_SLEEP
#define FS_TEAM_SLEEP 1      // s_sleep between two polls of the mailbox (units of 64 cycles)
#endif
typedef unsigned int fs_u4 __attribute__((ext_vector_type(4)));
constexpr int kTeamAuxSc1 = 16;      // cache-policy operand of the raw buffer intrinsics on gfx94x / gfx950: sc1 (device-coherent)
__device__ __forceinline__ void team_put2(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned long long bits, unsigned long long tag) {
  fs_u4 v;
  v.x = (unsigned)bits; v.y = (unsigned)(bits >> 32); v.z = (unsigned)tag; v.w = (unsigned)(tag >> 32);
  __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, kTeamAuxSc1);
}
__device__ __forceinline__ fs_u4 team_get2(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, kTeamAuxSc1); }
__device__ __forceinline__ unsigned long long team_tag(fs_u4 v) { return ((unsigned long long)v.w << 32) | v.z; }
__device__ __forceinline__ unsigned long long team_bits(fs_u4 v) { return ((unsigned long long)v.y << 32) | v.x; }

// the descriptor with its kind pinned to what the instantiation was compiled for (BCK >= 2)
template <int BCK, int SIDE, typename R>
__device__ __forceinline__ BCDesc<R> pinned(const BCDesc<R> &bc) {
  BCDesc<R> d = bc;
  if (BCK >= 2) d.kind = SIDE == 0 ? (int)FS_BC_FLOW_HYDROGRAPH : BCK - 2;
  return d;
}

template <typename R> struct Geometry<R, FS_SEC_RECT_UNIFORM> {
  static constexpr bool kConstT = true;      // dA/dh = b everywhere: no per-node top width to keep
  R b, rb, n, rn, z_us, z_ds, inv_nm1, dz;
  __device__ __forceinline__ void init(const KernelArgs<R> &a, int reach, int n_nodes) {
    b = a.geo_uniform[(size_t)FS_RU_WIDTH * a.B + reach];
    n = a.geo_uniform[(size_t)FS_RU_MANNING * a.B + reach];
    z_us = a.geo_uniform[(size_t)FS_RU_Z_US * a.B + reach];
    z_ds = a.geo_uniform[(size_t)FS_RU_Z_DS * a.B + reach];
    rb = R(1) / b; rn = R(1) / n;
    inv_nm1 = R(1) / R(n_nodes - 1);
    dz = (z_ds - z_us) * inv_nm1;
  }
  __device__ __forceinline__ R bed_step(int) const { return dz; }   // bed(node+1) - bed(node)
  __device__ __forceinline__ R terms_T() const { return b; }
  __device__ __forceinline__ R rT_const() const { return rb; }
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
    s.sm = R(1); s.rnm = rn; s.sfp = R(1); s.Tb = b; s.Am = R(0); s.Pm = b; s.km15 = s.kl15 = s.kr15 = R(0);   // (not compound: unused)
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
  R b, m, sm2, n, rn, z_us, z_ds, inv_nm1, dz;
  __device__ __forceinline__ void init(const KernelArgs<R> &a, int reach, int n_nodes) {
    b = a.geo_uniform[(size_t)FS_RU_WIDTH * a.B + reach];
    n = a.geo_uniform[(size_t)FS_RU_MANNING * a.B + reach];
    rn = R(1) / n;
    z_us = a.geo_uniform[(size_t)FS_RU_Z_US * a.B + reach];
    z_ds = a.geo_uniform[(size_t)FS_RU_Z_DS * a.B + reach];
    m = a.geo_uniform[(size_t)FS_TU_SIDE_SLOPE * a.B + reach];
    sm2 = R(2) * sqrt_(R(1) + m * m);
    inv_nm1 = R(1) / R(n_nodes - 1);
    dz = (z_ds - z_us) * inv_nm1;
  }
  __device__ __forceinline__ R bed_step(int) const { return dz; }
  __device__ __forceinline__ R terms_T() const { return R(0); }      // unused (kConstT == false)
  __device__ __forceinline__ R rT_const() const { return R(0); }
  __device__ __forceinline__ R bed(int node) const {
    const R w2 = R(node) * inv_nm1;
    return z_us * (R(1) - w2) + z_ds * w2;
  }
  __device__ __forceinline__ NodeTerms<R> terms(int, R h, R Q) const { return node_terms_trap(b, m, sm2, n, h, Q); }
  __device__ __forceinline__ SecParams<R> section(int node) const {
    SecParams<R> s;
    s.z = bed(node); s.b = b; s.m = m; s.nm = n; s.nl = n; s.nr = n; s.hbf = R(0);
    s.bl = R(0); s.br = R(0); s.mfp = R(0); s.curv = R(0); s.compound = false;
    s.sm = R(0.5) * sm2; s.rnm = rn; s.sfp = R(1); s.Tb = b; s.Am = R(0); s.Pm = b; s.km15 = s.kl15 = s.kr15 = R(0);   // (not compound: unused)
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
  R n_over, rn_over, k15_over;      // per-reach main-channel Manning n (ensembles), its reciprocal and its -1.5 power
  bool has_over;
  __device__ __forceinline__ void init(const KernelArgs<R> &a, int reach, int) {
    tab = a.geo_table + (size_t)reach * a.geo_reach_stride; N = a.N;       // (N: the row stride of the table)
    has_over = a.n_override != nullptr;
    n_over = has_over ? a.n_override[reach] : R(1);
    rn_over = R(1) / n_over; k15_over = pm15_(n_over);
  }
  __device__ __forceinline__ R terms_T() const { return R(0); }      // unused (kConstT == false)
  __device__ __forceinline__ R rT_const() const { return R(0); }
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
    s.sm = g(FS_GEOX_SM); s.sfp = g(FS_GEOX_SFP); s.Tb = g(FS_GEOX_TB); s.Am = g(FS_GEOX_AM); s.Pm = g(FS_GEOX_PM);
    s.rnm = has_over ? rn_over : g(FS_GEOX_RNM); s.km15 = has_over ? k15_over : g(FS_GEOX_KM15);
    s.kl15 = g(FS_GEOX_KL15); s.kr15 = g(FS_GEOX_KR15);
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
  const R *px, *pz, *plim, *ptz;
  const int32_t *pn;
  int pK;
  __device__ __forceinline__ void init(const KernelArgs<R> &a, int reach, int n_nodes) {
    tb.init(a, reach, n_nodes);
    const bool own = a.poly_reach_stride != 0;
    px = a.poly_x + (size_t)reach * a.poly_reach_stride; pz = a.poly_z + (size_t)reach * a.poly_reach_stride;
    plim = a.poly_lim + (own ? (size_t)reach * 2 * a.N : 0); pn = a.poly_n + (own ? (size_t)reach * a.N : 0);
    pK = a.poly_K;                                     // stage tables: N * poly_table_stride(P) numbers per reach (or shared)
    ptz = pK ? a.poly_tz + (own ? (size_t)reach * a.N * poly_table_stride(pK) : 0) : nullptr;
  }
  __device__ __forceinline__ R terms_T() const { return R(0); }
  __device__ __forceinline__ R rT_const() const { return R(0); }
  __device__ __forceinline__ R bed_step(int node) const { return tb.bed_step(node); }
  __device__ __forceinline__ R bed(int node) const { return tb.bed(node); }
  __device__ __forceinline__ SecParams<R> section(int node) const { return tb.section(node); }
  // the node's slot in the interval part of the stage tables (behind the [N][KP] breakpoints; fs_poly.hpp)
  __device__ __forceinline__ const R *table_slot(int node) const { return ptz + (size_t)tb.N * poly_table_bp(pK) + 2 * (size_t)node; }
  __device__ __forceinline__ PolyNode<R> poly(int node) const {
    PolyNode<R> p;
#ifdef FS_BOUNDS
    node = (int)FS_CHK(16, node, tb.N);
#endif
    const int N = tb.N;
    typedef const __attribute__((address_space(1))) R *GlobalR;              // (device memory: global loads, not flat ones)
    typedef const __attribute__((address_space(1))) int32_t *GlobalI;
    auto g = [&](int row) { return ((GlobalR)tb.tab)[(size_t)row * N + node]; };
    p.x = px + node; p.z = pz + node; p.stride = N; p.n = ((GlobalI)pn)[node];
    p.nl = g(FS_GEO_N_LEFT); p.nm = tb.has_over ? tb.n_over : g(FS_GEO_N_MAIN); p.nr = g(FS_GEO_N_RIGHT);
    p.liml = ((GlobalR)plim)[node]; p.limr = ((GlobalR)plim)[N + node];
    p.curv = g(FS_GEO_CURVATURE); p.zmin = g(FS_GEO_Z_BED);
    p.K = pK; p.KP = poly_table_bp(pK);
    p.tz = ptz ? ptz + (size_t)node * p.KP : nullptr; p.tco = ptz ? table_slot(node) : nullptr; p.cstride = N;
    return p;
  }
  __device__ __forceinline__ NodeTerms<R> terms(int node, R h, R Q) const {
    if (((const __attribute__((address_space(1))) int32_t *)pn)[node] > 0) return node_terms_poly(poly(node), h, Q);
    return node_terms_general_call(section(node), h, Q);
  }
  // The same starting from the table interval of the node's last evaluation (fs_poly.hpp: node_terms_poly_hinted).  kh is the
  // caller's per-row hint: -2 a node without a polyline, -1 nothing to start from, >= 0 the interval.
  __device__ __forceinline__ int hint_init(int node) const {
    return ((const __attribute__((address_space(1))) int32_t *)pn)[node] > 0 ? -1 : -2;
  }
  __device__ __forceinline__ static TermsHint<R> terms_scan(const Geometry g, int node, R h, R Q) {
    TermsHint<R> r;
    r.t = node_terms_poly(g.poly(node), h, Q, &r.k, &r.bc);
    return r;
  }
  // (The paths beside the hinted one take the node index through an empty asm: everything they derive from it - a dozen row
  // addresses of the parameter table, the polyline's station pointers - is loop-invariant, and the compiler otherwise computes it
  // ahead of the time loop and keeps it in registers the Newton loop does not have: they went to scratch, and the hinted path
  // fetched its own table pointer and interval from scratch ahead of every evaluation, a memory round trip in front of the one it needs.)
  __device__ __forceinline__ static int opaque(int node) { asm volatile("" : "+v"(node)); return node; }
  __device__ __forceinline__ NodeTerms<R> terms_hinted(int node, R h, R Q, int &kh, PolyBC<R> &bc) const {
#ifdef FS_BOUNDS
    node = (int)FS_CHK(15, node, tb.N);
    if (kh >= 0) kh = (int)FS_CHK(14, kh, pK);
#endif
    if (kh == -2) return node_terms_general_call(section(opaque(node)), h, Q);
    return node_terms_poly_hinted(table_slot(node), tb.N, tb.has_over, tb.n_over, kh, h, Q, bc,
                                  [&]() { return terms_scan(*this, opaque(node), h, Q); });
  }
  // (No device function of this library is called out of line.  Until round 3 the kernels with many rows per lane called the
  // polyline evaluation, and the edge walk behind the stage tables was a call everywhere: the compiler's interprocedural register
  // allocation then trusted a register summary of the callee that did not hold - the (8, 1) polyline kernel returned wrong numbers
  // from the second level on, or not, depending on unrelated code around the call, the (8, 4) one faulted, and both are correct
  // with -mllvm -enable-ipra=0.  Inlined, nothing is left to summarise; the kernels are also faster: (2, 1) +8 %, (8, 1) +16 %,
  // (8, 4) +32 % on the polyline ensemble, profiles/round3/polyline_calls.txt.)
  template <int BCK, int SIDE>
  __device__ __forceinline__ BCRow<R> boundary(const BCDesc<R> &bc_, int reach, int B, int level, int node, R h, R Q,
                                               R Qold, R dt, R Yprev, R *Ynew, int *flag) const {
    const BCDesc<R> bc = pinned<BCK, SIDE>(bc_);
    if (pn[node] > 0 && bc.kind == FS_BC_NORMAL_DEPTH)
      return bc_normal_depth_poly(poly(node), ((LdsParams<R>)bc.params)[0], ((LdsParams<R>)bc.params)[1], h, Q);   // (LDS copy, as bc_eval)
    if (BCK != 0 && pn[node] > 0 && bc.kind == FS_BC_STORAGE_CURVE) {
      const PolyNode<R> nd = poly(node);
      int ns_;
      const PolyEval<R> er = poly_eval_whole(nd, nd.zmin + h, &ns_);
      const PolyEval<R> ed = poly_eval_whole(nd, h + bc_param(bc, FS_SC_BED_LEVEL, reach, B), &ns_);
      EntryProps<R> pr{er.A, er.Rh, er.neq, er.dRdA, er.dAdh}, pd{ed.A, ed.Rh, ed.neq, ed.dRdA, ed.dAdh};
      return bc_storage_curve(bc, reach, B, level, pr, pd, h, Q, Qold, dt, Yprev, Ynew, flag);
    }
    return bc_eval<BCK != 0>(bc, reach, B, level, section(node), h, Q, Qold, dt, Yprev, Ynew, flag);
  }
};

// LDS carve-up for one reach
// (A, Se, Q/A) of the nodes as the last fold saw them (kSaveTerms); an empty base otherwise
template <typename R, int M, int T, bool SAVE> struct SavedTerms { R nt[3][M + 1][T]; };
template <typename R, int M, int T> struct SavedTerms<R, M, T, false> {};

// what a team member needs on top (kTeam): the records of the top tree over the team's (member, wave) segments and its results
template <typename R, bool TEAM> struct TeamSmem {
  R xtree[4][64];          // records of the top tree (wave 0 of every member computes it, redundantly)
  R xres[kTeamSlots][4];   // per (member, wave): p of its first row, m of its last row, m of its first row, m of the next one's first row
  R xtot;
  int32_t xwarn, xflagT, xstall, xjob;
};
template <typename R> struct TeamSmem<R, false> {};

template <typename R, int M, int W, bool SAVE, bool TEAM = false> struct Smem : SavedTerms<R, M, 64 * W, SAVE>, TeamSmem<R, TEAM> {
  static constexpr int T = 64 * W;
  R kc[4][M][T];           // per-cell level-k constants, lane-minor (conflict-free ds_read_b64)
  R tree[W][4][64];        // per-wave records of the in-wave tree (Elim: 4 numbers per merge, 63 merges)
  R xseg[2][W][8];         // wave segments, double-buffered by iteration parity
  R xbc[2][4];             // upstream boundary row on (p_0, m_0): aU, bU, rU
  R bcp[2][FS_BC_MAX_PARAMS];   // this reach's boundary parameters (fixed-size kinds), read every Newton iteration
  R xnorm[2][W];
  int32_t xflag[2];
  int32_t xg[2][W];        // conditioning monitor: high word of the largest |u3| of each wave's tree
};

// What back-substitution needs for row j of a lane's chunk: m_{j-1} = R3 - R1 p_a - R2 m_j (the running down row at the
// time row j was added, divided by its pivot).  The continuity residual rc_{j-1} that turns m_{j-1} into p_j is
// recomputed from the still un-updated state when the top width is constant (kConstT), else it is kept in qc.
template <typename R> struct LocalElim { Parked<R> R1, R2, R3, qc; };

// Minimum number of waves per SIMD a kernel is compiled for: caps its registers at 512 / n.  One wave per SIMD cannot
// hide the latency of the in-wave tree, a second one is worth 20-80 % wherever the kernel fits 256 registers or nearly
// does; forcing it on the larger kernels sends them to scratch (measured 0.25-0.8x).
#ifndef FS_W1_LANES
#define FS_W1_LANES 1      // one-wave fp64 kernels: the root segment and the upstream row meet through v_readlane instead of an LDS round trip (C5 fp64 +5.5 %, polyline ensemble +2.8 %, C4 +1.9 %; fp32: -0.5 %, left as it was)
#endif
#ifndef FS_TEAM_WPE
#define FS_TEAM_WPE 2      // team kernels with <= 8 rows per lane: two workgroups per CU (one computes while the other waits for its team)
#endif
template <typename R, int SEC, int M, int W, int BCK, bool TEAM = false> constexpr int min_waves() {
  if (TEAM && M <= 8 && sizeof(R) == 8) return FS_TEAM_WPE;
#ifdef FS_WPE_TRAP42      // experiment: the (4, 2) trapezoid kernel (272 registers) capped at 256, two waves per SIMD
  if (W == 2 && M == 4 && SEC == FS_SEC_TRAP_UNIFORM && sizeof(R) == 8) return FS_WPE_TRAP42;
#endif
  if (W > 1) return 1;          // multi-wave table kernels with 2 cells per lane at two waves per SIMD: no better than the 4- and 8-cell ones
  if (sizeof(R) == 4) return (M <= 8 && (SEC == FS_SEC_RECT_UNIFORM || SEC == FS_SEC_TRAP_UNIFORM)) ? FS_WPE_W1_F32_UNIFORM : FS_WPE_W1_F32;
  if (SEC == FS_SEC_RECT_UNIFORM && BCK >= 1 && M <= 8) return FS_WPE_RECT8;
  if (BCK >= 2 && M <= 2) return FS_WPE_PINNED_SHORT;
  if (BCK == 0 && M <= 2 && SEC == FS_SEC_TABLE) return FS_WPE_LEAN_SHORT;
  if (BCK == 0 && M == 4 && SEC == FS_SEC_TRAP_UNIFORM) return FS_WPE_TRAP4;
  if (BCK == 0 && M <= 2 && SEC == FS_SEC_IRREGULAR) return FS_WPE_LEAN_POLY;
  return FS_WPE_W1;
}

// Rows of the scalar system (fs_device.hpp): row k is the momentum row of cell k for k < N-1, the downstream boundary row
// for k = N-1 and an identity row (m_k = 0) beyond; lane t owns rows [t M, (t+1) M) and holds the nodes t M .. (t+1) M.
// RAGGED = false promises N = 64 W M: every row is a cell but the very last one, which is the boundary row.
// DIAG = false: no per-level history and no residual trace (batches created without FS_FLAG_HISTORY / FS_FLAG_TRACE): the
// stores are never executed there, but compiled in they cost the flagship kernel 1.1 %
// TAIL >= 0 (ragged one-wave kernels only; round 4): "every row a cell except in the tail lanes".  The batch's N is not the lane grid's,
// but the position of the downstream boundary row inside its lane is known at compile time (local row TAIL = (N - 1) mod M), so the
// per-row selects of the ragged form are gone: every row is assembled as a cell; in local row TAIL one select per number - taken by
// the ONE lane that owns node N - 1 - puts the boundary row in; the rows beyond it are not replaced by identity rows but left as the
// cells of clamped, frozen copies of node N - 1 ("phantom" cells: finite, diagonally dominant, and behind a boundary row whose
// super-diagonal is zero, so nothing upstream of it ever reads them - the real unknowns get the very bits of the ragged kernel);
// two lane masks hoisted out of the time loop keep the phantom rows out of the residual norm and the phantom nodes where they are.
// TEAM (round 4): a reach LONGER than one lane grid (N > 64 W M rows) advanced by a team of G = a.team_size workgroups, each holding
// C = 64 W M rows of it on chip exactly as a short reach is held - unknowns in registers, level constants in LDS, HBM read once per
// launch and written once - where the multi-pass kernel (fs_long.hpp) streams the Newton vector through memory twice per iteration.
// What the W waves of one workgroup exchange through LDS (step 4) the G W waves of a team exchange through a mailbox in device
// memory: every wave posts its segment, one release / acquire pair on a per-reach counter (agent scope) stands where the barrier
// stood, wave 0 of every member reduces the G W segments with one more DPP tree (the multi-pass kernel's top tree: identity segments
// pad the 64 lanes), closes the root with the upstream row and hands every wave its four numbers back through LDS.  All members
// see the same numbers and take the same decisions (convergence, failure, monitor), so the time loop needs no other exchange.
// Membership is by TICKET, not by blockIdx: a workgroup takes the next (reach, member) job when it starts, so the members of the
// oldest unfinished team are exactly the workgroups that started first - all resident, whatever order the dispatcher chose and
// whatever else runs on the device; a team therefore never waits for a workgroup that cannot start (no co-residency assumption
// beyond G <= CUs).  A wait that still exceeds ~2 s of s_memtime ends the reach with FS_TEAM_STALL instead of spinning on.
template <typename R, int SEC, int M, int W, bool RAGGED = true, int BCK = 0, bool DIAG = true, int TAIL = -1, bool TEAM = false>
__global__ __launch_bounds__(64 * W, (min_waves<R, SEC, M, W, BCK, TEAM>())) void preissmann_step_kernel(const KernelArgs<R> a) {
  static_assert(M >= 2, "a lane's segment needs two rows (its up and its down row)");
  constexpr bool kTail = TAIL >= 0;
  constexpr bool kTeam = TEAM;
  static_assert(!kTeam || (W > 1 && BCK >= 0 && !kTail), "team form: multi-wave kernels, boundary kinds of the device");
  static_assert(!kTail || (RAGGED && W == 1 && TAIL < M && SEC == FS_SEC_TABLE && BCK >= 2), "tail-only form: ragged one-wave table kernels compiled for a boundary pair");
  constexpr int T = 64 * W;
  using Geo = Geometry<R, SEC>;
  // Short general-section kernels keep (A, Se, Q/A) of every node of the current fold in LDS: if the iterate is accepted they
  // are the node terms of level k, and the level constants of level k+1 come from them instead of from another pass over the
  // sections (1 of 6 section passes of the polyline ensemble, 1 of 11 of C4)
  constexpr bool kRegGeo = FS_REG_GEO && SEC == FS_SEC_TABLE && W == 1 && M <= 2;
  constexpr bool kFlatBC = FS_FLAT_BC && BCK >= 2 && sizeof(R) == 8 && SEC == FS_SEC_TRAP_UNIFORM;   // measured: C5 fp64 +1.5 %; flagship -1.5 %, C4 -7 %, polyline -1.3 %
  constexpr bool kSaveTerms = FS_SAVE_TERMS && !Geometry<R, SEC>::kConstT && (M <= 2 || (sizeof(R) == 8 && M <= FS_SAVE_TERMS_MAXM_F64));
  __shared__ Smem<R, M, W, kSaveTerms, kTeam> sm;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int job = blockIdx.x;
  if constexpr (kTeam) {      // the next job in order of ARRIVAL (see above)
    if (t == 0) sm.xjob = (int)__hip_atomic_fetch_add(a.team_sync, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    job = sm.xjob;
  }
  const int G = kTeam ? a.team_size : 1;
  const int reach = kTeam ? job / G : job, member = kTeam ? job - reach * G : 0;
  const int gt = member * T + t;                 // the lane's place in the reach's lane grid (t itself unless the reach is a team's)
  constexpr bool kTagged = kTeam && FS_TEAM_TAGGED && sizeof(R) == 8;
  // the reach's mailbox of one iteration parity as a raw buffer (tagged form: (G W + 1) slots of kTeamWords (value, tag) pairs)
  auto mbox = [&](int par) __attribute__((always_inline)) {
    const size_t bytes = (size_t)(G * W + 1) * kTeamWords * 16;
    char *base = reinterpret_cast<char *>(a.team_mail) + ((size_t)reach * 2 + par) * bytes;
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x27000);
  };
  // One Newton iteration per launch for batches with host-evaluated boundary rows (FS_BC_HOST_ROW, fs_batch_iterate): the
  // kernels of boundary class -1 carry an iteration budget and the count of the open level across launches.
  constexpr bool kBudget = (BCK == -1);
  int it_entry = 0;
  if (kBudget && a.iter_budget > 0) {
    it_entry = a.it_done[reach];
    if (it_entry < 0) return;                 // this reach has closed the level: it waits for the others (whole workgroup)
  }
  // nodes of this reach; NS = a.N stays the row stride of every [B][N] array (heterogeneous batches: RAGGED kernels only)
  const int NS = a.N;
  const int N = (RAGGED && a.reach_nodes) ? a.reach_nodes[reach] : a.N, NC = N - 1;
  const int s0 = gt * M;                      // first node / row of this lane
  // lane (of the reach's lane grid: gt) that owns node N-1 and the downstream boundary row ...  (the non-team forms spelled out as they
  // always were: the flagship's code is sensitive, at the 1 % level, to how its constants come about - profiles/round4/README.md)
  const int tD = RAGGED ? NC / M : (kTeam ? G * T - 1 : T - 1);
  const int jD = kTail ? TAIL : (RAGGED ? NC - tD * M : M - 1);   // ... as its local node / row jD (0..M-1; tail-only form: fs_abi.hip picks the instantiation with TAIL = NC mod M)
  // tail-only form: 1.0 in the lanes up to / before the one that owns node N - 1, else 0.0 (multiplied in, never selected on)
  const R mle = (kTail && gt > tD) ? R(0) : R(1), mlt = (kTail && gt >= tD) ? R(0) : R(1);
  const size_t base = (size_t)reach * NS;

  Geo geo;
  geo.init(a, reach, N);
  // node terms of the lane's local node j (0..M); FS_REG_GEO: from section parameters loaded once per launch
  SecParams<R> secs[kRegGeo ? M + 1 : 1];
  if constexpr (kRegGeo) {
#pragma unroll
    for (int j = 0; j <= M; ++j) secs[j] = geo.section(min(s0 + j, N - 1));
  }
  // polyline nodes: the stage-table interval each of the lane's nodes was last evaluated in (Geometry::terms_hinted)
  constexpr bool kHinted = SEC == FS_SEC_IRREGULAR && M <= FS_POLY_HINT_MAXM && FS_POLY_HINT;
  int khint[kHinted ? M + 1 : 1];
  PolyBC<R> polybc[kHinted ? M + 1 : 1];      // (K, dK/dA, dA/dh) of the lane's nodes as last evaluated: the fused normal-depth row below
  if constexpr (kHinted) {
#pragma unroll
    for (int j = 0; j <= M; ++j) {
      khint[j] = geo.hint_init(min(s0 + j, N - 1));
      polybc[j].K = R(0); polybc[j].dKdA = R(0); polybc[j].dAdh = R(0);
    }
  }
  auto terms_at = [&](int j, R hh, R QQ) __attribute__((always_inline)) {
    if constexpr (kRegGeo) return node_terms_general(secs[j], hh, QQ);
    else if constexpr (kHinted) return geo.terms_hinted(min(s0 + j, N - 1), hh, QQ, khint[j], polybc[j]);
    else return geo.terms(min(s0 + j, N - 1), hh, QQ);
  };

  const bool own_scheme = BCK <= 0 && a.reach_scheme != nullptr;          // (the kernels compiled for one boundary pair are the benchmark shapes)
  const R th = own_scheme ? a.reach_scheme[reach] : a.theta, dt = own_scheme ? a.reach_scheme[(size_t)a.B + reach] : a.dt;
  const R dx_ = own_scheme ? a.reach_scheme[(size_t)2 * a.B + reach] : a.dx;
  // (tolerance and iteration cap of the reach's own run(), preissmann.py:101: read where they are used, so that the kernels compiled for one
  // boundary pair - own_scheme is false at compile time there - keep the code they had)
  auto tol_of = [&]() __attribute__((always_inline)) -> R { if constexpr (BCK <= 0) { if (own_scheme) return a.reach_scheme[(size_t)3 * a.B + reach]; } return a.tol; };
  auto max_iter_of = [&]() __attribute__((always_inline)) -> int { if constexpr (BCK <= 0) { if (own_scheme) return (int)a.reach_scheme[(size_t)4 * a.B + reach]; } return a.max_iter; };
  R r2dt = R(1) / (R(2) * dt);
  R cq = th / dx_;                            // theta/dx
  const R cqk = (R(1) - th) / dx_;            // (1-theta)/dx
  R hth = R(0.5) * th;
  const R hthk = R(0.5) * (R(1) - th);
  R g = R(kG);
  // scalings into the characteristic-like unknowns p = t dh + cq dQ, m = t dh - cq dQ (t = T/(2dt), fs_device.hpp):
  // 1/(2t) = dt/T, 1/(2cq), and the quotient of the two time / space weights k = (1/(2dt)) / (2cq)
  R i2c = R(0.5) / cq;
  R kap = r2dt * i2c;
  R dtcq = dt * cq;
  R hx = hth * i2c;                           // (theta/2) / (2cq)
  R ghth = g * hth;                           // the level constant kc2 carries the factor g as well: g Abar in one fma
  const R ghthk = g * hthk;
  R ghdt = g * hth * dt;

  // ---- unknowns of this lane: nodes s0 .. s0+M (clamped copies beyond the last node) ----
  R h[M + 1], Q[M + 1];
  R QoldD = R(0);                              // flow[k] at the last node (vol_in of the storage BC)

  // level-k constants of the 4-point stencil from the accepted state (h, Q) of level k:
  //   C = [sumA]/(2dt) + cq*dQ                 + kc0
  //   M = [sumQ]/(2dt) + cq*d(Q^2/A)           + kc1 + (g*hth*sumA + kc2)*(cq*dY + hth*sumSe + kc3)     (kc2 carries g)
  // Node s0 + M is lane + 1's node s0 (both lanes hold bitwise equal copies of its unknowns): with one wave per reach the
  // neighbour's terms arrive by a wave rotate.  Lane 63 receives lane 0's, which only a padding row or the downstream boundary
  // row ever looks at (and discards): the lane grid holds N rows, so lane 63's last row is never a cell, and its node M is a
  // clamped copy beyond the reach.  (Until round 3 lane 63 evaluated that node itself whenever the reach filled the wave - a
  // whole section evaluation per iteration, executed by 64 lanes for a result nobody read: a quarter of the polyline ensemble's
  // instructions.)
  constexpr bool kShareNode = FS_SHARE_NODE && W == 1 && !Geo::kConstT;
  auto last_node_terms = [&](const NodeTerms<R> &first, R hM, R QM) __attribute__((always_inline)) {
    auto rol = [](R v) { return dpp_mov<0x134>(v); };     // wave_rol:1
    NodeTerms<R> r;
    r.A = rol(first.A); r.T = rol(first.T); r.Se = rol(first.Se); r.eAT = rol(first.eAT); r.eQ = rol(first.eQ); r.v = rol(first.v);
    r.rT = rol(first.rT);
    (void)hM; (void)QM;
    return r;
  };
  auto write_level_constants = [&](const R(&hh)[M + 1], const R(&QQ)[M + 1]) __attribute__((always_inline)) {
    NodeTerms<R> L = terms_at(0, hh[0], QQ[0]);
    NodeTerms<R> Rlast;
    if (kShareNode) Rlast = last_node_terms(L, hh[M], QQ[M]);
#pragma unroll
    for (int c = 0; c < M; ++c) {
      const NodeTerms<R> Rn = (kShareNode && c == M - 1) ? Rlast : terms_at(c + 1, hh[c + 1], QQ[c + 1]);
      const R sumA = L.A + Rn.A;
      // explicit fmas only (no a*b + c left to the compiler's choice: see FS_PRIME)
      sm.kc[0][c][t] = fma_(cqk, QQ[c + 1] - QQ[c], -(sumA * r2dt));
      sm.kc[1][c][t] = fma_(cqk, fma_(QQ[c + 1], Rn.v, -(QQ[c] * L.v)), -((QQ[c + 1] + QQ[c]) * r2dt));
      sm.kc[2][c][t] = ghthk * sumA;
      sm.kc[3][c][t] = fma_(cqk, geo.bed_step(s0 + c) + (hh[c + 1] - hh[c]), hthk * (L.Se + Rn.Se));
      L = Rn;
#if FS_LEVEL_FENCE
      if ((c % FS_LEVEL_FENCE) == FS_LEVEL_FENCE - 1) __builtin_amdgcn_sched_barrier(0);
#endif
    }
  };
  auto save_terms = [&](int j, const NodeTerms<R> &nt) __attribute__((always_inline)) {
    if constexpr (kSaveTerms) { sm.nt[0][j][t] = nt.A; sm.nt[1][j][t] = nt.Se; sm.nt[2][j][t] = nt.v; }
  };
  auto level_constants_from_saved = [&](const R(&hh)[M + 1], const R(&QQ)[M + 1]) __attribute__((always_inline)) {
    if constexpr (kSaveTerms) {
      R A0 = sm.nt[0][0][t], Se0 = sm.nt[1][0][t], v0 = sm.nt[2][0][t];
#pragma unroll
      for (int c = 0; c < M; ++c) {
        const R A1 = sm.nt[0][c + 1][t], Se1 = sm.nt[1][c + 1][t], v1 = sm.nt[2][c + 1][t];
        const R sumA = A0 + A1;
        sm.kc[0][c][t] = fma_(cqk, QQ[c + 1] - QQ[c], -(sumA * r2dt));
        sm.kc[1][c][t] = fma_(cqk, fma_(QQ[c + 1], v1, -(QQ[c] * v0)), -((QQ[c + 1] + QQ[c]) * r2dt));
        sm.kc[2][c][t] = ghthk * sumA;
        sm.kc[3][c][t] = fma_(cqk, geo.bed_step(s0 + c) + (hh[c + 1] - hh[c]), hthk * (Se0 + Se1));
        A0 = A1; Se0 = Se1; v0 = v1;
      }
    }
  };

  // The launch starts with a priming pass: (h, Q) hold the accepted state of the entry level and run through the
  // acceptance block of the loop below once (level constants of the first level to solve, flow[k] of the storage row),
  // then they are replaced by the Newton start vector.  The level constants of every level - the first one of a launch
  // included - thus come from ONE instance of the code, and chunked stepping / a restart gives the bits of one launch.
  constexpr bool kPrime = FS_PRIME;
#pragma unroll
  for (int j = 0; j <= M; ++j) {
    const int node = min(s0 + j, N - 1);
    h[j] = kPrime ? a.hk[base + node] : a.hg[base + node];  Q[j] = kPrime ? a.Qk[base + node] : a.Qg[base + node];
  }
  // per-lane base pointers: every later access is base + immediate offset
  R *const hk_p = a.hk + base + s0, *const Qk_p = a.Qk + base + s0;
  R *const hg_p = a.hg + base + s0, *const Qg_p = a.Qg + base + s0;

  // Boundary descriptors of this reach: the parameters of the fixed-size kinds are copied to LDS once
  // (the boundary rows are evaluated every Newton iteration on the critical path of the first and the
  // last wave; a global load there costs more than the row itself), the hydrograph value of a level
  // is fetched when the level starts.
  BCDesc<R> usd = a.us, dsd = a.ds;
  if (BCK <= 0 && a.reach_kinds) { usd.kind = a.reach_kinds[reach]; dsd.kind = a.reach_kinds[(size_t)a.B + reach]; }
  if (t < 2 * FS_BC_MAX_PARAMS) {
    const int side = t / FS_BC_MAX_PARAMS, i = t - side * FS_BC_MAX_PARAMS;
    // (a reference and the kind next to it, not a copy of the descriptor with the kind patched in: the copy cost the flagship
    // kernel 1.2 % - seven more scalar loads inside its loops and 18 more AGPR moves - for code it never executes)
    const BCDesc<R> &src = side ? a.ds : a.us;
    const int skind = side ? dsd.kind : usd.kind;
    static constexpr int kCount[] = {0, 1, 1, 2, 4, 5, 10, 5};
    if (skind <= FS_BC_STORAGE && i < kCount[skind]) sm.bcp[side][i] = bc_param(src, i, reach, a.B);
    if (skind == FS_BC_NORMAL_DEPTH && i == 2) {        // derived: sign(S0) sqrt|S0| (hydraulics.py:4-13)
      const R S0 = bc_param(src, 0, reach, a.B);
      sm.bcp[side][2] = (S0 < R(0) ? R(-1) : R(1)) * sqrt_(fabs_(S0));
    }
  }
  if (usd.kind <= FS_BC_STORAGE) { usd.params = &sm.bcp[0][0]; usd.stride = 0; }
  if (dsd.kind <= FS_BC_STORAGE) { dsd.params = &sm.bcp[1][0]; dsd.stride = 0; }

  // Normal depth at a polyline node N-1 whose boundary bed level is the section's own z_min (boundary.py:80, :161-181 then
  // evaluate residual and derivative at the same stage): the row is K, dK/dA, dA/dh of node N-1 at the iterate - exactly what
  // the fold's evaluation of that node has just produced.  The row is then written in the fold, from that evaluation, and the
  // boundary's own section evaluation (a whole wave waiting on one lane: three dependent fetches and ~600 instructions per
  // Newton iteration, a third of the polyline ensemble's time) is not made.
  bool fuseD = false;
  R fuse_sg = R(1), fuse_rt = R(0);
  if constexpr (kHinted && !kFlatBC) {
    __syncthreads();                                    // (sm.bcp written above)
    if (pinned<BCK, 1>(dsd).kind == FS_BC_NORMAL_DEPTH && geo.hint_init(N - 1) != -2 && sm.bcp[1][1] == geo.bed(N - 1)) {
      fuseD = true;
      const R S0 = sm.bcp[1][0];
      fuse_sg = S0 < R(0) ? R(-1) : R(1); fuse_rt = sqrt_(fabs_(S0));
    }
  }
  // a compile-time constant in the kernels compiled for a boundary pair
  const bool ds_storage = BCK >= 2 ? bc_is_storage(BCK - 2) : bc_is_storage(dsd.kind);
  R Yprev = (ds_storage && gt == tD) ? a.Yprev[reach] : R(0);
  int status = a.status[reach];
  // FS_ILL_CONDITIONED is a warning that sticks to the reach, not a failure: the run goes on (and a later launch finds it here)
  constexpr bool kMonitor = DIAG || FS_MONITOR_ALL;
  bool warn = status == FS_ILL_CONDITIONED;
  if (warn) status = FS_OK;
  bool primed = false;                        // (h, Q) hold the Newton vector (after the priming pass), not the entry state
  int parity = 0;
  int exchanges = 0;                          // team form: exchanges made in this launch (the reach's counter stands at exchanges G after each)
  if (t == 0) { sm.xflag[0] = 0; sm.xflag[1] = 0; }
  __syncthreads();

  if (!kPrime) {   // accepted state of the entry level -> 4 constants per cell in LDS (a second instance of that code)
    R hk[M + 1], Qk[M + 1];
#pragma unroll
    for (int j = 0; j <= M; ++j) {
      const int node = min(s0 + j, N - 1);
      hk[j] = a.hk[base + node]; Qk[j] = a.Qk[base + node];
    }
    if (gt == tD) {
#pragma unroll
      for (int j = RAGGED ? 0 : M - 1; j < M; ++j) if (j == jD) QoldD = Qk[j];
    }
    write_level_constants(hk, Qk);
    primed = true;
  }
#ifdef FS_STAMP
  unsigned long long stamp_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev_ = __builtin_amdgcn_s_memtime();
#endif
  for (int step = kPrime ? -1 : 0; step < a.n_steps && status == FS_OK; ++step) {
    int prime = kPrime && step < 0;             // the priming pass (opaque to the optimiser: one loop body, not two)
    if (kPrime) asm volatile("" : "+s"(prime));
    const int level = a.level0 + step + 1;
    if (!prime) {
      if (usd.target) usd.tgt = usd.target[(size_t)level * a.B + reach];
      if (dsd.target) dsd.tgt = dsd.target[(size_t)level * a.B + reach];
    }
    int it = kBudget ? it_entry : 0;
    int budget = (kBudget && a.iter_budget > 0) ? a.iter_budget : 0x7fffffff;
    bool converged = false;
    R Ynew = Yprev;
    while (!converged && status == FS_OK) {
      R dh[M + 1], dQ[M + 1];                   // the update, pending until the acceptance block is through
      if (prime) {
        converged = true;
      } else {
      if (kBudget && budget-- <= 0) break;
      ++it;
      if constexpr (BCK <= 0) { if (it - 1 >= max_iter_of()) { status = FS_MAX_ITER; break; } }
      else
      if (it - 1 >= a.max_iter) { status = FS_MAX_ITER; break; }     // preissmann.py:124-126
      parity ^= 1;
      bool grow = false;                        // conditioning monitor: this iteration's segments grew past the limit
      FS_T(7);

      // opaque lane offset (an integer, so the accesses stay LDS ds_read, not flat): the 4*M level
      // constants are Newton-loop invariants and would otherwise be hoisted into registers
      int kco = t;
      asm volatile("" : "+v"(kco));
      // the same for the lane number the tree-slot addresses derive from: as loop invariants the addresses
      // are hoisted and spilled to scratch, and every reload is a full memory round trip inside the tree
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const R *kcb = &sm.kc[0][0][0] + kco;
      // second half of the constants from a base of its own: beyond 64 KB a ds_read has no immediate offset left, and every
      // one of those loads would come with an address add (32 per iteration in the 4096-node kernels)
      int kco2 = kco + 2 * M * T;
      if constexpr (4 * M * T * sizeof(R) > 65536) asm volatile("" : "+v"(kco2));
      const R *kcb2 = &sm.kc[0][0][0] + kco2;
#ifdef FS_DUMPKC
      if (it == 1 && a.dbg && t == (reach & 63) && level < 12) {      // diagnostic builds: what the first iteration of a level starts from
        unsigned long long *o = a.dbg + ((size_t)reach * 16 + 4 + level) * 12;
        for (int i = 0; i < 4; ++i) o[i] = __double_as_longlong((double)sm.kc[i][0][t]);
        for (int i = 0; i < 4; ++i) o[4 + i] = __double_as_longlong((double)sm.kc[i][M - 1][t]);
        o[8] = __double_as_longlong((double)h[0]); o[9] = __double_as_longlong((double)Q[0]);
        o[10] = __double_as_longlong((double)h[M]); o[11] = __double_as_longlong((double)Q[M]);
      }
#endif

      // ================= 1. boundary rows (boundary.py:56-242) =================
      // lane 0: the upstream row on (dh_0, dQ_0); the lane of node N-1: the downstream row, which is row N-1 of the
      // scalar system and enters that lane's fold below like any other row
      BCRow<R> Urow, Drow;
      Drow.dh = R(1); Drow.dq = R(0); Drow.res = R(0);
      R nrm2 = R(0);
      int myflag = 0;                          // team form: the flag a storage row raised in this lane (the only value: FS_STORAGE_RANGE)
      if constexpr (kFlatBC) {
        // Boundary kinds fixed at compile time: EVERY lane evaluates both rows, on its own numbers, as straight-line code;
        // only lane 0 / the lane of node N-1 keep what comes out.  A wave executes a divergent branch for one lane at the
        // price of all 64 anyway - but a branch is a block of its own, and the row's dependent chain (a pow, a conveyance)
        // then runs alone at dependent-issue latency ahead of the fold; as straight-line code it is scheduled into the first
        // cell's independent work.  Pays only in the one-wave-per-SIMD trapezoid kernel (C5 fp64 +2 %), see kFlatBC.
        R dummy = R(0), YnewAll = Yprev;
        int fU = 0, fD = 0;
        Urow = geo.template boundary<BCK, 0>(usd, reach, a.B, level, 0, h[0], Q[0], R(0), dt, R(0), &dummy, &fU);
        R hD = h[0], QD = Q[0];
#pragma unroll
        for (int j = RAGGED ? 1 : M - 1; j < M; ++j) if (j == jD) { hD = h[j]; QD = Q[j]; }
        Drow = geo.template boundary<BCK, 1>(dsd, reach, a.B, level, N - 1, hD, QD, QoldD, dt, Yprev, &YnewAll, &fD);
        const bool isD = gt == tD;
        nrm2 = (gt == 0 ? Urow.res * Urow.res : R(0)) + (isD ? Drow.res * Drow.res : R(0));
        if (isD) {
          Ynew = YnewAll;
          if (fD) sm.xflag[parity] = fD;
        }
      } else {
        if (gt == 0) {
          R dummy; int flag = 0;
          Urow = geo.template boundary<BCK, 0>(usd, reach, a.B, level, 0, h[0], Q[0], R(0), dt, R(0), &dummy, &flag);
          nrm2 = Urow.res * Urow.res;
        }
        if (gt == tD && !fuseD) {
          R hD = h[0], QD = Q[0];
          int flag = 0;
#pragma unroll
          for (int j = RAGGED ? 1 : M - 1; j < M; ++j) if (j == jD) { hD = h[j]; QD = Q[j]; }
          Drow = geo.template boundary<BCK, 1>(dsd, reach, a.B, level, N - 1, hD, QD, QoldD, dt, Yprev, &Ynew, &flag);
          nrm2 += Drow.res * Drow.res;
          if (flag) sm.xflag[parity] = flag;
          if (kTeam && flag) myflag = flag;
        }
      }
      FS_T(1);

      // ================= 2. local assembly + fold (registers only) =================
      LocalElim<R> el[M - 1];
      R iTn[Geo::kConstT ? 1 : M + 1];           // dt / T = 1/(2t) of the nodes, only when the top width varies
      Seg<R> seg;                                 // running rows of the lane's segment
      seg.u1 = R(0); seg.u3 = R(-1); seg.ru = R(0);
      R upU1 = R(0), upU3 = R(0), upRu = R(0);    // the lane's finished up row, kept for the way back
      R rcLast = R(0);                            // rc of the lane's last row (links node M: p_M = rc - m_{M-1})
      constexpr bool kW1Lanes = FS_W1_LANES && W == 1 && !kTeam && sizeof(R) == 8;
      R xb0 = R(0), xb1 = R(0), xb2 = R(0);      // kW1Lanes: the upstream row, valid in lane 0
      {
        NodeTerms<R> L = terms_at(0, h[0], Q[0]);
        NodeTerms<R> Rlast;
        if (kShareNode) Rlast = last_node_terms(L, h[M], Q[M]);
        save_terms(0, L);
        R i2tL = dt * L.rT;
        if (!Geo::kConstT) iTn[0] = i2tL;
        if (gt == 0) {                          // upstream row on (p_0, m_0): aU p_0 + bU m_0 = -res
          const R x = Urow.dh * i2tL, y = Urow.dq * i2c;
          if constexpr (kTeam) {                // (a team's: into the mailbox, next to the segments)
            if constexpr (kTagged) {
              const unsigned o = (unsigned)(G * W) * kTeamWords * 16u;
              const unsigned long long tg = ((unsigned long long)a.team_epoch << 32) | (unsigned)(exchanges + 1);
              const __amdgpu_buffer_rsrc_t mb = mbox(parity);
              team_put2(mb, o, __double_as_longlong((double)(x + y)), tg); team_put2(mb, o + 16, __double_as_longlong((double)(x - y)), tg);
              team_put2(mb, o + 32, __double_as_longlong((double)(-Urow.res)), tg);
            } else {
            R *q = a.team_mail + (((size_t)reach * 2 + parity) * (G * W + 1) + G * W) * kTeamWords;
            team_put(q + 0, x + y); team_put(q + 1, x - y); team_put(q + 2, -Urow.res);
            }
          } else if constexpr (kW1Lanes) {
            xb0 = x + y; xb1 = x - y; xb2 = -Urow.res;
          } else {
          sm.xbc[parity][0] = x + y; sm.xbc[parity][1] = x - y; sm.xbc[parity][2] = -Urow.res;
          }
        }
        R rcPrev = R(0);
#pragma unroll
        for (int c = 0; c < M; ++c) {
          const R k0 = kcb[(0 * M + c) * T], k1 = kcb[(1 * M + c) * T], k2 = kcb2[(0 * M + c) * T], k3 = kcb2[(1 * M + c) * T];
          const NodeTerms<R> Rn = (kShareNode && c == M - 1) ? Rlast : terms_at(c + 1, h[c + 1], Q[c + 1]);
          save_terms(c + 1, Rn);
          const R i2tR = dt * Rn.rT;
          if (!Geo::kConstT) iTn[c + 1] = i2tR;
          Row<R> row;
          {
            const R sumA = L.A + Rn.A;
            const R Cres = sumA * r2dt + cq * (Q[c + 1] - Q[c]) + k0;                            // :220-249
            const R gA = fma_(ghth, sumA, k2);                                                    // g Abar
            const R S = cq * (geo.bed_step(s0 + c) + (h[c + 1] - h[c])) + hth * (L.Se + Rn.Se) + k3;
            const R Mres = (Q[c + 1] + Q[c]) * r2dt + cq * (Q[c + 1] * Rn.v - Q[c] * L.v) + k1 + gA * S;   // :251-301
            // momentum entries (preissmann.py:496-733) scaled by 1/(2t) of their node (dh) and 1/(2cq) (dQ):
            //   X0 = pm0 dt/T0, Y0 = pm1/(2cq), X1 = sm0 dt/T1, Y1 = sm1/(2cq)
            const R gAdt = gA * dt, gAx = gA * hx, sdt = ghdt * S;
            const R X0 = fma_(gAdt, fma_(hth, L.eAT, -(cq * L.rT)), fma_(dtcq, L.v * L.v, sdt));      // :558-612
            const R X1 = fma_(gAdt, fma_(hth, Rn.eAT, cq * Rn.rT), fma_(-dtcq, Rn.v * Rn.v, sdt));    // :496-550
            const R Y0 = fma_(gAx, L.eQ, kap - L.v);                                                 // :677-733
            const R Y1 = fma_(gAx, Rn.eQ, kap + Rn.v);                                               // :619-675
            const R ga = X1 + Y1;
            row.al = X0 + Y0; row.D = (X0 - Y0) - ga; row.de = X1 - Y1;
            row.rho0 = fma_(ga, Cres, -Mres);             // qm - ga rc with qm = -Mres, rc = -Cres
            row.rc = -Cres;
            R r2 = fma_(Cres, Cres, Mres * Mres);
            if constexpr (kTail) {
              if (c == TAIL) {                 // the boundary row, in the one lane that owns node N - 1 (its residual was added in step 1)
                const bool isD = gt == tD;
                const R x = Drow.dh * i2tL, y = Drow.dq * i2c;
                row.al = isD ? x + y : row.al; row.D = isD ? x - y : row.D; row.de = isD ? R(0) : row.de;
                row.rho0 = isD ? -Drow.res : row.rho0; row.rc = isD ? R(0) : row.rc;
              }
              nrm2 = fma_(r2, c < TAIL ? mle : mlt, nrm2);    // cells only: not the boundary row, not the phantom cells behind it
            } else
            // rows beyond the cells: the downstream boundary row (on p and m of node N-1), then identity rows
            if (RAGGED || c == M - 1) {
              const int k = s0 + c;
              const bool cell = RAGGED ? (k < NC) : (kTeam ? gt != G * T - 1 : t != T - 1);      // (k = s0 + c counts along the whole reach)
              const bool bcr = RAGGED ? (k == NC) : true;
              if (!cell) {
                BCRow<R> Dr = Drow;
                r2 = R(0);
                if constexpr (kHinted) {
                  if (fuseD) {                 // normal depth from the fold's own evaluation of node N-1 (= the row's left node)
                    Dr = normal_depth_row_poly(fuse_sg, fuse_rt, polybc[c].K, polybc[c].dKdA, polybc[c].dAdh, Q[c]);
                    r2 = bcr ? Dr.res * Dr.res : R(0);
                  }
                }
                const R x = Dr.dh * i2tL, y = Dr.dq * i2c;
                row.al = bcr ? x + y : R(0); row.D = bcr ? x - y : R(1); row.de = R(0);
                row.rho0 = bcr ? -Dr.res : R(0); row.rc = R(0);
              }
            }
            if constexpr (!kTail) nrm2 += r2;
          }
          if (c == 0) {
            seg.d1 = row.al; seg.d2 = row.D; seg.d3 = row.de; seg.rd = row.rho0;
          } else {
            // forward elimination of m_{c-1} (pivot: the running down row), the up row follows (fs_device.hpp)
            const R r = frcp(seg.d2);
            const R R1 = seg.d1 * r, R2 = seg.d3 * r, R3 = seg.rd * r;
            LocalElim<R> &e = el[c - 1];
            e.R1.put(R1); e.R2.put(R2); e.R3.put(R3);
            if (!Geo::kConstT) e.qc.put(rcPrev);
            const R rho = fma_(-row.al, rcPrev, row.rho0);
            seg.d1 = row.al * R1; seg.d2 = fma_(row.al, R2, row.D); seg.d3 = row.de; seg.rd = fma_(row.al, R3, rho);
            seg.u1 = fma_(-seg.u3, R1, seg.u1); seg.ru = fma_(-seg.u3, R3, seg.ru); seg.u3 = -(seg.u3 * R2);
          }
          rcPrev = row.rc;
          // the running rows must exist here: keeps row c's elimination inside cell c's scheduling region
          // (long chunks only: with M <= 4 the cells' node terms interleave profitably, measured on C4)
          if constexpr (M >= 8 && (sizeof(R) == 4 ? FS_PIN_SEG_F32 : FS_PIN_SEG) != 0)
            asm volatile("" :: "v"(seg.u1), "v"(seg.u3), "v"(seg.ru), "v"(seg.d1), "v"(seg.d2), "v"(seg.d3), "v"(seg.rd), "v"(rcPrev));
          L = Rn; i2tL = i2tR;
#if FS_CELL_FENCE
          if ((c % FS_CELL_FENCE) == FS_CELL_FENCE - 1) __builtin_amdgcn_sched_barrier(0);
#endif
        }
        seg.rc = rcPrev; rcLast = rcPrev;
        upU1 = seg.u1; upU3 = seg.u3; upRu = seg.ru;
      }
      int gi = 0;                                  // conditioning monitor: largest |u3| of the segments this lane has seen
      if constexpr (kMonitor) gi = hi_abs(seg.u3);

      FS_T(0);
#if FS_PHASE_FENCE & 1
      __builtin_amdgcn_sched_barrier(0);   // phases are not interleaved: it only costs registers (measured around the down-sweep: +5 %)
#endif
      // ================= 3. in-wave tree (up-sweep) =================
      constexpr bool kTreeRegs = FS_TREE_REGS && W > 1 && M >= 8 && !DIAG;
      Elim<R> rec[kTreeRegs ? 6 : 1];
      auto up_level = [&](auto lc) __attribute__((always_inline)) {
        constexpr int l = decltype(lc)::value;
        constexpr int d = 1 << l;
        const Seg<R> left = seg_from_below<d>(seg);
        Seg<R> mg; Elim<R> e;
        merge(left, seg, mg, e);
        if constexpr (kTreeRegs) {
          rec[l] = e;
        } else if ((lane & (2 * d - 1)) == (2 * d - 1)) {
          const int slot = (64 - (64 >> l)) + (ln >> (l + 1));
          R *p = &sm.tree[wave][0][slot];
          p[0 * 64] = e.A1; p[1 * 64] = e.A2; p[2 * 64] = e.A3; p[3 * 64] = e.rc;
        }
        seg = mg;      // in every lane: a lane that does not survive this level is not read again (no select, no branch around the merge)
        if constexpr (kMonitor) gi = max_(max_(tree_from_below<d>(gi), gi), hi_abs(mg.u3));
      };
      up_level(std::integral_constant<int, 0>{}); up_level(std::integral_constant<int, 1>{});
      up_level(std::integral_constant<int, 2>{}); up_level(std::integral_constant<int, 3>{});
      up_level(std::integral_constant<int, 4>{}); up_level(std::integral_constant<int, 5>{});
#if FS_PHASE_FENCE & 1
      __builtin_amdgcn_sched_barrier(0);
#endif
      nrm2 = wave_sum(nrm2);
      if constexpr (kTeam) {
        const int wflag = __builtin_amdgcn_ballot_w64(myflag != 0) != 0 ? (int)FS_STORAGE_RANGE : 0;      // raised by a lane of this wave
        if constexpr (kTagged) {
          if (lane == 63) {
            const unsigned o = (unsigned)(member * W + wave) * kTeamWords * 16u;
            const unsigned long long tg = ((unsigned long long)a.team_epoch << 32) | (unsigned)(exchanges + 1);
            const __amdgpu_buffer_rsrc_t mb = mbox(parity);
            auto put = [&](int i, R v) __attribute__((always_inline)) { team_put2(mb, o + 16u * i, __double_as_longlong((double)v), tg); };
            put(0, seg.u1); put(1, seg.u3); put(2, seg.ru); put(3, seg.d1); put(4, seg.d2); put(5, seg.d3); put(6, seg.rd); put(7, seg.rc); put(8, nrm2);
            team_put2(mb, o + 16u * 9, ((unsigned long long)(unsigned)wflag << 32) | (unsigned)gi, tg);
          }
        } else {
        if (lane == 63) {                    // the wave's slot of the team's mailbox (device memory; published by the release below)
          R *p = a.team_mail + (((size_t)reach * 2 + parity) * (G * W + 1) + (member * W + wave)) * kTeamWords;
          team_put(p + 0, seg.u1); team_put(p + 1, seg.u3); team_put(p + 2, seg.ru); team_put(p + 3, seg.d1); team_put(p + 4, seg.d2);
          team_put(p + 5, seg.d3); team_put(p + 6, seg.rd); team_put(p + 7, seg.rc); team_put(p + 8, nrm2);
          int32_t *pi = reinterpret_cast<int32_t *>(p + 9);
          team_put(pi, gi); team_put(pi + 1, wflag);
        }
        team_posted();
        }
      } else if constexpr (kW1Lanes) {
        // nothing to post: the root segment is read from lane 63 below
      } else
      if (lane == 63) {
        R *p = sm.xseg[parity][wave];
        p[0] = seg.u1; p[1] = seg.u3; p[2] = seg.ru; p[3] = seg.d1; p[4] = seg.d2; p[5] = seg.d3; p[6] = seg.rd; p[7] = seg.rc;
        sm.xnorm[parity][wave] = nrm2;
        if constexpr (kMonitor) sm.xg[parity][wave] = gi;
      }
      FS_T(2);
      // the wave's own top tree records for the way down (step 5), requested ahead of the barrier: their LDS latency passes
      // during the cross-wave step instead of after it (FS_PREFETCH_DOWN levels; the wave wrote them itself, in order)
      auto load_rec_early = [&](auto lc) __attribute__((always_inline)) {
        constexpr int l = decltype(lc)::value;
        Elim<R> e;
        const int slot = (64 - (64 >> l)) + (ln >> (l + 1));
        const R *p = &sm.tree[wave][0][slot];
        e.A1 = p[0 * 64]; e.A2 = p[1 * 64]; e.A3 = p[2 * 64]; e.rc = p[3 * 64];
        return e;
      };
      constexpr int kPre = (W > 1 && M >= 8 && !DIAG && !kTreeRegs) ? FS_PREFETCH_DOWN : 0;
      Elim<R> pre5, pre4, pre3;
      if constexpr (kPre >= 1) pre5 = load_rec_early(std::integral_constant<int, 5>{});
      if constexpr (kPre >= 2) pre4 = load_rec_early(std::integral_constant<int, 4>{});
      if constexpr (kPre >= 3) pre3 = load_rec_early(std::integral_constant<int, 3>{});
      // (the tagged team form could do without this barrier - wave 0 polls the mailbox for every wave's post, its own workgroup's included - and is
      // 6 % SLOWER without it: a wave 0 that starts polling while its neighbours still fold takes their issue slots and their memory path)
      // (one-wave reaches: no s_barrier is emitted for 64 threads, and dropping the workgroup fence with it changes nothing - measured, C4 / C5 / polylines +-0.3 %)
      __syncthreads();
      FS_T(3);
      // rc of the lane's rows for the back-substitution (step 5) does not depend on the solve: computed here, its instructions
      // fill the waits of the cross-wave step (a chain of dependent merges and reciprocals) instead of lengthening step 5
      constexpr bool kRcEarly = FS_RC_EARLY && Geo::kConstT && FS_LAUNDER_BACK && W > 1 && M >= 8 && !DIAG && !RAGGED;
      R rcE[kRcEarly ? M - 1 : 1];
      if constexpr (kRcEarly) {
        asm volatile("" : "+v"(kco));
        kcb = &sm.kc[0][0][0] + kco;
#pragma unroll
        for (int j = 0; j <= M; ++j) asm volatile("" : "+v"(h[j]), "+v"(Q[j]));
#pragma unroll
        for (int j = 0; j + 1 < M; ++j)
          rcE[j] = -(geo.terms_T() * (h[j] + h[j + 1]) * r2dt + cq * (Q[j + 1] - Q[j]) + kcb[(0 * M + j) * T]);
      }

      // ================= 4. across waves: fold, close with the upstream row, unfold =================
#if FS_XWAVE_FENCE & 1
      __builtin_amdgcn_sched_barrier(0);
#endif
      R tot = R(0);
      R pL, mR;                        // p of this wave's first row, m of its last one
      R mAw = R(0), mBw = R(0);        // m of this wave's first row / of the next wave's first row (shared nodes, below)
      if constexpr (kTeam) {
        // ---- the team's exchange: where one workgroup has a barrier, G workgroups have a counter in device memory ----
        // (every poster has seen its stores acknowledged before the __syncthreads above, thread 0 counts the workgroup in after it; the
        // counter of a reach only grows: the e-th exchange of the launch is complete when it reaches e G)
        ++exchanges;
        if (!kTagged && t == 0) {
          unsigned long long *cnt = a.team_sync + 1 + reach;
          __hip_atomic_fetch_add(cnt, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned long long want = (unsigned long long)exchanges * (unsigned long long)G;
          const unsigned long long t_in = __builtin_amdgcn_s_memtime();
          int stall = 0;
          // (relaxed polls: an acquire per poll would invalidate the caches of the whole XCD on every turn; one acquire fence follows the wait)
          while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memtime() - t_in > kTeamPatience) { stall = 1; break; }     // give the reach up, do not spin on
          }
          sm.xstall = stall;
        }
        FS_T(11);
        if constexpr (!kTagged) __syncthreads();
        if (wave == 0) {
          // the top tree over the S = G W posted segments, one per lane, identity segments beyond (the multi-pass kernel's, fs_long.hpp)
          const int S = G * W;
          Seg<R> xs;
          R nr, aU, bU, rU;
          int gx, fl;
          if constexpr (kTagged) {
            // lane s polls the ten words of slot s, every lane the three of the upstream row, until all carry this exchange's tag
            const unsigned long long want = ((unsigned long long)a.team_epoch << 32) | (unsigned)exchanges;
            const __amdgpu_buffer_rsrc_t mb = mbox(parity);
            const unsigned o = (unsigned)(lane < S ? lane : 0) * kTeamWords * 16u, ou = (unsigned)S * kTeamWords * 16u;
            const unsigned long long t_in = __builtin_amdgcn_s_memtime();
            fs_u4 w[13];
            int stall = 0;
            for (;;) {
#pragma unroll
              for (int i = 0; i < 10; ++i) w[i] = team_get2(mb, o + 16u * i);
#pragma unroll
              for (int i = 0; i < 3; ++i) w[10 + i] = team_get2(mb, ou + 16u * i);
              bool ok = true;
#pragma unroll
              for (int i = 0; i < 13; ++i) ok = ok && team_tag(w[i]) == want;
              if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
              __builtin_amdgcn_s_sleep(FS_TEAM_SLEEP);
              if (__builtin_amdgcn_s_memtime() - t_in > kTeamPatience) { stall = 1; break; }     // give the reach up, do not spin on
            }
            if (lane == 0) sm.xstall = stall;
            auto val = [&](int i) { return (R)__longlong_as_double((long long)team_bits(w[i])); };
            xs.u1 = val(0); xs.u3 = val(1); xs.ru = val(2); xs.d1 = val(3); xs.d2 = val(4); xs.d3 = val(5); xs.rd = val(6); xs.rc = val(7);
            nr = val(8);
            gx = (int)(unsigned)team_bits(w[9]); fl = (int)(unsigned)(team_bits(w[9]) >> 32);
            aU = val(10); bU = val(11); rU = val(12);
          } else {
          const R *mail = a.team_mail + ((size_t)reach * 2 + parity) * (S + 1) * kTeamWords;
          const R *q = mail + (size_t)(lane < S ? lane : 0) * kTeamWords;
          auto ld = [&](int i) { return __hip_atomic_load(q + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
          xs.u1 = ld(0); xs.u3 = ld(1); xs.ru = ld(2); xs.d1 = ld(3); xs.d2 = ld(4); xs.d3 = ld(5); xs.rd = ld(6); xs.rc = ld(7);
          nr = ld(8);
          const int32_t *qi = reinterpret_cast<const int32_t *>(q + 9);
          gx = __hip_atomic_load(qi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          fl = __hip_atomic_load(qi + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          aU = __hip_atomic_load(mail + (size_t)S * kTeamWords + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bU = __hip_atomic_load(mail + (size_t)S * kTeamWords + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          rU = __hip_atomic_load(mail + (size_t)S * kTeamWords + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if (lane >= S) { xs.u1 = R(0); xs.u3 = R(0); xs.ru = R(0); xs.d1 = R(0); xs.d2 = R(1); xs.d3 = R(0); xs.rd = R(0); xs.rc = R(0); nr = R(0); gx = 0; fl = 0; }
          const R u1o = xs.u1, u3o = xs.u3, ruo = xs.ru;
          auto xup = [&](auto lc) __attribute__((always_inline)) {
            constexpr int l = decltype(lc)::value;
            constexpr int d = 1 << l;
            const Seg<R> left = seg_from_below<d>(xs);
            Seg<R> mg; Elim<R> e;
            merge(left, xs, mg, e);
            if ((lane & (2 * d - 1)) == (2 * d - 1)) {
              R *w = &sm.xtree[0][(64 - (64 >> l)) + (lane >> (l + 1))];
              w[0 * 64] = e.A1; w[1 * 64] = e.A2; w[2 * 64] = e.A3; w[3 * 64] = e.rc;
            }
            xs = mg;
            gx = max_(max_(tree_from_below<d>(gx), gx), hi_abs(mg.u3));
            fl = max_(tree_from_below<d>(fl), fl);
          };
          // (levels that would only merge identity segments are skipped: S <= 2^nl lanes carry something; the root then sits in lane 2^nl - 1)
          const int nl = S <= 8 ? 3 : (S <= 16 ? 4 : (S <= 32 ? 5 : 6));
          xup(std::integral_constant<int, 0>{}); xup(std::integral_constant<int, 1>{}); xup(std::integral_constant<int, 2>{});
          if (nl > 3) xup(std::integral_constant<int, 3>{});
          if (nl > 4) xup(std::integral_constant<int, 4>{});
          if (nl > 5) xup(std::integral_constant<int, 5>{});
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          R p0, m0, ml;
          close_root(xs, aU, bU, rU, p0, m0, ml);            // valid in the root's lane
          const int rl = (1 << nl) - 1;
          R px = read_lane(p0, rl), mx = read_lane(ml, rl);
          auto xdown = [&](auto lc) __attribute__((always_inline)) {
            constexpr int l = decltype(lc)::value;
            const R *w = &sm.xtree[0][(64 - (64 >> l)) + (lane >> (l + 1))];
            Elim<R> e;
            e.A1 = w[0 * 64]; e.A2 = w[1 * 64]; e.A3 = w[2 * 64]; e.rc = w[3 * 64];
            const R sep = separator(e, px, mx);
            const bool upper = ((lane >> l) & 1) != 0;
            px = upper ? e.rc - sep : px;
            mx = upper ? mx : sep;
          };
          if (nl > 5) xdown(std::integral_constant<int, 5>{});
          if (nl > 4) xdown(std::integral_constant<int, 4>{});
          if (nl > 3) xdown(std::integral_constant<int, 3>{});
          xdown(std::integral_constant<int, 2>{}); xdown(std::integral_constant<int, 1>{}); xdown(std::integral_constant<int, 0>{});
          const R ma = fma_(-u1o, px, fma_(-u3o, mx, ruo));
          R mb = dpp_mov<0x134>(ma);                // wave_rol:1 : m of the next segment's first row
          if (lane == 63) mb = R(0);
          if (lane < S) { R *o = sm.xres[lane]; o[0] = px; o[1] = mx; o[2] = ma; o[3] = (lane == S - 1) ? R(0) : mb; }
          const R tsum = wave_sum(nr);               // (a fixed order: the same total in every member)
          // (cross-lane reads in uniform control flow, fs_long.hpp)
          const int gtop = __builtin_amdgcn_readlane(gx, rl), ftop = __builtin_amdgcn_readlane(fl, rl);
          if (lane == 0) { sm.xtot = tsum; sm.xwarn = gtop > growth_limit_bits<R>() ? 1 : 0; sm.xflagT = ftop; }
        }
        __syncthreads();
        tot = sm.xtot;
        {
          const R *xr = sm.xres[member * W + wave];
          pL = xr[0]; mR = xr[1]; mAw = xr[2]; mBw = xr[3];
        }
        if constexpr (kMonitor) { if (sm.xwarn != 0) grow = true; }
        if (sm.xflagT != 0) status = sm.xflagT;
        if (sm.xstall != 0) status = FS_TEAM_STALL;
      } else if constexpr (kW1Lanes) {
        // One wave: the root segment sits in lane 63, the upstream row in lane 0.  Lane 63 closes the system on its own registers (every
        // lane executes it, lane 63's result is the one read) - the arithmetic of the general path below, without its LDS round trip.
        tot = nrm2;                                    // (wave_sum left the total in every lane)
        const R aU = read_lane(xb0, 0), bU = read_lane(xb1, 0), rU = read_lane(xb2, 0);
        R p0, m0, ml;
        close_root(seg, aU, bU, rU, p0, m0, ml);
        pL = read_lane(p0, 63); mR = read_lane(ml, 63);
        if constexpr (kMonitor) { if (__builtin_amdgcn_readlane(gi, 63) > growth_limit_bits<R>()) grow = true; }
      } else if constexpr (W > 1 && W >= FS_XLANES_MINW) {
        // A second, small tree over the W wave segments, one segment per LANE: lane w of every wave takes the segment of
        // wave w and the W - 1 merges run as log2 W DPP levels (row_shr:1/2/4) exactly like the in-wave tree - the same
        // merges in the same order as the per-thread fold below, so the bits do not change - with 8 live numbers per lane
        // instead of 16 W.  Every wave does it redundantly on its own copy of the records (no second barrier).
        constexpr int LW = W == 2 ? 1 : (W == 4 ? 2 : 3);
        static_assert(W == 2 || W == 4 || W == 8, "cross-wave tree: 2, 4 or 8 waves per reach");
        const int wl = ln & (W - 1);
        const R *ps = sm.xseg[parity][wl];
        Seg<R> xs;
        xs.u1 = ps[0]; xs.u3 = ps[1]; xs.ru = ps[2]; xs.d1 = ps[3]; xs.d2 = ps[4]; xs.d3 = ps[5]; xs.rd = ps[6]; xs.rc = ps[7];
        const R u1o = xs.u1, u3o = xs.u3, ruo = xs.ru;       // the wave's own up row (m of its first row, below)
#pragma unroll
        for (int w = 0; w < W; ++w) tot += sm.xnorm[parity][w];
        int gx = 0;
        if constexpr (kMonitor) gx = sm.xg[parity][wl];
        // Records stay in registers (valid in the lane that survives its level: low l + 1 bits set); on the way down the
        // group's record comes from the group's last lane by a quad permute (groups of 2 and 4 lanes) or a readlane (8).
        Elim<R> xe[LW];
        auto xup = [&](auto lc) __attribute__((always_inline)) {
          constexpr int l = decltype(lc)::value;
          constexpr int d = 1 << l;
          if constexpr (l < LW) {
            const Seg<R> left = seg_from_below<d>(xs);
            Seg<R> mg;
            merge(left, xs, mg, xe[l]);
            xs = mg;
            if constexpr (kMonitor) gx = max_(max_(tree_from_below<d>(gx), gx), hi_abs(mg.u3));
          }
        };
        xup(std::integral_constant<int, 0>{}); xup(std::integral_constant<int, 1>{}); xup(std::integral_constant<int, 2>{});
        R p0, m0, ml;
        close_root(xs, sm.xbc[parity][0], sm.xbc[parity][1], sm.xbc[parity][2], p0, m0, ml);     // valid in lane W - 1
        R px = read_lane(p0, W - 1), mx = read_lane(ml, W - 1);
        auto xdown = [&](auto lc) __attribute__((always_inline)) {
          constexpr int l = decltype(lc)::value;
          if constexpr (l < LW) {
            Elim<R> e;
            if constexpr (l == 0) {        // quad_perm:[1,1,3,3]
              e.A1 = dpp_mov<0xF5>(xe[l].A1); e.A2 = dpp_mov<0xF5>(xe[l].A2); e.A3 = dpp_mov<0xF5>(xe[l].A3); e.rc = dpp_mov<0xF5>(xe[l].rc);
            } else if constexpr (l == 1) { // quad_perm:[3,3,3,3]
              e.A1 = dpp_mov<0xFF>(xe[l].A1); e.A2 = dpp_mov<0xFF>(xe[l].A2); e.A3 = dpp_mov<0xFF>(xe[l].A3); e.rc = dpp_mov<0xFF>(xe[l].rc);
            } else {
              e.A1 = read_lane(xe[l].A1, 7); e.A2 = read_lane(xe[l].A2, 7); e.A3 = read_lane(xe[l].A3, 7); e.rc = read_lane(xe[l].rc, 7);
            }
            const R sep = separator(e, px, mx);
            const bool upper = ((wl >> l) & 1) != 0;
            px = upper ? e.rc - sep : px;
            mx = upper ? mx : sep;
          }
        };
        xdown(std::integral_constant<int, 2>{}); xdown(std::integral_constant<int, 1>{}); xdown(std::integral_constant<int, 0>{});
        // m of every wave's first row from its own up row: the node there is shared with the wave before, and both
        // copies must move by the same bits
        const R max_ = fma_(-u1o, px, fma_(-u3o, mx, ruo));
        const int ws = __builtin_amdgcn_readfirstlane(wave);
        pL = read_lane(px, ws); mR = read_lane(mx, ws); mAw = read_lane(max_, ws);
        mBw = read_lane(max_, (ws + 1) & (W - 1));
        if (ws == W - 1) mBw = R(0);
        if constexpr (kMonitor) { if (__builtin_amdgcn_readlane(gx, W - 1) > growth_limit_bits<R>()) grow = true; }
      } else if constexpr (FS_XWAVE_CONT && !kMonitor && sizeof(R) == 8 && W > 1) {
        // The W wave segments and the upstream row as ONE tridiagonal system.  With the links p_a(w) = rc_{w-1} - x_{w-1} and every
        // wave's up row  m_a(w) = ru_w - u1_w p_a(w) - u3_w x_w  substituted into the down rows, the unknowns y = (p_0, x_0 .. x_{W-1}),
        // x_w = m of wave w's last row, satisfy
        //   row U  :                 (aU - bU u1_0) p_0 - bU u3_0 x_0                          = rU - bU ru_0
        //   row k  : -d1_k x_{k-1} + (d2_k + d3_k u1_{k+1}) x_k - d3_k u3_{k+1} x_{k+1}        = rd_k - d1_k rc_{k-1} - d3_k (ru_{k+1} - u1_{k+1} rc_k)
        //            (k = 0: + d1_0 p_0 and no rc_{-1} term; k = W-1: nothing to its right)
        // Forward continuants th_i = B_i th_{i-1} - A_i C_{i-1} th_{i-2} with right-hand sides rho_i = th_{i-1} R_i - A_i rho_{i-1},
        // backward ones ph_i, sg_i alike; det = th_W and  y_i = (ph_{i+1} rho_i - C_i th_{i-1} sg_{i+1}) / det : two independent chains
        // of W + 1 steps and ONE reciprocal where the pairwise fold below has four dependent reciprocal chains (two merge levels, the
        // two of the root closure).  The rows are diagonally dominant wherever the elimination itself is sound (fs_device.hpp); the
        // kernels that watch the conditioning (kMonitor) keep the fold, whose merged segments the monitor reads.
        Seg<R> sw[W];
#pragma unroll
        for (int w = 0; w < W; ++w) {
          const R *p = sm.xseg[parity][w];
          sw[w].u1 = p[0]; sw[w].u3 = p[1]; sw[w].ru = p[2]; sw[w].d1 = p[3]; sw[w].d2 = p[4]; sw[w].d3 = p[5];
          sw[w].rd = p[6]; sw[w].rc = p[7];
          tot += sm.xnorm[parity][w];
        }
        constexpr int n = W;                         // unknowns y_0 .. y_n
        const R aU = sm.xbc[parity][0], bU = sm.xbc[parity][1], rU = sm.xbc[parity][2];
        R Ac[n + 1], Bc[n + 1], Cc[n + 1], Rc[n + 1], AC[n + 1];     // AC[i] = A_i C_{i-1}
        Ac[0] = R(0); Bc[0] = fma_(-bU, sw[0].u1, aU); Cc[0] = -(bU * sw[0].u3); Rc[0] = fma_(-bU, sw[0].ru, rU); AC[0] = R(0);
#pragma unroll
        for (int k = 0; k < W; ++k) {
          const int i = k + 1;
          Ac[i] = k == 0 ? sw[0].d1 : -sw[k].d1;
          const R rdk = k == 0 ? sw[0].rd : fma_(-sw[k].d1, sw[k - 1].rc, sw[k].rd);
          if (k + 1 < W) {
            Bc[i] = fma_(sw[k].d3, sw[k + 1].u1, sw[k].d2); Cc[i] = -(sw[k].d3 * sw[k + 1].u3);
            Rc[i] = fma_(-sw[k].d3, fma_(-sw[k + 1].u1, sw[k].rc, sw[k + 1].ru), rdk);
          } else {
            Bc[i] = sw[k].d2; Cc[i] = R(0); Rc[i] = rdk;
          }
          AC[i] = Ac[i] * Cc[i - 1];
        }
        R th[n + 1], rho[n + 1], ph[n + 2], sg[n + 2];
        th[0] = Bc[0]; rho[0] = Rc[0];
        th[1] = fma_(Bc[1], th[0], -AC[1]); rho[1] = fma_(th[0], Rc[1], -(Ac[1] * rho[0]));
#pragma unroll
        for (int i = 2; i <= n; ++i) { th[i] = fma_(Bc[i], th[i - 1], -(AC[i] * th[i - 2])); rho[i] = fma_(th[i - 1], Rc[i], -(Ac[i] * rho[i - 1])); }
        ph[n + 1] = R(1); sg[n + 1] = R(0);
        ph[n] = Bc[n]; sg[n] = Rc[n];
        ph[n - 1] = fma_(Bc[n - 1], ph[n], -AC[n]); sg[n - 1] = fma_(ph[n], Rc[n - 1], -(Cc[n - 1] * sg[n]));
#pragma unroll
        for (int i = n - 2; i >= 1; --i) { ph[i] = fma_(Bc[i], ph[i + 1], -(AC[i + 1] * ph[i + 2])); sg[i] = fma_(ph[i + 1], Rc[i], -(Cc[i] * sg[i + 1])); }
        const R rdet = frcp(th[n]);
        R y[n + 1];
        y[0] = fma_(ph[1], rho[0], -(Cc[0] * sg[1])) * rdet;
#pragma unroll
        for (int i = 1; i < n; ++i) y[i] = fma_(ph[i + 1], rho[i], -(Cc[i] * th[i - 1] * sg[i + 1])) * rdet;
        y[n] = rho[n] * rdet;
        R pw[W], ma[W + 1];
        pw[0] = y[0];
#pragma unroll
        for (int w = 1; w < W; ++w) pw[w] = sw[w - 1].rc - y[w];
        // m of every wave's first row from its own up row: the node there is shared with the wave before, and both
        // copies must move by the same bits
#pragma unroll
        for (int w = 0; w < W; ++w) ma[w] = fma_(-sw[w].u1, pw[w], fma_(-sw[w].u3, y[w + 1], sw[w].ru));
        ma[W] = R(0);
        pL = pw[0]; mR = y[1]; mAw = ma[0]; mBw = ma[1];
#pragma unroll
        for (int w = 1; w < W; ++w)
          if (wave == w) { pL = pw[w]; mR = y[w + 1]; mAw = ma[w]; mBw = ma[w + 1]; }
      } else {
      {
        // pairwise tree over the W wave segments (depth log2 W instead of a serial chain of W-1 merges;
        // the two merges of a level are independent and overlap); every thread does all of it
        Seg<R> sw[W], sw0[W];
        Elim<R> we[W > 1 ? W - 1 : 1];
#pragma unroll
        for (int w = 0; w < W; ++w) {
          const R *p = sm.xseg[parity][w];
          sw[w].u1 = p[0]; sw[w].u3 = p[1]; sw[w].ru = p[2]; sw[w].d1 = p[3]; sw[w].d2 = p[4]; sw[w].d3 = p[5];
          sw[w].rd = p[6]; sw[w].rc = p[7];
          sw0[w] = sw[w];
          tot += sm.xnorm[parity][w];
        }
        int gx = 0;
        if constexpr (kMonitor) {
#pragma unroll
          for (int w = 0; w < W; ++w) gx = max_(gx, sm.xg[parity][w]);
        }
#pragma unroll
        for (int st = 1; st < W; st *= 2)
#pragma unroll
          for (int i = 0; i + st < W; i += 2 * st) {
            merge(sw[i], sw[i + st], sw[i], we[i + st - 1]);
            if constexpr (kMonitor) gx = max_(gx, hi_abs(sw[i].u3));
          }
        if constexpr (kMonitor) { if (gx > growth_limit_bits<R>()) grow = true; }
        R pw[W], mw[W], p0, m0, ml;      // pw[w], mw[w]: the two numbers of wave w (valid for the group heads while unfolding)
        close_root(sw[0], sm.xbc[parity][0], sm.xbc[parity][1], sm.xbc[parity][2], p0, m0, ml);
        pw[0] = p0; mw[0] = ml;
#pragma unroll
        for (int st = W / 2; st >= 1; st /= 2)
#pragma unroll
          for (int i = 0; i + st < W; i += 2 * st) {
            const Elim<R> &e = we[i + st - 1];
            const R sep = separator(e, pw[i], mw[i]);
            pw[i + st] = e.rc - sep; mw[i + st] = mw[i]; mw[i] = sep;
          }
        // m of every wave's first row from its own up row: the node there is shared with the wave before, and both
        // copies must move by the same bits
        R ma[W + 1];
#pragma unroll
        for (int w = 0; w < W; ++w) ma[w] = fma_(-sw0[w].u1, pw[w], fma_(-sw0[w].u3, mw[w], sw0[w].ru));
        ma[W] = R(0);
        pL = pw[0]; mR = mw[0]; mAw = ma[0]; mBw = ma[1];
#pragma unroll
        for (int w = 1; w < W; ++w)
          if (wave == w) { pL = pw[w]; mR = mw[w]; mAw = ma[w]; mBw = ma[w + 1]; }
      }
      }
#if FS_XWAVE_FENCE & 2
      __builtin_amdgcn_sched_barrier(0);
#endif
      FS_T(4);
      if (!kTeam && (BCK < 2 || ds_storage) && sm.xflag[parity] != 0) status = sm.xflag[parity];     // only the storage rows raise a flag (a team's: through the mailbox, above)
      // ||R|| = sqrt(tot) (utility.py:20-22) is NaN or beyond the blow-up bound (preissmann.py:135-137) exactly when tot is
      // not finite (the bound exceeds the root of the largest finite number in either precision)
      if (!(tot <= finite_max<R>())) status = FS_NAN;
      bool below;                                                      // ||R|| < tolerance (preissmann.py:153)
      if (DIAG || M < 8) {               // (the two-rows-per-lane kernels sit on their register cap: the branch below costs C4 19 %)
        const R err = sqrt_(tot);
        if constexpr (BCK <= 0) below = err < tol_of(); else
        below = err < a.tol;
        if (DIAG && a.trace && gt == 0 && it <= FS_TRACE_CAP) a.trace[((size_t)level * FS_TRACE_CAP + (it - 1)) * a.B + reach] = err;
      } else {
        // without a trace to write the root is only needed when tot lies within rounding of tol^2 (the decision must be
        // the one sqrt() gives, bit for bit: Newton counts are compared with the reference's)
        if constexpr (BCK <= 0) {
          const R tl = tol_of();
          const R t2 = tl * tl, band = R(16) * eps_of<R>() * t2;
          below = tot < t2 - band ? true : (tot > t2 + band ? false : sqrt_(tot) < tl);
        } else {      // (spelled as it always was: the benchmark kernels keep their machine code, tools/isa_digest.py)
        const R t2 = a.tol * a.tol, band = R(16) * eps_of<R>() * t2;
        below = tot < t2 - band ? true : (tot > t2 + band ? false : sqrt_(tot) < a.tol);
        }
      }
      converged = status == FS_OK && below;
      // The warning is about the system the level's result comes from - the Jacobian at the accepted iterate.  An iterate on the
      // way there may pass a small pivot by accident (a soak draw did: a three-node reach behind a reservoir, condition number
      // 3e4 throughout, |u3| > 2^10 once in the 12 iterations of its first level); Newton does not remember that.
      if (converged && grow) warn = true;

      FS_T(5);
      // ================= 5. separators down the tree, local back-substitution ============
      // Every lane carries two numbers of the group of 2^(l+1) lanes it belongs to at the current level: the p of the
      // group's first row and the m of its last one.  The group's record is one LDS slot that all its lanes read (a
      // broadcast read); each lane recovers the separator itself and keeps it as its new right number (lower half) or
      // turns it into its new left one (upper half): no cross-lane traffic, two dependent fp64 operations per level.
      auto load_rec = [&](auto lc) __attribute__((always_inline)) {
        constexpr int l = decltype(lc)::value;
        Elim<R> e;
        const int slot = (64 - (64 >> l)) + (ln >> (l + 1));
        const R *p = &sm.tree[wave][0][slot];
        e.A1 = p[0 * 64]; e.A2 = p[1 * 64]; e.A3 = p[2 * 64]; e.rc = p[3 * 64];
        return e;
      };
      auto down_level = [&](auto lc, const Elim<R> &e) __attribute__((always_inline)) {
        constexpr int l = decltype(lc)::value;
        const R sep = separator(e, pL, mR);
        const bool upper = ((lane >> l) & 1) != 0;
        pL = upper ? e.rc - sep : pL;
        mR = upper ? mR : sep;
      };
      {
        // records are requested one level ahead (the memory clobbers keep the compiler from requesting all six up front)
        using I5 = std::integral_constant<int, 5>; using I4 = std::integral_constant<int, 4>;
        using I3 = std::integral_constant<int, 3>; using I2 = std::integral_constant<int, 2>;
        using I1 = std::integral_constant<int, 1>; using I0 = std::integral_constant<int, 0>;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (kTreeRegs) {
          // the group's record from the group's last lane: readlane (the wave, its halves), row_newbcast (rows of 16, halves of a
          // row through the bank mask), quad_perm (quads, pairs)
          auto fetch = [&](auto lc2, R v) __attribute__((always_inline)) {
            constexpr int l = decltype(lc2)::value;
            if constexpr (l == 5) return read_lane(v, 63);
            else if constexpr (l == 4) { const R lo = read_lane(v, 31), hi = read_lane(v, 63); return lane < 32 ? lo : hi; }
            else if constexpr (l == 3) return dpp_mov<0x15F>(v);
            else if constexpr (l == 2) return dpp_mov_banks<0x15F, 0xC>(dpp_mov_banks<0x157, 0x3>(R(0), v), v);
            else if constexpr (l == 1) return dpp_mov<0xFF>(v);
            else return dpp_mov<0xF5>(v);
          };
          auto rec_of = [&](auto lc2) __attribute__((always_inline)) {
            constexpr int l = decltype(lc2)::value;
            Elim<R> e;
            e.A1 = fetch(lc2, rec[l].A1); e.A2 = fetch(lc2, rec[l].A2); e.A3 = fetch(lc2, rec[l].A3); e.rc = fetch(lc2, rec[l].rc);
            return e;
          };
          down_level(I5{}, rec_of(I5{})); down_level(I4{}, rec_of(I4{})); down_level(I3{}, rec_of(I3{}));
          down_level(I2{}, rec_of(I2{})); down_level(I1{}, rec_of(I1{})); down_level(I0{}, rec_of(I0{}));
        } else {
        Elim<R> r5, r4, r3;
        if constexpr (kPre >= 1) r5 = pre5; else r5 = load_rec(I5{});
        if constexpr (kPre >= 2) r4 = pre4; else r4 = load_rec(I4{});
        asm volatile("" ::: "memory");
        down_level(I5{}, r5);
        if constexpr (kPre >= 3) r3 = pre3; else r3 = load_rec(I3{});
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
        }
        asm volatile("" : "+v"(pL), "+v"(mR));
        __builtin_amdgcn_sched_barrier(0);
      }
      // m of the lane's first row from its up row; a wave's first lane takes the number the cross-wave step computed, so
      // that the last lane of the wave before (which gets the same number) moves its copy of the shared node identically
      R mA = fma_(-upU1, pL, fma_(-upU3, mR, upRu));
      if (W > 1 && lane == 0) mA = mAw;
      R mB = dpp_mov<0x134>(mA);                // wave_rol:1 - m of the next lane's first row = this lane's node M
      if (lane == 63) mB = mBw;

      // The update is kept pending in dh/dQ (they take the registers the elimination records free up):
      // the accepted iterate must still be intact for the level-constant pass below.
      if (Geo::kConstT && FS_LAUNDER_BACK && !kRcEarly) {
        // The continuity residuals are recomputed below on purpose (one value per node less to keep
        // across the solve).  Hide the operands so that common-subexpression elimination does not
        // resurrect the fold's copies of dQ / kc0 and keep 2 values per node alive instead.
        asm volatile("" : "+v"(kco));
        kcb = &sm.kc[0][0][0] + kco;
#pragma unroll
        for (int j = 0; j <= M; ++j) asm volatile("" : "+v"(h[j]), "+v"(Q[j]));
      }
      {
        const R i2tc = dt * geo.rT_const();
        auto i2t_of = [&](int j) { return Geo::kConstT ? i2tc : iTn[j]; };
        // rc of row j for the link p_{j+1} = rc_j - m_j: minus the continuity residual of cell j, 0 beyond the cells
        auto rc_of = [&](int j) __attribute__((always_inline)) {
          if constexpr (kRcEarly) return rcE[j];
          if (Geo::kConstT) {
            const R v = -(geo.terms_T() * (h[j] + h[j + 1]) * r2dt + cq * (Q[j + 1] - Q[j]) + kcb[(0 * M + j) * T]);
            return (RAGGED && s0 + j >= NC) ? R(0) : v;
          }
          return j + 1 < M ? el[j].qc.get() : rcLast;
        };
        R mj = mR;                                  // m of row M-1
        {
          // node M: p from this lane's last row, m from the next lane.  rcLast, not a recomputed residual: the next lane's
          // p of its first row is (this rc) - (this m), and both copies of the shared node must move by the same bits
          const R pM = rcLast - mR;
          dh[M] = (pM + mB) * i2t_of(M); dQ[M] = (pM - mB) * i2c;
          if constexpr (kTail) { dh[M] *= mlt; dQ[M] *= mlt; }       // (node M > TAIL always: a phantom node from the owner lane on)
        }
#pragma unroll
        for (int j = M - 1; j >= 1; --j) {
          // m_{j-1} = R3 - R1 p_a - R2 m_j ; row 0's m comes from the up row (mA)
          const R mprev = j == 1 ? mA : fma_(-el[j - 1].R2.get(), mj, fma_(-el[j - 1].R1.get(), pL, el[j - 1].R3.get()));
          const R pj = rc_of(j - 1) - mprev;
          dh[j] = (pj + mj) * i2t_of(j); dQ[j] = (pj - mj) * i2c;
          if constexpr (kTail) { dh[j] *= (j <= TAIL ? mle : mlt); dQ[j] *= (j <= TAIL ? mle : mlt); }     // phantom nodes stay where they are
          mj = mprev;
        }
        dh[0] = (pL + mA) * i2t_of(0); dQ[0] = (pL - mA) * i2c;
        if constexpr (kTail) { dh[0] *= mle; dQ[0] *= mle; }
      }
      FS_T(6);
      if constexpr (((sizeof(R) == 4 ? FS_PHASE_FENCE_F32 : FS_PHASE_FENCE) & 8) != 0) {
#pragma unroll
        for (int j = 1; j < M; ++j) asm volatile("" : "+v"(dh[j]), "+v"(dQ[j]));
        __builtin_amdgcn_sched_barrier(0);
      }
#if FS_PHASE_FENCE & 1
      __builtin_amdgcn_sched_barrier(0);
#endif
      }   // !prime

      // ================= 6. accepted iterate -> level k =================
      if (converged) {
        if (!prime && gt == 0) {
          a.hydro[((size_t)level * 4 + 0) * a.B + reach] = h[0];
          a.hydro[((size_t)level * 4 + 1) * a.B + reach] = Q[0];
          a.iters[(size_t)level * a.B + reach] = it;
        }
        if (!prime) {
          // Level k+1 is rebuilt from registers, so the accepted state only has to reach HBM when
          // somebody can look at it: at the last level of this launch (fs_batch_get_state, next launch)
          // and, if a history is kept, at every level.  (A transposed, fully coalesced write-back
          // through LDS was measured too: no gain, three extra barriers.)
          const bool last = (step == a.n_steps - 1);
          R *const hh_p = (DIAG && a.hist_h) ? a.hist_h + ((size_t)level * a.B + reach) * NS + s0 : nullptr;
          R *const hQ_p = (DIAG && a.hist_h) ? a.hist_Q + ((size_t)level * a.B + reach) * NS + s0 : nullptr;
          if (last || hh_p) {
#pragma unroll
            for (int j = 0; j < M; ++j) {
              if (!RAGGED || s0 + j < N) {
                if (last) { hk_p[j] = h[j]; Qk_p[j] = Q[j]; }
                if (hh_p) { hh_p[j] = h[j]; hQ_p[j] = Q[j]; }
              }
            }
          }
        }
        FS_T(8);
        if (gt == tD) {
#pragma unroll
          for (int j = RAGGED ? 0 : M - 1; j < M; ++j)
            if (j == jD) {
              if (!prime) {
                a.hydro[((size_t)level * 4 + 2) * a.B + reach] = h[j];
                a.hydro[((size_t)level * 4 + 3) * a.B + reach] = Q[j];
              }
              QoldD = Q[j];                                             // flow[k] of the next level's storage row
            }
          if (!prime) {
            Yprev = Ynew;
            if (ds_storage) a.stage_hist[(size_t)level * a.B + reach] = Ynew;
          }
        }
        FS_T(9);
        if (kSaveTerms) {                                             // level constants of the next level
          if (prime) {                           // no fold has run yet: the node terms of the entry state go where a fold leaves them
            NodeTerms<R> L = terms_at(0, h[0], Q[0]);
            save_terms(0, L);
#pragma unroll
            for (int c = 0; c < M; ++c) save_terms(c + 1, terms_at(c + 1, h[c + 1], Q[c + 1]));
          }
          level_constants_from_saved(h, Q);
        } else {
          write_level_constants(h, Q);
        }
        FS_T(10);
      }

      FS_T(5);
      if (prime) {                               // Newton start vector of the first level to solve
#pragma unroll
        for (int j = 0; j <= M; ++j) {
          const int node = min(s0 + j, N - 1);
          h[j] = a.hg[base + node];  Q[j] = a.Qg[base + node];
        }
        primed = true;
      } else {
#pragma unroll
        for (int j = 0; j <= M; ++j) { h[j] += dh[j]; Q[j] += dQ[j]; }     // preissmann.py:146-147
      }
    }
    if (prime) continue;
    if (status != FS_OK && gt == 0) a.iters[(size_t)level * a.B + reach] = it - (status == FS_MAX_ITER ? 1 : 0);
    if (kBudget && a.iter_budget > 0) {
      if (t == 0) a.it_done[reach] = (converged || status != FS_OK) ? -1 : it;
      if (!converged) break;                  // budget spent: the Newton vector goes back to hg / Qg below
    }
  }

  // ---- Newton start vector of the next level + per-reach bookkeeping ----
  if (primed) {                               // (a reach that came in failed keeps what it had)
#pragma unroll
    for (int j = 0; j < M; ++j)
      if (!RAGGED || s0 + j < N) { hg_p[j] = h[j]; Qg_p[j] = Q[j]; }
  }
  if (gt == 0) a.status[reach] = (status == FS_OK && warn) ? (int)FS_ILL_CONDITIONED : status;
  if (ds_storage && gt == tD) a.Yprev[reach] = Yprev;
#ifdef FS_STAMP
  if (a.dbg && lane == 0)
    for (int i = 0; i < 12; ++i) a.dbg[((size_t)reach * 16 + (member * W + wave) % 16) * 12 + i] = stamp_[i];
#endif
}


// ---------------------------------------------------------------------------------------------
// Post-processing (reference: Solver.prepare_results, solver.py:65-127).  One thread per V consecutive
// (reach, node) elements walks the stored levels: 16-byte loads / stores (V = 2 doubles, 4 floats), consecutive
// threads on consecutive elements, 16 B in and up to 56 B out per element - a plain HBM-bound stream.
// ---------------------------------------------------------------------------------------------
template <typename R> struct DeriveArgs {
  int32_t B, N, first, n, section_mode;
  const R *hist_h, *hist_Q;       // [levels][B][N]
  const R *geo_uniform, *geo_table;
  const R *poly_x, *poly_z;       // IRREGULAR (see KernelArgs)
  const int32_t *poly_n;
  int64_t geo_reach_stride, poly_reach_stride;      // per-reach tables (see KernelArgs), 0: shared
  R *level, *area, *top, *froude, *vel, *cel, *amp, *peak;   // [n][B][N] (peak: [B][N]) or nullptr
  const int32_t *reach_nodes;     // [B] or nullptr: nodes of each reach of a ragged batch (fs_batch_set_reach_nodes); entries beyond come out 0
};

template <typename R, int V> struct Pack { typedef R type __attribute__((ext_vector_type(V))); };

// V elements from / to p: one 16-byte access when the thread's elements all exist and the row is aligned (whole)
template <typename R, int V> __device__ __forceinline__ void load_pack(const R *__restrict__ p, bool whole, int cnt, R (&out)[V]) {
  if (whole) {
    const typename Pack<R, V>::type q = *reinterpret_cast<const typename Pack<R, V>::type *>(p);
#pragma unroll
    for (int e = 0; e < V; ++e) out[e] = q[e];
  } else {
#pragma unroll
    for (int e = 0; e < V; ++e) out[e] = e < cnt ? p[e] : R(1);
  }
}
template <typename R, int V> __device__ __forceinline__ void store_pack(R *__restrict__ p, bool whole, int cnt, const R (&in)[V]) {
  if (whole) {
    typename Pack<R, V>::type q;
#pragma unroll
    for (int e = 0; e < V; ++e) q[e] = in[e];
    __builtin_nontemporal_store(q, reinterpret_cast<typename Pack<R, V>::type *>(p));     // written once, read by nobody on the device
  } else {
#pragma unroll
    for (int e = 0; e < V; ++e) if (e < cnt) p[e] = in[e];
  }
}

template <typename R, int V> __global__ __launch_bounds__(256) void derive_fields_kernel(const DeriveArgs<R> a) {
  const size_t BN = (size_t)a.B * a.N;
  const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * V;
  if (i0 >= BN) return;
  const int cnt = (int)(BN - i0 < (size_t)V ? BN - i0 : (size_t)V);
  const bool whole = cnt == V && BN % V == 0;        // every level's row of this thread starts on a 16-byte boundary
  SecParams<R> s[V];
  PolyNode<R> pnode[V];
  bool beyond[V];                            // ragged batches: a slot past the reach's own node count (no history there: results 0)
#pragma unroll
  for (int e = 0; e < V; ++e) {
    const size_t i = i0 + (e < cnt ? e : 0);
    const int reach = (int)(i / a.N), node = (int)(i - (size_t)reach * a.N);
    const int nodes_r = a.reach_nodes ? a.reach_nodes[reach] : a.N;        // the step kernels' Geometry::init(a, reach, n_nodes)
    beyond[e] = node >= nodes_r;
    pnode[e].n = 0;
    if (a.section_mode == FS_SEC_TABLE || a.section_mode == FS_SEC_IRREGULAR) {
      auto g = [&](int row) { return a.geo_table[(size_t)reach * a.geo_reach_stride + (size_t)row * a.N + node]; };
      s[e].z = g(FS_GEO_Z_BED); s[e].b = g(FS_GEO_B_MAIN); s[e].m = g(FS_GEO_M_MAIN);
      s[e].compound = g(FS_GEO_IS_COMPOUND) > R(0.5);
      s[e].hbf = g(FS_GEO_H_BANKFULL); s[e].bl = g(FS_GEO_B_FP_LEFT); s[e].br = g(FS_GEO_B_FP_RIGHT); s[e].mfp = g(FS_GEO_M_FP);
      if (a.section_mode == FS_SEC_IRREGULAR) {
        const size_t po = (size_t)reach * a.poly_reach_stride, no = a.poly_reach_stride ? (size_t)reach * a.N : 0;
        if (a.poly_n[no + node] > 0) {
          pnode[e].x = a.poly_x + po + node; pnode[e].z = a.poly_z + po + node; pnode[e].stride = a.N; pnode[e].n = a.poly_n[no + node];
          pnode[e].tz = nullptr; pnode[e].K = 0; pnode[e].KP = 0;
        }
      }
    } else {
      const R z_us = a.geo_uniform[(size_t)FS_RU_Z_US * a.B + reach], z_ds = a.geo_uniform[(size_t)FS_RU_Z_DS * a.B + reach];
      const R w2 = R(node) * (R(1) / R(nodes_r - 1));       // the reach's own node count, as Geometry<R, *_UNIFORM>::bed
      s[e].z = z_us * (R(1) - w2) + z_ds * w2;
      s[e].b = a.geo_uniform[(size_t)FS_RU_WIDTH * a.B + reach];
      s[e].m = a.section_mode == FS_SEC_TRAP_UNIFORM ? a.geo_uniform[(size_t)FS_TU_SIDE_SLOPE * a.B + reach] : R(0);
      s[e].compound = false; s[e].hbf = s[e].bl = s[e].br = s[e].mfp = R(0);
    }
  }
  R h0[V], peak[V];                          // depth[0] (amplitude reference, solver.py:96-97)
  load_pack<R, V>(a.hist_h + i0, whole, cnt, h0);
#pragma unroll
  for (int e = 0; e < V; ++e) peak[e] = R(-3.0e38);
  for (int k = 0; k < a.n; ++k) {
    const size_t src = (size_t)(a.first + k) * BN + i0, dst = (size_t)k * BN + i0;
    R h[V], Q[V], lev[V], A[V], T[V], Fr[V], vel[V], cel[V], am[V];
    load_pack<R, V>(a.hist_h + src, whole, cnt, h);
    load_pack<R, V>(a.hist_Q + src, whole, cnt, Q);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      if (beyond[e]) { h[e] = R(1); Q[e] = R(0); }        // (never written by the step kernel)
      // area and top width: cross_section.py:623-679 (incl. the over-bank convention)
      const SecParams<R> &se = s[e];
      const R d = fmax_(R(0), h[e]);
      R Te = se.b + R(2) * se.m * d;
      R Ae = (se.b + Te) / R(2) * d;
      if (se.compound && d > se.hbf) {
        const R dfp = d - se.hbf, Tb = se.b + R(2) * se.m * se.hbf;
        Ae = (se.b + Tb) / R(2) * se.hbf + (se.bl + R(0.5) * se.mfp * dfp) * dfp + (se.br + R(0.5) * se.mfp * dfp) * dfp;
        Te = (se.bl + Tb + se.br) + R(2) * se.mfp * dfp;
      }
      if (d <= R(0)) { Ae = R(0); Te = R(0); }
      if (pnode[e].n > 0) poly_area_top(pnode[e], h[e] + se.z, Ae, Te);     // cross_section.py:248-328
      const R Ve = Q[e] / Ae;
      lev[e] = h[e] + se.z; A[e] = Ae; T[e] = Te; vel[e] = Ve;
      {                                        // hydraulics.py:155-168 with its clamps
        const R Vc = Q[e] / fmax_(Ae, R(1e-6)), D = Ae / fmax_(Te, R(1e-6));
        Fr[e] = Vc / sqrt_(R(kG) * fmax_(D, R(1e-6)));
      }
      cel[e] = Ve + sqrt_(R(kG) * Ae / Te);
      am[e] = h[e] - h0[e];
      peak[e] = fmax_(peak[e], am[e]);
      if (beyond[e]) { lev[e] = A[e] = T[e] = Fr[e] = vel[e] = cel[e] = am[e] = R(0); peak[e] = R(0); }
    }
    if (a.level) store_pack<R, V>(a.level + dst, whole, cnt, lev);
    if (a.area) store_pack<R, V>(a.area + dst, whole, cnt, A);
    if (a.top) store_pack<R, V>(a.top + dst, whole, cnt, T);
    if (a.froude) store_pack<R, V>(a.froude + dst, whole, cnt, Fr);
    if (a.vel) store_pack<R, V>(a.vel + dst, whole, cnt, vel);
    if (a.cel) store_pack<R, V>(a.cel + dst, whole, cnt, cel);
    if (a.amp) store_pack<R, V>(a.amp + dst, whole, cnt, am);
  }
  if (a.peak) store_pack<R, V>(a.peak + i0, whole, cnt, peak);
}

}  // namespace fs

#include "fs_long.hpp"
