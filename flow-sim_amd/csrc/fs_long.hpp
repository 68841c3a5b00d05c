// fs_long.hpp - the step kernel for reaches longer than one lane grid (N > 64 W M rows).
//
// The reference has no limit on the number of nodes (solver.py:34-38, :53-55: round(L / dx) + 1); preissmann_step_kernel has
// one because a reach's unknowns live in the registers of one workgroup and its level constants in that workgroup's LDS
// (4 096 rows: 128 KB).  Here a workgroup walks its reach in P passes of C = 64 W M rows each, and what the short kernel
// keeps on chip lives in memory the workgroup owns, read and written through L2 / the Infinity Cache:
//
//   * the Newton vector itself stays in hg / Qg ([B][N], node-major) - loaded at the start of a pass, stored at its end;
//   * the four level constants per cell go to a per-reach scratch ([4][P C], lane-minor within a pass like the LDS layout of
//     the short kernel: consecutive lanes read consecutive doubles);
//   * a Newton iteration makes two sweeps over the passes with the SAME code (one loop body, so both sweeps execute the
//     same instructions and produce the same bits):
//       sweep 0: fold + in-wave tree of every pass -> one segment per (pass, wave), at most 64 in all;
//       then wave 0 reduces those segments with one more DPP tree (identity segments pad the 64 lanes), closes the root with
//                the upstream boundary row and sends (p of the first row, m of the last row, m of the two shared nodes) of
//                every (pass, wave) back through LDS;
//       sweep 1: fold + in-wave tree again (their records are recomputed rather than stored: 2 KB of LDS per wave instead of
//                128 KB for 64 sub-trees), way down, back-substitution, acceptance, update, store.
//     The price is a second fold per iteration and ~130 B of L2 traffic per node and iteration - far from any HBM bound
//     (state and scratch of the reaches in flight, 256 x 1 MB at 16 384 nodes, sit in the Infinity Cache).
//
// Same arithmetic as the short kernel (same node terms, rows, merges), general form only: ragged node counts, boundary kinds
// switched at run time (class 0 for the uniform section modes, -1 - incl. the general storage row - for tables and polylines),
// diagnostics compiled in (history, residual trace, conditioning monitor).  The kernels of boundary class -1 (tables, polylines)
// carry the iteration budget of fs_batch_iterate like their on-chip counterparts: a boundary that only the host can evaluate - a
// RatingCurve subclass with moving gates, a Python callable as reservoir outlet (boundary.py:56-141; FS_BC_HOST_ROW) - runs on a
// reach of any length, one Newton iteration per launch.  Nothing has to be carried across launches here: the Newton vector lives
// in hg / Qg anyway, the level constants are rebuilt from hk / Qk (untouched while a level is open) at every launch.
#pragma once

namespace fs {

template <typename R, int W, int M> struct SmemLong {
  // per-wave staging of the wave's 64 M (+ 1) nodes of a pass: global memory is read and written with consecutive lanes on
  // consecutive nodes, a lane takes its M + 1 consecutive nodes from here (one pad per 8 numbers: the lanes' chunks start in
  // different banks)
  R stage[W][2][FS_LONG_COALESCE ? 64 * M + 8 * M + 8 : 1];
  R tree[W][4][64];        // in-wave tree records of the pass being worked on
  R xtree[4][64];          // records of the top tree over the (pass, wave) segments (wave 0)
  R xseg[64][8];           // one segment per (pass, wave)
  R xres[64][4];           // per (pass, wave): p of its first row, m of its last row, m of its first row, m of the next one's first row
  R xbc[4];                // upstream boundary row on (p_0, m_0)
  R bcp[2][FS_BC_MAX_PARAMS];
  R xnorm[W];
  R xtot;
  int32_t xg[64];
  int32_t xflag, xwarn;
};

// waves per SIMD the kernel is compiled for: without the level constants in flight the uniform-section kernels need 298 registers;
// capped at 256 two workgroups share a CU and one covers the other's memory waits (FS_LONG_WPE)
template <typename R, int SEC> constexpr int long_min_waves() {
  return (FS_LONG_RECOMPUTE && (SEC == FS_SEC_RECT_UNIFORM || SEC == FS_SEC_TRAP_UNIFORM) && sizeof(R) == 8) ? FS_LONG_WPE : 1;
}

template <typename R, int SEC, int M, int W, int BCK>
__global__ __launch_bounds__(64 * W, (long_min_waves<R, SEC>())) void preissmann_long_kernel(const KernelArgs<R> a) {
  static_assert(BCK == 0 || BCK == -1, "long reaches: run-time boundary kinds only");
  constexpr int T = 64 * W, C = T * M;
  using Geo = Geometry<R, SEC>;
  // Uniform sections: the four level constants of a cell are recomputed in every sweep from the accepted state of level k (two node
  // evaluations of ~40 instructions) instead of being fetched from a scratch: 16 B instead of 32 B per node and sweep for a kernel
  // that is bound by HBM.  Same expressions as the stored ones, hence the same bits.
  constexpr bool kRecompute = FS_LONG_RECOMPUTE && (SEC == FS_SEC_RECT_UNIFORM || SEC == FS_SEC_TRAP_UNIFORM);
  __shared__ SmemLong<R, W, M> sm;

  const int reach = blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // fs_batch_iterate (boundary class -1): the iterations already spent on the open level, -1 once this reach has closed it
  constexpr bool kBudget = (BCK == -1);
  int it_entry = 0;
  if (kBudget && a.iter_budget > 0) {
    it_entry = a.it_done[reach];
    if (it_entry < 0) return;                 // (whole workgroup)
  }
  const int NS = a.N;
  const int N = a.reach_nodes ? a.reach_nodes[reach] : a.N, NC = N - 1;
  const int P = a.passes;                                     // P C >= N rows, P W <= 64 segments
  const size_t RP = (size_t)P * C;
  R *const kcg = kRecompute ? nullptr : a.kc_scratch + (size_t)reach * 4 * RP;
  const size_t base = (size_t)reach * NS;
  // the row N-1 (downstream boundary row) and the node N-1: pass, lane, local index
  const int pD = NC / C, tD = (NC - pD * C) / M, jD = NC - pD * C - tD * M;

  Geo geo;
  geo.init(a, reach, N);
  const bool own_scheme = a.reach_scheme != nullptr;
  const R th = own_scheme ? a.reach_scheme[reach] : a.theta, dt = own_scheme ? a.reach_scheme[(size_t)a.B + reach] : a.dt;
  const R dx_ = own_scheme ? a.reach_scheme[(size_t)2 * a.B + reach] : a.dx;
  const R tol_r = own_scheme ? a.reach_scheme[(size_t)3 * a.B + reach] : a.tol;       // the reach's own run(tolerance, max_iter)
  const int max_it = own_scheme ? (int)a.reach_scheme[(size_t)4 * a.B + reach] : a.max_iter;
  const R r2dt = R(1) / (R(2) * dt), cq = th / dx_, cqk = (R(1) - th) / dx_, hth = R(0.5) * th, hthk = R(0.5) * (R(1) - th);
  const R g = R(kG), i2c = R(0.5) / cq, kap = r2dt * i2c, dtcq = dt * cq, hx = hth * i2c, ghth = g * hth, ghthk = g * hthk, ghdt = g * hth * dt;

  BCDesc<R> usd = a.us, dsd = a.ds;
  if (a.reach_kinds) { usd.kind = a.reach_kinds[reach]; dsd.kind = a.reach_kinds[(size_t)a.B + reach]; }
  if (t < 2 * FS_BC_MAX_PARAMS) {
    const int side = t / FS_BC_MAX_PARAMS, i = t - side * FS_BC_MAX_PARAMS;
    BCDesc<R> src = side ? a.ds : a.us;
    src.kind = side ? dsd.kind : usd.kind;
    static constexpr int kCount[] = {0, 1, 1, 2, 4, 5, 10, 5};
    if (src.kind <= FS_BC_STORAGE && i < kCount[src.kind]) sm.bcp[side][i] = bc_param(src, i, reach, a.B);
    if (src.kind == FS_BC_NORMAL_DEPTH && i == 2) {
      const R S0 = bc_param(src, 0, reach, a.B);
      sm.bcp[side][2] = (S0 < R(0) ? R(-1) : R(1)) * sqrt_(fabs_(S0));
    }
  }
  if (usd.kind <= FS_BC_STORAGE) { usd.params = &sm.bcp[0][0]; usd.stride = 0; }
  if (dsd.kind <= FS_BC_STORAGE) { dsd.params = &sm.bcp[1][0]; dsd.stride = 0; }
  const bool ds_storage = bc_is_storage(dsd.kind);
  const bool isD = t == tD;                                   // (in pass pD) the lane of node N-1
  R Yprev = ds_storage ? a.Yprev[reach] : R(0);
  R QoldD = R(0);
  int status = a.status[reach];
  bool warn = status == FS_ILL_CONDITIONED;
  if (warn) status = FS_OK;
  if (t == 0) { sm.xflag = 0; sm.xwarn = 0; }

  auto node_terms = [&](int node, R hh, R QQ) __attribute__((always_inline)) {
    return geo.terms(node, hh, QQ);
  };
  // nodes g0 .. g0 + M of a pass's lane (clamped copies beyond the last node)
  auto load_nodes = [&](const R *hs, const R *Qs, int g0, R(&h)[M + 1], R(&Q)[M + 1]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j <= M; ++j) {
      const int node = min(g0 + j, N - 1);
      h[j] = hs[base + node]; Q[j] = Qs[base + node];
    }
  };
  // The same nodes with consecutive lanes on consecutive nodes (a lane's own M + 1 nodes are 8 M bytes apart from its neighbour's:
  // every load instruction of load_nodes touches 64 separate 64-byte segments, and the texture addresser was busy 82 % of the
  // kernel's time with them, profiles/round3/long_mem.txt), transposed through the wave's staging buffer.  LDS executes a wave's
  // instructions in order; the fences keep the compiler from moving the reads over the writes.
  constexpr bool kCo = FS_LONG_COALESCE != 0;
  auto pad = [](int i) { return i + (i >> 3); };
  auto wave_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto load_nodes_co = [&](const R *hs, const R *Qs, int p, R(&h)[M + 1], R(&Q)[M + 1]) __attribute__((always_inline)) {
    const int wb = p * C + wave * (64 * M);                   // first node of this wave in this pass
    auto one = [&](const R *src, R *st, R(&out)[M + 1]) __attribute__((always_inline)) {
      R tv[M + 1];
#pragma unroll
      for (int k = 0; k < M; ++k) tv[k] = src[base + min(wb + lane + 64 * k, N - 1)];
      if (lane == 0) tv[M] = src[base + min(wb + 64 * M, N - 1)];     // the node the wave shares with its right neighbour
#pragma unroll
      for (int k = 0; k < M; ++k) st[pad(lane + 64 * k)] = tv[k];
      if (lane == 0) st[pad(64 * M)] = tv[M];
      wave_sync();
#pragma unroll
      for (int j = 0; j <= M; ++j) out[j] = st[pad(lane * M + j)];
    };
    if constexpr (FS_LONG_COALESCE == 2) {      // one array after the other (half the numbers in flight)
      one(hs, sm.stage[wave][0], h);
      one(Qs, sm.stage[wave][1], Q);
      wave_sync();
    } else {
    R *const sh = sm.stage[wave][0], *const sq = sm.stage[wave][1];
    R th[M + 1], tq[M + 1];
#pragma unroll
    for (int k = 0; k < M; ++k) {
      const int node = min(wb + lane + 64 * k, N - 1);
      th[k] = hs[base + node]; tq[k] = Qs[base + node];
    }
    if (lane == 0) {                                          // the node the wave shares with its right neighbour
      const int node = min(wb + 64 * M, N - 1);
      th[M] = hs[base + node]; tq[M] = Qs[base + node];
    }
#pragma unroll
    for (int k = 0; k < M; ++k) { sh[pad(lane + 64 * k)] = th[k]; sq[pad(lane + 64 * k)] = tq[k]; }
    if (lane == 0) { sh[pad(64 * M)] = th[M]; sq[pad(64 * M)] = tq[M]; }
    wave_sync();
#pragma unroll
    for (int j = 0; j <= M; ++j) { h[j] = sh[pad(lane * M + j)]; Q[j] = sq[pad(lane * M + j)]; }
    wave_sync();
    }
  };
  // values of the lane's nodes g0 .. g0 + M - 1 to hd / Qd (nodes beyond the reach are not written)
  auto store_nodes_co = [&](R *hd, R *Qd, int p, const R *hv, const R *Qv) __attribute__((always_inline)) {
    const int wb = p * C + wave * (64 * M);
    R *const sh = sm.stage[wave][0], *const sq = sm.stage[wave][1];
#pragma unroll
    for (int j = 0; j < M; ++j) { sh[pad(lane * M + j)] = hv[j]; sq[pad(lane * M + j)] = Qv[j]; }
    wave_sync();
#pragma unroll
    for (int k = 0; k < M; ++k) {
      const int e = wb + lane + 64 * k;
      const R x = sh[pad(lane + 64 * k)], y = sq[pad(lane + 64 * k)];
      if (e < N) { hd[base + e] = x; Qd[base + e] = y; }
    }
    wave_sync();
  };
  auto kc_at = [&](int i, int p, int c) __attribute__((always_inline)) -> R & { return kcg[(size_t)i * RP + ((size_t)p * M + c) * T + t]; };
  // the four constants of cell (node, node + 1) from the node terms and unknowns of level k (one source for the stored and the recomputed ones)
  auto level_constants = [&](const NodeTerms<R> &L, const NodeTerms<R> &Rn, R h0, R h1, R Q0, R Q1, int cell, R(&k)[4]) __attribute__((always_inline)) {
    const R sumA = L.A + Rn.A;
    k[0] = fma_(cqk, Q1 - Q0, -(sumA * r2dt));
    k[1] = fma_(cqk, fma_(Q1, Rn.v, -(Q0 * L.v)), -((Q1 + Q0) * r2dt));
    k[2] = ghthk * sumA;
    k[3] = fma_(cqk, geo.bed_step(cell) + (h1 - h0), hthk * (L.Se + Rn.Se));
  };
  auto write_level_constants = [&](int p, int g0, const R(&hh)[M + 1], const R(&QQ)[M + 1]) __attribute__((always_inline)) {
    NodeTerms<R> L = node_terms(min(g0, N - 1), hh[0], QQ[0]);
#pragma unroll
    for (int c = 0; c < M; ++c) {
      const NodeTerms<R> Rn = node_terms(min(g0 + c + 1, N - 1), hh[c + 1], QQ[c + 1]);
      R k[4];
      level_constants(L, Rn, hh[c], hh[c + 1], QQ[c], QQ[c + 1], g0 + c, k);
      kc_at(0, p, c) = k[0]; kc_at(1, p, c) = k[1]; kc_at(2, p, c) = k[2]; kc_at(3, p, c) = k[3];
      L = Rn;
    }
  };

  // ---- level constants of the entry level from the accepted state ----
  for (int p = 0; p < P; ++p) {
    const int g0 = p * C + t * M;
    if (p * C > NC) break;
    R hk[M + 1], Qk[M + 1];
    load_nodes(a.hk, a.Qk, g0, hk, Qk);
    if (p == pD && isD) {
#pragma unroll
      for (int j = 0; j < M; ++j) if (j == jD) QoldD = Qk[j];
    }
    if constexpr (!kRecompute) write_level_constants(p, g0, hk, Qk);
  }
  __syncthreads();

  constexpr bool kPf = FS_LONG_PREFETCH != 0 && kCo;
  int pf[4] = {0, 0, 0, 0};
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
      if (it - 1 >= max_it) { status = FS_MAX_ITER; break; }
      R nrm2 = R(0);
      for (int phase = 0; phase < 2; ++phase) {
        for (int p = 0; p < P; ++p) {
          const int g0 = p * C + t * M;                                  // first row / node of this lane in this pass
          if (p * C > NC) {                                              // nothing but padding: identity segments
            if (phase == 0 && lane == 63) {
              R *q = sm.xseg[p * W + wave];
              q[0] = R(0); q[1] = R(0); q[2] = R(0); q[3] = R(0); q[4] = R(1); q[5] = R(0); q[6] = R(0); q[7] = R(0);
              sm.xg[p * W + wave] = 0;
            }
            continue;
          }
          R h[M + 1], Q[M + 1];
          if constexpr (kCo) load_nodes_co(a.hg, a.Qg, p, h, Q); else load_nodes(a.hg, a.Qg, g0, h, Q);
          R hk_[kRecompute ? M + 1 : 1], Qk_[kRecompute ? M + 1 : 1];     // accepted state of level k at the lane's nodes
          if constexpr (kRecompute) { if constexpr (kCo) load_nodes_co(a.hk, a.Qk, p, hk_, Qk_); else load_nodes(a.hk, a.Qk, g0, hk_, Qk_); }
          if constexpr (kPf) {
            // the lines of the next pass's state, requested a pass ahead (one dword per 64 bytes, results unused): when the
            // pass gets there its loads find them in L2.  The four registers stay reserved until the loads have landed -
            // the real loads above are younger than the last pass's requests and have been waited for.
            asm volatile("" :: "v"(pf[0]), "v"(pf[1]), "v"(pf[2]), "v"(pf[3]));
            int pn = p + 1;
            if (pn >= P || pn * C > NC) pn = phase == 0 ? 0 : -1;
            if (pn >= 0) {
              const size_t o = base + min(pn * C + wave * (64 * M) + lane * 8, N - 1);
              pf[0] = *reinterpret_cast<const volatile int *>(a.hg + o); pf[1] = *reinterpret_cast<const volatile int *>(a.Qg + o);
              if constexpr (kRecompute) { pf[2] = *reinterpret_cast<const volatile int *>(a.hk + o); pf[3] = *reinterpret_cast<const volatile int *>(a.Qk + o); }
            }
          }
          if (phase == 1) __syncthreads();          // every lane holds its nodes before any lane stores updated ones

          // ---- boundary rows (boundary.py:56-242) ----
          BCRow<R> Urow, Drow;
          Urow.dh = R(1); Urow.dq = R(0); Urow.res = R(0);
          Drow.dh = R(1); Drow.dq = R(0); Drow.res = R(0);
          if (p == 0 && t == 0) {
            R dummy; int flag = 0;
            Urow = geo.template boundary<BCK, 0>(usd, reach, a.B, level, 0, h[0], Q[0], R(0), dt, R(0), &dummy, &flag);
            if (phase == 0) nrm2 += Urow.res * Urow.res;
          }
          if (p == pD && isD) {
            R hD = h[0], QD = Q[0];
            int flag = 0;
#pragma unroll
            for (int j = 1; j < M; ++j) if (j == jD) { hD = h[j]; QD = Q[j]; }
            Drow = geo.template boundary<BCK, 1>(dsd, reach, a.B, level, N - 1, hD, QD, QoldD, dt, Yprev, &Ynew, &flag);
            if (phase == 0) nrm2 += Drow.res * Drow.res;
            if (flag) sm.xflag = flag;
          }

          // ---- local assembly + fold (the short kernel's, ragged form) ----
          LocalElim<R> el[M - 1];
          R iTn[M + 1];
          Seg<R> seg;
          seg.u1 = R(0); seg.u3 = R(-1); seg.ru = R(0);
          R rcLast;
          {
            NodeTerms<R> L = node_terms(min(g0, N - 1), h[0], Q[0]);
            R i2tL = dt * L.rT;
            iTn[0] = i2tL;
            if (p == 0 && t == 0 && phase == 0) {
              const R x = Urow.dh * i2tL, y = Urow.dq * i2c;
              sm.xbc[0] = x + y; sm.xbc[1] = x - y; sm.xbc[2] = -Urow.res;
            }
            R rcPrev = R(0);
            NodeTerms<R> Lk;
            if constexpr (kRecompute) Lk = node_terms(min(g0, N - 1), hk_[0], Qk_[0]);
#pragma unroll
            for (int c = 0; c < M; ++c) {
              R kk[4];
              if constexpr (kRecompute) {
                const NodeTerms<R> Rk = node_terms(min(g0 + c + 1, N - 1), hk_[c + 1], Qk_[c + 1]);
                level_constants(Lk, Rk, hk_[c], hk_[c + 1], Qk_[c], Qk_[c + 1], g0 + c, kk);
                Lk = Rk;
              } else {
                kk[0] = kc_at(0, p, c); kk[1] = kc_at(1, p, c); kk[2] = kc_at(2, p, c); kk[3] = kc_at(3, p, c);
              }
              const R k0 = kk[0], k1 = kk[1], k2 = kk[2], k3 = kk[3];
              const NodeTerms<R> Rn = node_terms(min(g0 + c + 1, N - 1), h[c + 1], Q[c + 1]);
              const R i2tR = dt * Rn.rT;
              iTn[c + 1] = i2tR;
              Row<R> row;
              {
                const R sumA = L.A + Rn.A;
                const R Cres = sumA * r2dt + cq * (Q[c + 1] - Q[c]) + k0;
                const R gA = fma_(ghth, sumA, k2);
                const R S = cq * (geo.bed_step(g0 + c) + (h[c + 1] - h[c])) + hth * (L.Se + Rn.Se) + k3;
                const R Mres = (Q[c + 1] + Q[c]) * r2dt + cq * (Q[c + 1] * Rn.v - Q[c] * L.v) + k1 + gA * S;
                const R gAdt = gA * dt, gAx = gA * hx, sdt = ghdt * S;
                const R X0 = fma_(gAdt, fma_(hth, L.eAT, -(cq * L.rT)), fma_(dtcq, L.v * L.v, sdt));
                const R X1 = fma_(gAdt, fma_(hth, Rn.eAT, cq * Rn.rT), fma_(-dtcq, Rn.v * Rn.v, sdt));
                const R Y0 = fma_(gAx, L.eQ, kap - L.v);
                const R Y1 = fma_(gAx, Rn.eQ, kap + Rn.v);
                const R ga = X1 + Y1;
                row.al = X0 + Y0; row.D = (X0 - Y0) - ga; row.de = X1 - Y1;
                row.rho0 = fma_(ga, Cres, -Mres);
                row.rc = -Cres;
                R r2 = fma_(Cres, Cres, Mres * Mres);
                const int k = g0 + c;
                if (k >= NC) {                         // the downstream boundary row, then identity rows
                  const bool bcr = k == NC;
                  const R x = Drow.dh * i2tL, y = Drow.dq * i2c;
                  row.al = bcr ? x + y : R(0); row.D = bcr ? x - y : R(1); row.de = R(0);
                  row.rho0 = bcr ? -Drow.res : R(0); row.rc = R(0);
                  r2 = R(0);
                }
                if (phase == 0) nrm2 += r2;
              }
              if (c == 0) {
                seg.d1 = row.al; seg.d2 = row.D; seg.d3 = row.de; seg.rd = row.rho0;
              } else {
                const R r = frcp(seg.d2);
                const R R1 = seg.d1 * r, R2 = seg.d3 * r, R3 = seg.rd * r;
                LocalElim<R> &e = el[c - 1];
                e.R1.put(R1); e.R2.put(R2); e.R3.put(R3); e.qc.put(rcPrev);
                const R rho = fma_(-row.al, rcPrev, row.rho0);
                seg.d1 = row.al * R1; seg.d2 = fma_(row.al, R2, row.D); seg.d3 = row.de; seg.rd = fma_(row.al, R3, rho);
                seg.u1 = fma_(-seg.u3, R1, seg.u1); seg.ru = fma_(-seg.u3, R3, seg.ru); seg.u3 = -(seg.u3 * R2);
              }
              rcPrev = row.rc;
              L = Rn; i2tL = i2tR;
              __builtin_amdgcn_sched_barrier(0);
            }
            seg.rc = rcPrev; rcLast = rcPrev;
          }
          const R upU1 = seg.u1, upU3 = seg.u3, upRu = seg.ru;
          int gi = hi_abs(seg.u3);

          // ---- in-wave tree, up ----
          auto up_level = [&](auto lc) __attribute__((always_inline)) {
            constexpr int l = decltype(lc)::value;
            constexpr int d = 1 << l;
            const Seg<R> left = seg_from_below<d>(seg);
            Seg<R> mg; Elim<R> e;
            merge(left, seg, mg, e);
            if ((lane & (2 * d - 1)) == (2 * d - 1)) {
              R *q = &sm.tree[wave][0][(64 - (64 >> l)) + (lane >> (l + 1))];
              q[0 * 64] = e.A1; q[1 * 64] = e.A2; q[2 * 64] = e.A3; q[3 * 64] = e.rc;
            }
            seg = mg;
            gi = max_(max_(tree_from_below<d>(gi), gi), hi_abs(mg.u3));
          };
          up_level(std::integral_constant<int, 0>{}); up_level(std::integral_constant<int, 1>{});
          up_level(std::integral_constant<int, 2>{}); up_level(std::integral_constant<int, 3>{});
          up_level(std::integral_constant<int, 4>{}); up_level(std::integral_constant<int, 5>{});

          if (phase == 0) {
            if (lane == 63) {
              R *q = sm.xseg[p * W + wave];
              q[0] = seg.u1; q[1] = seg.u3; q[2] = seg.ru; q[3] = seg.d1; q[4] = seg.d2; q[5] = seg.d3; q[6] = seg.rd; q[7] = seg.rc;
              sm.xg[p * W + wave] = gi;
            }
            continue;
          }

          // ---- sweep 1: way down, back-substitution, acceptance, update ----
          // the records of this pass's tree were written by this wave's own lanes just above: LDS executes a wave's
          // instructions in order, the fence keeps the compiler from hoisting the reads over the predicated stores
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const R *xr = sm.xres[p * W + wave];
          R pL = xr[0], mR = xr[1];
          auto down_level = [&](auto lc) __attribute__((always_inline)) {
            constexpr int l = decltype(lc)::value;
            const R *q = &sm.tree[wave][0][(64 - (64 >> l)) + (lane >> (l + 1))];
            Elim<R> e;
            e.A1 = q[0 * 64]; e.A2 = q[1 * 64]; e.A3 = q[2 * 64]; e.rc = q[3 * 64];
            const R sep = separator(e, pL, mR);
            const bool upper = ((lane >> l) & 1) != 0;
            pL = upper ? e.rc - sep : pL;
            mR = upper ? mR : sep;
          };
          down_level(std::integral_constant<int, 5>{}); down_level(std::integral_constant<int, 4>{});
          down_level(std::integral_constant<int, 3>{}); down_level(std::integral_constant<int, 2>{});
          down_level(std::integral_constant<int, 1>{}); down_level(std::integral_constant<int, 0>{});
          R mA = fma_(-upU1, pL, fma_(-upU3, mR, upRu));
          if (lane == 0) mA = xr[2];
          R mB = dpp_mov<0x134>(mA);                // wave_rol:1
          if (lane == 63) mB = xr[3];
          R dh[M + 1], dQ[M + 1];
          {
            auto rc_of = [&](int j) __attribute__((always_inline)) { return j + 1 < M ? el[j].qc.get() : rcLast; };
            R mj = mR;
            {
              const R pM = rcLast - mR;
              dh[M] = (pM + mB) * iTn[M]; dQ[M] = (pM - mB) * i2c;
            }
#pragma unroll
            for (int j = M - 1; j >= 1; --j) {
              const R mprev = j == 1 ? mA : fma_(-el[j - 1].R2.get(), mj, fma_(-el[j - 1].R1.get(), pL, el[j - 1].R3.get()));
              const R pj = rc_of(j - 1) - mprev;
              dh[j] = (pj + mj) * iTn[j]; dQ[j] = (pj - mj) * i2c;
              mj = mprev;
            }
            dh[0] = (pL + mA) * iTn[0]; dQ[0] = (pL - mA) * i2c;
          }
          if (converged) {                           // the pre-update iterate is the level's result
            if (p == 0 && t == 0) {
              a.hydro[((size_t)level * 4 + 0) * a.B + reach] = h[0];
              a.hydro[((size_t)level * 4 + 1) * a.B + reach] = Q[0];
              a.iters[(size_t)level * a.B + reach] = it;
            }
            const bool last = step == a.n_steps - 1;
            R *const hh_p = a.hist_h ? a.hist_h + ((size_t)level * a.B + reach) * NS + g0 : nullptr;
            R *const hQ_p = a.hist_h ? a.hist_Q + ((size_t)level * a.B + reach) * NS + g0 : nullptr;
            if constexpr (kCo) { if (last || kRecompute) store_nodes_co(a.hk, a.Qk, p, h, Q); }
#pragma unroll
            for (int j = 0; j < M; ++j) {
              if (g0 + j < N) {
                if (!kCo && (last || kRecompute)) { a.hk[base + g0 + j] = h[j]; a.Qk[base + g0 + j] = Q[j]; }
                if (hh_p) { hh_p[j] = h[j]; hQ_p[j] = Q[j]; }
              }
            }
            if (p == pD && isD) {
#pragma unroll
              for (int j = 0; j < M; ++j)
                if (j == jD) {
                  a.hydro[((size_t)level * 4 + 2) * a.B + reach] = h[j];
                  a.hydro[((size_t)level * 4 + 3) * a.B + reach] = Q[j];
                  QoldD = Q[j];
                }
              Yprev = Ynew;
              if (ds_storage) a.stage_hist[(size_t)level * a.B + reach] = Ynew;
            }
            if constexpr (!kRecompute) write_level_constants(p, g0, h, Q);
          }
          if constexpr (kCo) {
            R hn[M], Qn[M];
#pragma unroll
            for (int j = 0; j < M; ++j) { hn[j] = h[j] + dh[j]; Qn[j] = Q[j] + dQ[j]; }
            store_nodes_co(a.hg, a.Qg, p, hn, Qn);
          } else {
#pragma unroll
          for (int j = 0; j < M; ++j)
            if (g0 + j < N) { a.hg[base + g0 + j] = h[j] + dh[j]; a.Qg[base + g0 + j] = Q[j] + dQ[j]; }
          }
        }   // passes

        if (phase == 0) {
          // ---- between the sweeps: the top tree over the (pass, wave) segments, by wave 0 ----
          nrm2 = wave_sum(nrm2);
          if (lane == 63) sm.xnorm[wave] = nrm2;
          __syncthreads();
          if (wave == 0) {
            const int S = P * W;
            const int sl = lane < S ? lane : 0;
            const R *q = sm.xseg[sl];
            Seg<R> xs;
            xs.u1 = q[0]; xs.u3 = q[1]; xs.ru = q[2]; xs.d1 = q[3]; xs.d2 = q[4]; xs.d3 = q[5]; xs.rd = q[6]; xs.rc = q[7];
            int gx = sm.xg[sl];
            if (lane >= S) { xs.u1 = R(0); xs.u3 = R(0); xs.ru = R(0); xs.d1 = R(0); xs.d2 = R(1); xs.d3 = R(0); xs.rd = R(0); xs.rc = R(0); gx = 0; }
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
            };
            xup(std::integral_constant<int, 0>{}); xup(std::integral_constant<int, 1>{}); xup(std::integral_constant<int, 2>{});
            xup(std::integral_constant<int, 3>{}); xup(std::integral_constant<int, 4>{}); xup(std::integral_constant<int, 5>{});
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            R p0, m0, ml;
            close_root(xs, sm.xbc[0], sm.xbc[1], sm.xbc[2], p0, m0, ml);        // valid in lane 63
            R px = read_lane(p0, 63), mx = read_lane(ml, 63);
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
            xdown(std::integral_constant<int, 5>{}); xdown(std::integral_constant<int, 4>{}); xdown(std::integral_constant<int, 3>{});
            xdown(std::integral_constant<int, 2>{}); xdown(std::integral_constant<int, 1>{}); xdown(std::integral_constant<int, 0>{});
            const R ma = fma_(-u1o, px, fma_(-u3o, mx, ruo));
            R mb = dpp_mov<0x134>(ma);              // wave_rol:1 : m of the next segment's first row
            if (lane == 63) mb = R(0);
            if (lane < S) { R *o = sm.xres[lane]; o[0] = px; o[1] = mx; o[2] = ma; o[3] = (lane == S - 1) ? R(0) : mb; }
            R tot = R(0);
#pragma unroll
            for (int w = 0; w < W; ++w) tot += sm.xnorm[w];
            // (read in uniform control flow: inside a branch only lane 0 takes, the compiler may compute gx for lane 0 alone)
            const int gtop = __builtin_amdgcn_readlane(gx, 63);
            if (lane == 0) {
              sm.xtot = tot;
              sm.xwarn = gtop > growth_limit_bits<R>() ? 1 : 0;
            }
          }
          __syncthreads();
          const R tot = sm.xtot;
          if (sm.xflag != 0) status = sm.xflag;
          if (!(tot <= finite_max<R>())) status = FS_NAN;
          const R err = sqrt_(tot);
          if (a.trace && t == 0 && it <= FS_TRACE_CAP) a.trace[((size_t)level * FS_TRACE_CAP + (it - 1)) * a.B + reach] = err;
          converged = status == FS_OK && err < tol_r;
          if (converged && sm.xwarn != 0) warn = true;          // (the system at the accepted iterate: see the step kernel)
        }
      }   // sweeps
      __syncthreads();                              // this iteration's stores before the next iteration's loads
    }
    if (status != FS_OK && t == 0) a.iters[(size_t)level * a.B + reach] = it - (status == FS_MAX_ITER ? 1 : 0);
    if (kBudget && a.iter_budget > 0) {
      if (t == 0) a.it_done[reach] = (converged || status != FS_OK) ? -1 : it;
      if (!converged) break;                  // budget spent: the Newton vector is in hg / Qg already
    }
  }
  if (t == 0) a.status[reach] = (status == FS_OK && warn) ? (int)FS_ILL_CONDITIONED : status;
  if (ds_storage && t == tD) a.Yprev[reach] = Yprev;
}

}  // namespace fs
