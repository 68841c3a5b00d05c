// fs_ensemble.hpp - the step kernel for SHORT reaches with general sections (ensembles: BASELINE configs[3], 121 nodes).
//
// Same Newton iteration, same rows, same segment algebra as fs_kernel.hpp - a different use of the machine.  The
// unrolled kernel keeps a lane's nodes in registers, which the general section code (compound trapezoids, curvature)
// can only afford at two rows per lane: one reach per wave, a 6-level tree, two boundary lanes and every per-iteration
// fixed cost paid for ONE member.  Here
//   * a wave carries G members (G = 2: 32 lanes x M rows each), every lane-wide instruction - tree levels, boundary rows,
//     norm, closure - serves G members at once, and each group runs its OWN time loop (no waiting for the slowest
//     member of a level: the wave iterates until every group has done its levels);
//   * the lane's nodes, level constants and elimination records live in LDS (lane-minor, conflict free) and the
//     rows are a rolled loop: the section code exists once, the kernel needs ~1/2 of the registers, 6 waves fit a CU;
//   * every node is stored once (by its owner); a lane reads its last row's right node from the next lane's slot, so
//     there are no shared copies to keep consistent;
//   * the launch primes itself through the loop's own acceptance block (level constants of the first level), so chunked
//     stepping is bitwise one launch by construction.
// Batches without history / trace only (the dispatch in fs_abi.hip sends those to the unrolled kernels).
#pragma once
#include "fs_kernel.hpp"

namespace fs {

template <typename R, int M, int G> struct EnsSmem {
  R h[M][64], Q[M][64];    // the lane's nodes (node s0 + c in slot c)
  R iT[M][64];             // dt / T of those nodes as the last fold saw them (back-substitution)
  R kc[4][M][64];          // level-k constants of the lane's cells
  R rec[3][M][64];         // rows 1..M-1: m_{j-1} = R3 - R1 p_a - R2 m_j
  R rc[M][64];             // rc of the lane's rows (p_{j+1} = rc_j - m_j)
  R tree[4][64];           // records of the in-group tree
  R mail[G][12];           // per group: upstream row (3), root segment (8), squared residual norm
  int32_t flag[G];
};

// sum over each group of LM lanes, valid in the group's last lane
template <int LM, typename R> __device__ __forceinline__ R group_sum_last(R v) {
  v += dpp_zero<0x111, 0xF, 0xF>(v);        // row_shr:1
  v += dpp_zero<0x112, 0xF, 0xF>(v);        // row_shr:2
  v += dpp_zero<0x114, 0xF, 0xF>(v);        // row_shr:4
  v += dpp_zero<0x118, 0xF, 0xF>(v);        // row_shr:8   -> lane 15 of a row: the row's sum
  if (LM >= 32) v += dpp_zero<0x142, 0xA, 0xF>(v);        // row_bcast15 into rows 1 and 3
  if (LM >= 64) v += dpp_zero<0x143, 0xC, 0xF>(v);        // row_bcast31 into rows 2 and 3
  return v;
}

template <typename R, int SEC, int M, int G, int BCK>
__global__ __launch_bounds__(64, 2) void ensemble_step_kernel(const KernelArgs<R> a) {
  static_assert(M >= 2 && (G == 1 || G == 2 || G == 4), "rows per lane >= 2, 64 / G lanes per member");
  constexpr int LM = 64 / G;                       // lanes per member
  constexpr int LV = LM == 64 ? 6 : LM == 32 ? 5 : 4;   // levels of the in-group tree
  using Geo = Geometry<R, SEC>;
  __shared__ EnsSmem<R, M, G> sm;

  const int lane = threadIdx.x;
  const int grp = lane / LM, tl = lane - grp * LM;      // group (member) of this lane, lane within the group
  const int member = blockIdx.x * G + grp;
  const bool live = member < a.B;
  const int reach = live ? member : a.B - 1;             // (groups beyond the batch run along on a copy and store nothing)
  const int N = a.N, NC = N - 1;
  const int s0 = tl * M;
  const int tD = NC / M, jD = NC - tD * M;               // lane / slot of node N-1 = row of the downstream boundary
  const size_t base = (size_t)reach * N;
  const int nxt = lane < 63 ? lane + 1 : 63;             // the lane whose first node is this lane's node M

  Geo geo;
  geo.init(a, reach);
  const R th = a.theta, dt = a.dt;
  const R r2dt = R(1) / (R(2) * dt), cq = th / a.dx, cqk = (R(1) - th) / a.dx, hth = R(0.5) * th, hthk = R(0.5) * (R(1) - th);
  const R g = R(kG);
  const R i2c = R(0.5) / cq, kap = r2dt * i2c, dtcq = dt * cq, ghx = g * hth * i2c, ghdt = g * hth * dt;

  BCDesc<R> usd = a.us, dsd = a.ds;
  const bool ds_storage = BCK >= 2 ? bc_is_storage(BCK - 2) : bc_is_storage(a.ds.kind);
  R Yprev = (ds_storage && tl == tD) ? a.Yprev[reach] : R(0);
  R Ynew = Yprev, QoldD = R(0);
  int status = a.status[reach];
  int level = a.level0, done = 0, it = 0;
  bool act = live && status == FS_OK;                     // this group still has levels to do in this launch
  bool primed = false;

  // (h, Q) <- accepted state of the entry level: the priming pass below runs it through the acceptance block
#pragma unroll 1
  for (int c = 0; c < M; ++c) {
    const int node = min(s0 + c, N - 1);
    sm.h[c][lane] = a.hk[base + node]; sm.Q[c][lane] = a.Qk[base + node];
  }
  if (tl == 0) sm.flag[grp] = 0;

  bool prime = true;
  while (true) {
    if (!prime && __builtin_amdgcn_ballot_w64(act) == 0) break;
    __syncthreads();                                      // nodes updated by their owners are read by the lane before
    bool converged = false;
    R pL = R(0), mR = R(0), mA = R(0);
    const int lv = level + 1;                             // the level this group is solving
    if (!prime) {
      if (act && it >= a.max_iter) {                      // preissmann.py:124-126
        status = FS_MAX_ITER; act = false;
        if (tl == 0) a.iters[(size_t)lv * a.B + reach] = it;
      }
      it += act ? 1 : 0;
      if (it == 1 && act) {                               // a new level: its boundary targets
        if (usd.target) usd.tgt = usd.target[(size_t)lv * a.B + reach];
        if (dsd.target) dsd.tgt = dsd.target[(size_t)lv * a.B + reach];
      }

      // ---- boundary rows (boundary.py:56-242): the group's first lane and the lane of node N-1 ----
      BCRow<R> Urow, Drow;
      Urow.dh = R(0); Urow.dq = R(1); Urow.res = R(0);
      Drow.dh = R(1); Drow.dq = R(0); Drow.res = R(0);
      R nrm2 = R(0);
      int bflag = 0;
      if (tl == 0) {
        R dummy;
        Urow = geo.template boundary<BCK, 0>(usd, reach, a.B, lv, 0, sm.h[0][lane], sm.Q[0][lane], R(0), dt, R(0), &dummy, &bflag);
        nrm2 = Urow.res * Urow.res;
        bflag = 0;
      }
      if (tl == tD) {
        Drow = geo.template boundary<BCK, 1>(dsd, reach, a.B, lv, N - 1, sm.h[jD][lane], sm.Q[jD][lane], QoldD, dt, Yprev, &Ynew, &bflag);
        nrm2 += Drow.res * Drow.res;
        sm.flag[grp] = bflag;
      }

      // ---- assembly + fold: a rolled loop over the lane's rows ----
      NodeTerms<R> L = geo.terms(min(s0, N - 1), sm.h[0][lane], sm.Q[0][lane]);
      NodeTerms<R> Rsh;                                  // node M = the next lane's first node (wave_rol:1)
      {
        auto rol = [](R v) { return dpp_mov<0x134>(v); };
        Rsh.A = rol(L.A); Rsh.T = rol(L.T); Rsh.Se = rol(L.Se); Rsh.eA = rol(L.eA); Rsh.eQ = rol(L.eQ); Rsh.v = rol(L.v); Rsh.rT = rol(L.rT);
      }
      R i2tL = dt * L.rT;
      sm.iT[0][lane] = i2tL;
      if (tl == 0) {                                      // upstream row on (p_0, m_0): aU p_0 + bU m_0 = -res
        const R x = Urow.dh * i2tL, y = Urow.dq * i2c;
        sm.mail[grp][0] = x + y; sm.mail[grp][1] = x - y; sm.mail[grp][2] = -Urow.res;
      }
      Seg<R> seg;
      seg.u1 = R(0); seg.u3 = R(-1); seg.ru = R(0); seg.d1 = R(0); seg.d2 = R(1); seg.d3 = R(0); seg.rd = R(0);
      R rcPrev = R(0);
      R hc = sm.h[0][lane], Qc = sm.Q[0][lane];
#pragma unroll 1
      for (int c = 0; c < M; ++c) {
        const R k0 = sm.kc[0][c][lane], k1 = sm.kc[1][c][lane], k2 = sm.kc[2][c][lane], k3 = sm.kc[3][c][lane];
        R hn, Qn;
        NodeTerms<R> Rn;
        if (c < M - 1) {
          hn = sm.h[c + 1][lane]; Qn = sm.Q[c + 1][lane];
          Rn = geo.terms(min(s0 + c + 1, N - 1), hn, Qn);
        } else {
          hn = sm.h[0][nxt]; Qn = sm.Q[0][nxt];
          Rn = Rsh;
        }
        const R i2tR = dt * Rn.rT;
        if (c < M - 1) sm.iT[c + 1][lane] = i2tR;
        Row<R> row;
        {
          const R sumA = L.A + Rn.A;
          const R Cres = sumA * r2dt + cq * (Qn - Qc) + k0;                                   // preissmann.py:220-249
          const R avgA = hth * sumA + k2;
          const R S = cq * (geo.bed_step(s0 + c) + (hn - hc)) + hth * (L.Se + Rn.Se) + k3;
          const R gA = g * avgA;
          const R Mres = (Qn + Qc) * r2dt + cq * (Qn * Rn.v - Qc * L.v) + k1 + gA * S;      // :251-301
          const R gAdt = gA * dt, gAx = avgA * ghx, sdt = ghdt * S;
          const R X0 = fma_(gAdt, L.rT * fma_(hth, L.eA, -cq), fma_(dtcq, L.v * L.v, sdt));       // :558-612 scaled by dt/T0
          const R X1 = fma_(gAdt, Rn.rT * fma_(hth, Rn.eA, cq), fma_(-dtcq, Rn.v * Rn.v, sdt));    // :496-550 scaled by dt/T1
          const R Y0 = fma_(gAx, L.eQ, kap - L.v);                                                 // :677-733 / (2cq)
          const R Y1 = fma_(gAx, Rn.eQ, kap + Rn.v);                                               // :619-675 / (2cq)
          const R ga = X1 + Y1;
          row.al = X0 + Y0; row.D = (X0 - Y0) - ga; row.de = X1 - Y1;
          row.rho0 = fma_(ga, Cres, -Mres);
          row.rc = -Cres;
          R r2 = fma_(Cres, Cres, Mres * Mres);
          const int k = s0 + c;
          if (k >= NC) {                                   // the downstream boundary row, then identity rows
            const bool bcr = k == NC;
            const R x = Drow.dh * i2tL, y = Drow.dq * i2c;
            row.al = bcr ? x + y : R(0); row.D = bcr ? x - y : R(1); row.de = R(0);
            row.rho0 = bcr ? -Drow.res : R(0); row.rc = R(0);
            r2 = R(0);
          }
          nrm2 += r2;
        }
        if (c == 0) {
          seg.d1 = row.al; seg.d2 = row.D; seg.d3 = row.de; seg.rd = row.rho0;
        } else {
          const R r = frcp(seg.d2);
          const R R1 = seg.d1 * r, R2 = seg.d3 * r, R3 = seg.rd * r;
          sm.rec[0][c][lane] = R1; sm.rec[1][c][lane] = R2; sm.rec[2][c][lane] = R3;
          const R rho = fma_(-row.al, rcPrev, row.rho0);
          seg.d1 = row.al * R1; seg.d2 = fma_(row.al, R2, row.D); seg.d3 = row.de; seg.rd = fma_(row.al, R3, rho);
          seg.u1 = fma_(-seg.u3, R1, seg.u1); seg.ru = fma_(-seg.u3, R3, seg.ru); seg.u3 = -(seg.u3 * R2);
        }
        sm.rc[c][lane] = row.rc;
        rcPrev = row.rc;
        L = Rn; i2tL = i2tR; hc = hn; Qc = Qn;
      }
      seg.rc = rcPrev;
      const R upU1 = seg.u1, upU3 = seg.u3, upRu = seg.ru;

      // ---- in-group tree, up ----
      auto up_level = [&](auto lc) {
        constexpr int l = decltype(lc)::value;
        if constexpr (l < LV) {
          constexpr int d = 1 << l;
          const Seg<R> left = seg_from_below<d>(seg);
          Seg<R> mg; Elim<R> e;
          merge(left, seg, mg, e);
          if ((lane & (2 * d - 1)) == (2 * d - 1)) {
            const int slot = (64 - (64 >> l)) + (lane >> (l + 1));
            sm.tree[0][slot] = e.A1; sm.tree[1][slot] = e.A2; sm.tree[2][slot] = e.A3; sm.tree[3][slot] = e.rc;
          }
          seg = mg;
        }
      };
      up_level(std::integral_constant<int, 0>{}); up_level(std::integral_constant<int, 1>{});
      up_level(std::integral_constant<int, 2>{}); up_level(std::integral_constant<int, 3>{});
      up_level(std::integral_constant<int, 4>{}); up_level(std::integral_constant<int, 5>{});
      nrm2 = group_sum_last<LM>(nrm2);
      if (tl == LM - 1) {
        R *p = sm.mail[grp];
        p[3] = seg.u1; p[4] = seg.u3; p[5] = seg.ru; p[6] = seg.d1; p[7] = seg.d2; p[8] = seg.d3; p[9] = seg.rd; p[10] = seg.rc;
        p[11] = nrm2;
      }
      __syncthreads();

      // ---- close the group's root with its upstream row, convergence test ----
      {
        const R *p = sm.mail[grp];
        Seg<R> S;
        S.u1 = p[3]; S.u3 = p[4]; S.ru = p[5]; S.d1 = p[6]; S.d2 = p[7]; S.d3 = p[8]; S.rd = p[9]; S.rc = p[10];
        R p0, m0, ml;
        close_root(S, p[0], p[1], p[2], p0, m0, ml);
        pL = p0; mR = ml;
        const R err = sqrt_(p[11]);                                      // utility.py:20-22
        if (act) {
          if ((BCK < 2 || ds_storage) && sm.flag[grp] != 0) status = sm.flag[grp];
          if (!(err == err) || !(err <= huge_norm<R>())) status = FS_NAN;
          if (status != FS_OK) {
            act = false;
            if (tl == 0) a.iters[(size_t)lv * a.B + reach] = it;
          }
        }
        converged = act && err < a.tol;                                  // preissmann.py:153
      }

      // ---- separators down the tree ----
      auto down_level = [&](auto lc) {
        constexpr int l = decltype(lc)::value;
        if constexpr (l < LV) {
          const int slot = (64 - (64 >> l)) + (lane >> (l + 1));
          Elim<R> e;
          e.A1 = sm.tree[0][slot]; e.A2 = sm.tree[1][slot]; e.A3 = sm.tree[2][slot]; e.rc = sm.tree[3][slot];
          const R sep = separator(e, pL, mR);
          const bool upper = ((lane >> l) & 1) != 0;
          pL = upper ? e.rc - sep : pL;
          mR = upper ? mR : sep;
        }
      };
      down_level(std::integral_constant<int, 5>{}); down_level(std::integral_constant<int, 4>{});
      down_level(std::integral_constant<int, 3>{}); down_level(std::integral_constant<int, 2>{});
      down_level(std::integral_constant<int, 1>{}); down_level(std::integral_constant<int, 0>{});
      mA = fma_(-upU1, pL, fma_(-upU3, mR, upRu));          // m of the lane's first row from its up row
    } else {
      converged = true;                                      // priming pass: only the acceptance block's level constants
    }

    // ---- accepted iterate -> level k: the state in LDS is still the pre-update one ----
    if (converged) {
      if (!prime) {
        if (tl == 0) {
          a.hydro[((size_t)lv * 4 + 0) * a.B + reach] = sm.h[0][lane];
          a.hydro[((size_t)lv * 4 + 1) * a.B + reach] = sm.Q[0][lane];
          a.iters[(size_t)lv * a.B + reach] = it;
        }
        if (tl == tD) {
          a.hydro[((size_t)lv * 4 + 2) * a.B + reach] = sm.h[jD][lane];
          a.hydro[((size_t)lv * 4 + 3) * a.B + reach] = sm.Q[jD][lane];
          Yprev = Ynew;
          if (ds_storage) a.stage_hist[(size_t)lv * a.B + reach] = Ynew;
        }
        level = lv; done += 1; it = 0;
        if (done == a.n_steps) {                             // the group's last level of this launch: state to HBM
#pragma unroll 1
          for (int c = 0; c < M; ++c)
            if (s0 + c < N) { a.hk[base + s0 + c] = sm.h[c][lane]; a.Qk[base + s0 + c] = sm.Q[c][lane]; }
        }
      }
      if (tl == tD) QoldD = sm.Q[jD][lane];                  // flow[k] of the next level's storage row
      // level constants of the next level from the accepted state (one instance of this code: the priming pass runs here too)
      NodeTerms<R> L = geo.terms(min(s0, N - 1), sm.h[0][lane], sm.Q[0][lane]);
      NodeTerms<R> Rsh;
      {
        auto rol = [](R v) { return dpp_mov<0x134>(v); };
        Rsh.A = rol(L.A); Rsh.Se = rol(L.Se); Rsh.v = rol(L.v);
      }
      R hc = sm.h[0][lane], Qc = sm.Q[0][lane];
#pragma unroll 1
      for (int c = 0; c < M; ++c) {
        R hn, Qn, An, Sen, vn;
        if (c < M - 1) {
          hn = sm.h[c + 1][lane]; Qn = sm.Q[c + 1][lane];
          const NodeTerms<R> Rn = geo.terms(min(s0 + c + 1, N - 1), hn, Qn);
          An = Rn.A; Sen = Rn.Se; vn = Rn.v;
        } else {
          hn = sm.h[0][nxt]; Qn = sm.Q[0][nxt];
          An = Rsh.A; Sen = Rsh.Se; vn = Rsh.v;
        }
        const R sumA = L.A + An;
        sm.kc[0][c][lane] = fma_(cqk, Qn - Qc, -(sumA * r2dt));
        sm.kc[1][c][lane] = fma_(cqk, fma_(Qn, vn, -(Qc * L.v)), -((Qn + Qc) * r2dt));
        sm.kc[2][c][lane] = hthk * sumA;
        sm.kc[3][c][lane] = fma_(cqk, geo.bed_step(s0 + c) + (hn - hc), hthk * (L.Se + Sen));
        L.A = An; L.Se = Sen; L.v = vn; hc = hn; Qc = Qn;
      }
    }

    if (prime) {
      __syncthreads();                                       // every lane has read its neighbour's first node
#pragma unroll 1
      for (int c = 0; c < M; ++c) {                          // Newton start vector of the first level to solve
        const int node = min(s0 + c, N - 1);
        sm.h[c][lane] = a.hg[base + node]; sm.Q[c][lane] = a.Qg[base + node];
      }
      primed = true; prime = false;
      act = act && a.n_steps > 0;
    } else {
      // ---- local back-substitution and update x += delta (preissmann.py:146-147), nodes of this lane only ----
      const bool upd = act || converged;                     // (a group that has just closed its last level still updates)
      __syncthreads();                                       // the acceptance block has read the neighbour's first node
      R mj = mR;
#pragma unroll 1
      for (int j = M - 1; j >= 0; --j) {
        R mprev = mA, pj = pL;
        if (j >= 1) {
          if (j >= 2) mprev = fma_(-sm.rec[1][j][lane], mj, fma_(-sm.rec[0][j][lane], pL, sm.rec[2][j][lane]));
          pj = sm.rc[j - 1][lane] - mprev;
        }
        const R dh = (pj + mj) * sm.iT[j][lane], dQ = (pj - mj) * i2c;
        if (upd && s0 + j < N) { sm.h[j][lane] += dh; sm.Q[j][lane] += dQ; }
        mj = mprev;
      }
      if (converged && done == a.n_steps) act = false;       // this group has done its levels
    }
  }

  // ---- Newton start vector of the next level + per-reach bookkeeping ----
  __syncthreads();
  if (primed && live) {
#pragma unroll 1
    for (int c = 0; c < M; ++c)
      if (s0 + c < N) { a.hg[base + s0 + c] = sm.h[c][lane]; a.Qg[base + s0 + c] = sm.Q[c][lane]; }
  }
  if (live && tl == 0) a.status[reach] = status;
  if (live && ds_storage && tl == tD) a.Yprev[reach] = Yprev;
}

}  // namespace fs
