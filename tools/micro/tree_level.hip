// cycles of the in-wave up-sweep (6 levels; VARIANT 0 as in the kernel's first version, 1 without record stores,
// 2 every lane merges) and, VARIANT >= 3, of the broadcast down-sweep of fs_kernel.hpp, one wave per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../include/flowsim_abi.h"
#include "../../flow-sim_amd/csrc/fs_device.hpp"
using namespace fs;
#ifndef VARIANT
#define VARIANT 0
#endif
template <int l> __device__ __forceinline__ void up(Seg<double> &seg, int lane, double *slots) {
  constexpr int d = 1 << l;
  const Seg<double> left = seg_from_below<d>(seg);
  Seg<double> mg; Elim<double> e;
  merge(left, seg, mg, e);
#if VARIANT == 2 || VARIANT == 6
  seg = mg;       // every lane: the lanes that do not survive this level are never read again
#endif
#if VARIANT == 6
  {
    const bool sv = (lane & (2 * d - 1)) == (2 * d - 1);
    const int slot = sv ? (64 - (64 >> l)) + (lane >> (l + 1)) : 63;
    double *p = slots + slot;
    p[0 * 64] = e.w10; p[1 * 64] = e.w11; p[2 * 64] = e.w20; p[3 * 64] = e.w21; p[4 * 64] = e.pm0;
    p[5 * 64] = e.pm1; p[6 * 64] = e.qm;  p[7 * 64] = e.sc0; p[8 * 64] = e.sc1; p[9 * 64] = e.qc;
  }
#else
  if ((lane & (2 * d - 1)) == (2 * d - 1)) {
#if VARIANT != 1
    const int slot = (64 - (64 >> l)) + (lane >> (l + 1));
    double *p = slots + slot;
    p[0 * 64] = e.w10; p[1 * 64] = e.w11; p[2 * 64] = e.w20; p[3 * 64] = e.w21; p[4 * 64] = e.pm0;
    p[5 * 64] = e.pm1; p[6 * 64] = e.qm;  p[7 * 64] = e.sc0; p[8 * 64] = e.sc1; p[9 * 64] = e.qc;
#endif
#if VARIANT != 2
    seg = mg;
#endif
  }
#endif
}
template <int l> __device__ __forceinline__ void down_b(double &aL0, double &aL1, double &aR0, double &aR1, int lane, const double *slots) {
  Elim<double> e;
  const int slot = (64 - (64 >> l)) + (lane >> (l + 1));
  const double *p = slots + slot;
  e.w10 = p[0 * 64]; e.w11 = p[1 * 64]; e.w20 = p[2 * 64]; e.w21 = p[3 * 64]; e.pm0 = p[4 * 64];
  e.pm1 = p[5 * 64]; e.qm = p[6 * 64];  e.sc0 = p[7 * 64]; e.sc1 = p[8 * 64]; e.qc = p[9 * 64];
  double m0, m1;
  back(e, aL0, aL1, aR0, aR1, m0, m1);
  const bool upper = ((lane >> l) & 1) != 0;
  aL0 = upper ? m0 : aL0; aL1 = upper ? m1 : aL1;
  aR0 = upper ? aR0 : m0; aR1 = upper ? aR1 : m1;
}
__global__ __launch_bounds__(256, 1) void k(double *out, unsigned long long *cyc, int reps) {
  __shared__ double tree[4][10][64];
  __shared__ double big[16000];            // keep one workgroup per CU like the real kernel
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  big[t] = t;
  Seg<double> s0;
  const double x = 1.0 + 0.001 * t;
  s0.pc0 = 0.4 * x; s0.pc1 = -0.002; s0.sc0 = 0.41 * x; s0.sc1 = 0.002; s0.qc = 1e-3 * x;
  s0.pm0 = -0.5 * x; s0.pm1 = 0.9; s0.sm0 = 0.52 * x; s0.sm1 = 1.1; s0.qm = 2e-3 * x;
  double acc = 0;
  unsigned long long total = 0;
  for (int r = 0; r < reps; ++r) {
    Seg<double> seg = s0;
    seg.qc += acc * 1e-30;
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("" : "+v"(seg.qc));
    up<0>(seg, lane, &tree[wave][0][0]); up<1>(seg, lane, &tree[wave][0][0]); up<2>(seg, lane, &tree[wave][0][0]);
    up<3>(seg, lane, &tree[wave][0][0]); up<4>(seg, lane, &tree[wave][0][0]); up<5>(seg, lane, &tree[wave][0][0]);
    asm volatile("" : "+v"(seg.qc), "+v"(seg.qm), "+v"(seg.pc0), "+v"(seg.sm0));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if VARIANT >= 3
    double dR0 = seg.qc, dR1 = seg.qm;
    const double *sl = &tree[wave][0][0];
    asm volatile("" : "+v"(dR0), "+v"(dR1));
    const unsigned long long t0b = __builtin_amdgcn_s_memtime();
    { double aL0 = 0.1, aL1 = 0.2, aR0 = dR0, aR1 = dR1;
      down_b<5>(aL0, aL1, aR0, aR1, lane, sl); down_b<4>(aL0, aL1, aR0, aR1, lane, sl); down_b<3>(aL0, aL1, aR0, aR1, lane, sl);
      down_b<2>(aL0, aL1, aR0, aR1, lane, sl); down_b<1>(aL0, aL1, aR0, aR1, lane, sl); down_b<0>(aL0, aL1, aR0, aR1, lane, sl);
      dR0 = aR0 + aL0; dR1 = aR1 + aL1; }
    asm volatile("" : "+v"(dR0), "+v"(dR1));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    total += t1 - t0b;
    acc += dR0 + dR1;
#else
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    total += t1 - t0;
    acc += seg.qc + seg.qm + seg.pc0 + seg.sm0;
#endif
  }
  out[blockIdx.x * 256 + t] = acc + tree[wave][3][lane] + big[(t * 7) % 16000];
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = total;
}
int main() {
  const int blocks = 256, reps = 2000;
  double *o; unsigned long long *c;
  hipMalloc(&o, blocks * 256 * 8); hipMalloc(&c, blocks * 4 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<<<blocks, 256>>>(o, c, 10);
  hipEventRecord(e0);
  k<<<blocks, 256>>>(o, c, reps);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 4);
  hipMemcpy(h.data(), c, blocks * 4 * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += v;
  printf("variant %d: up-sweep %.0f memtime ticks per 6 levels; wall %.1f ns per sweep (kernel %.3f ms)\n", VARIANT,
         s / h.size() / reps, ms * 1e6 / reps, ms);
  return 0;
}
