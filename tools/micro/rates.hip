// Issue cost of the instructions the flagship fold is made of, one wave per SIMD (as the flagship runs): cycles per wave64
// instruction from s_memtime around 8 independent chains x 64 repetitions.   hipcc -O3 --offload-arch=gfx950 -o rates rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define REP 128
#define UNR 32
template <int OP> __global__ __launch_bounds__(256) void k(double *out, unsigned long long *cyc, double seed) {
  __shared__ double lds[256 * 2]; double a[8]; float f[8]; unsigned long long msk = __builtin_amdgcn_readfirstlane(blockIdx.x) * 0x9E3779B97F4A7C15ull + 0x5555555555555555ull; int sg = 0; unsigned long long saved = 0;
  const unsigned ldsaddr = threadIdx.x * 16; lds[threadIdx.x * 2] = seed; asm volatile("v_accvgpr_write_b32 a1, %0" :: "v"((float)seed) : "a1");
  for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 1e-3 + i; f[i] = (float)a[i]; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int r = 0; r < REP; ++r) {
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a[i]) : "v"(seed));
      if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
      if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
      if (OP == 3) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
      if (OP == 4) asm volatile("v_log_f32 %0, %0" : "+v"(f[i]));
      if (OP == 5) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));
      if (OP == 6) asm volatile("v_cvt_f32_f64 %0, %1\n\tv_cvt_f64_f32 %1, %0" : "+v"(f[i]), "+v"(a[i]));      // a pair per chain
      if (OP == 7) asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(a[i]) : "v"(f[i]));
      if (OP == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
      if (OP == 9) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"((float)seed));
      if (OP == 10) {        // a write and a read per chain, eight different accumulation registers
        if (i == 0) asm volatile("v_accvgpr_write_b32 a10, %0\n\tv_accvgpr_read_b32 %0, a10" : "+v"(f[i]) :: "a10");
        if (i == 1) asm volatile("v_accvgpr_write_b32 a11, %0\n\tv_accvgpr_read_b32 %0, a11" : "+v"(f[i]) :: "a11");
        if (i == 2) asm volatile("v_accvgpr_write_b32 a12, %0\n\tv_accvgpr_read_b32 %0, a12" : "+v"(f[i]) :: "a12");
        if (i == 3) asm volatile("v_accvgpr_write_b32 a13, %0\n\tv_accvgpr_read_b32 %0, a13" : "+v"(f[i]) :: "a13");
        if (i == 4) asm volatile("v_accvgpr_write_b32 a14, %0\n\tv_accvgpr_read_b32 %0, a14" : "+v"(f[i]) :: "a14");
        if (i == 5) asm volatile("v_accvgpr_write_b32 a15, %0\n\tv_accvgpr_read_b32 %0, a15" : "+v"(f[i]) :: "a15");
        if (i == 6) asm volatile("v_accvgpr_write_b32 a16, %0\n\tv_accvgpr_read_b32 %0, a16" : "+v"(f[i]) :: "a16");
        if (i == 7) asm volatile("v_accvgpr_write_b32 a17, %0\n\tv_accvgpr_read_b32 %0, a17" : "+v"(f[i]) :: "a17");
      }
      if (OP == 11) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(f[i]));
      if (OP == 12) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[i]));
      if (OP == 13) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"((float)seed) : "vcc");
      if (OP == 14) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(f[i]) : "v"((float)seed), "s"(msk));
      if (OP == 15) asm volatile("v_accvgpr_read_b32 %0, a1\n\tv_mul_f32 %0, %0, %0" : "+v"(f[i]));     // a read and a multiply per chain
      if (OP == 16) asm volatile("v_mov_b32 %0, %0" : "+v"(f[i]));
      if (OP == 17) asm volatile("v_readlane_b32 %0, %1, 7\n\tv_mul_f32 %1, %0, %1" : "=s"(sg), "+v"(f[i]));    // a readlane and a multiply by it
      if (OP == 18) asm volatile("ds_read_b64 %0, %1" : "=v"(a[i]) : "v"(ldsaddr));
      if (OP == 19) asm volatile("ds_write_b64 %0, %1" :: "v"(ldsaddr), "v"(a[i]) : "memory");
      if (OP == 20) asm volatile("v_fma_f64 %0, %0, %1, %0\n\tv_accvgpr_write_b32 a2, %2" : "+v"(a[i]) : "v"(seed), "v"(f[i]) : "a2");
      if (OP == 21) asm volatile("v_fma_f64 %0, %0, %1, %0\n\tds_write_b64 %2, %0" : "+v"(a[i]) : "v"(seed), "v"(ldsaddr) : "memory");
      if (OP == 22) asm volatile("v_mov_b64 %0, %0" : "+v"(a[i]));
      // Partial EXEC around an LDS read.  The statement saves EXEC in a scalar pair of its own, narrows it, reads, and puts the SAVED mask
      // back (not a literal -1), and it tells the compiler what it touches: "+v" (lanes outside the mask keep their old value - with
      // "=v" the compiler was told all 64 lanes are written), the "exec" clobber, and "memory" (LDS).  Round 3's version wrote EXEC
      // behind the compiler's back; see profiles/round4/rates_fault.md for what its ISA showed and what it did not.
      if (OP == 23) asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tds_read_b64 %0, %2\n\ts_mov_b64 exec, %1"
                                 : "+v"(a[i]), "=&s"(saved) : "v"(ldsaddr) : "exec", "memory");           // one lane active
      if (OP == 24) asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 0xffff\n\tds_read_b64 %0, %2\n\ts_mov_b64 exec, %1"
                                 : "+v"(a[i]), "=&s"(saved) : "v"(ldsaddr) : "exec", "memory");           // sixteen lanes active
      if (OP == 25) asm volatile("ds_read_b64 %0, %1" : "=v"(a[i]) : "v"(0u));                                                                // every lane the same address

    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = sg + lds[(threadIdx.x * 2 + 1) & 511]; for (int i = 0; i < 8; ++i) s += a[i] + f[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// every HIP call is checked where it is made: a queue that died in kernel n must not be discovered (or not) in kernel n + 3
#define CK(expr)                                                                                              \
  do {                                                                                                        \
    hipError_t e_ = (expr);                                                                                   \
    if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); std::fflush(stdout); std::exit(2); } \
  } while (0)
template <int OP> double run(const char *name, double *out, unsigned long long *cyc) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256), 0, 0, out, cyc, 1.0000001);
  CK(hipGetLastError());
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(256);
  CK(hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  double m = 0; for (auto v : h) m += v; m /= 256;
  // s_memtime counts at 100 MHz on this family; report relative to v_fma_f64
  printf("%-18s %10.1f s_memtime ticks, %7.3f ms for %d instructions per wave: %.2f ticks, %.2f ns per instruction (%.2f cycles at 2.4 GHz)\n", name, m, ms,
         REP * UNR * 8, m / (REP * UNR * 8), ms * 1e6 / (REP * UNR * 8), ms * 1e6 / (REP * UNR * 8) * 2.4);
  std::fflush(stdout);              // a line per kernel reaches the log before the next kernel is launched
  return m;
}
int main() {
  double *out; unsigned long long *cyc;
  CK(hipMalloc(&out, 256 * 256 * 8)); CK(hipMalloc(&cyc, 256 * 8));
  // where the two buffers lie: a fault address can then be placed next to them or elsewhere (round 3's record had no such line)
  std::printf("out  [%p, %p)\ncyc  [%p, %p)\n", (void *)out, (void *)(out + 256 * 256), (void *)cyc, (void *)(cyc + 256)); std::fflush(stdout);
  run<0>("warm-up", out, cyc);
  double f = run<0>("v_fma_f64", out, cyc);
  const char *n[] = {"", "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_log_f32", "v_exp_f32", "cvt f64>f32>f64 pair", "v_cvt_f64_f32 (same dst)", "v_rcp_f32", "v_mul_f32",
                     "accvgpr write+read pair", "v_mov_b32_dpp", "v_rsq_f64", "v_cndmask_b32 vcc", "v_cndmask_b32 sgpr", "accvgpr_read + mul_f32", "v_mov_b32", "readlane + mul_f32", "ds_read_b64", "ds_write_b64",
                     "fma + accvgpr_write", "fma + ds_write_b64", "v_mov_b64", "ds_read_b64 1 lane", "ds_read_b64 16 lanes", "ds_read_b64 broadcast"};
  double r[26];
  r[1] = run<1>(n[1], out, cyc); r[2] = run<2>(n[2], out, cyc); r[3] = run<3>(n[3], out, cyc); r[4] = run<4>(n[4], out, cyc);
  r[5] = run<5>(n[5], out, cyc); r[6] = run<6>(n[6], out, cyc); r[7] = run<7>(n[7], out, cyc); r[8] = run<8>(n[8], out, cyc);
  r[9] = run<9>(n[9], out, cyc); r[10] = run<10>(n[10], out, cyc); r[11] = run<11>(n[11], out, cyc); r[12] = run<12>(n[12], out, cyc);
  r[13] = run<13>(n[13], out, cyc); r[14] = run<14>(n[14], out, cyc); r[15] = run<15>(n[15], out, cyc); r[16] = run<16>(n[16], out, cyc);
  r[17] = run<17>(n[17], out, cyc); r[18] = run<18>(n[18], out, cyc); r[19] = run<19>(n[19], out, cyc); r[20] = run<20>(n[20], out, cyc); r[21] = run<21>(n[21], out, cyc);
  r[22] = run<22>(n[22], out, cyc); r[23] = run<23>(n[23], out, cyc); r[24] = run<24>(n[24], out, cyc); r[25] = run<25>(n[25], out, cyc);
  printf("\nrelative to v_fma_f64 (one wave per SIMD, 4 waves per CU, 8 independent chains):\n");
  for (int i = 1; i < 26; ++i) printf("  %-26s %.2f\n", n[i], r[i] / f);
  CK(hipDeviceSynchronize()); CK(hipFree(out)); CK(hipFree(cyc));
  std::printf("done: every launch synchronised, every HIP call returned hipSuccess\n");
  return 0;
}
