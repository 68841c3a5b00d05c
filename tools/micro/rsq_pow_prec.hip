// accuracy on the device of fs::frsq (1/sqrt(x), third-order correction of the v_rsq_f64 seed), of the raw seed, and of fs::pow_pos (x^b written out)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "../../include/flowsim_abi.h"
#include "../../flow-sim_amd/csrc/fs_device.hpp"
__global__ void k(const double *x, const double *b, double *y, double *s, double *p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { y[i] = fs::frsq(x[i]); s[i] = __builtin_amdgcn_rsq(x[i]); p[i] = fs::pow_pos(x[i], b[i]); }
}
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 2; } } while (0)
int main() {
  const int n = 1 << 20; std::vector<double> x(n), b(n), y(n), s(n), p(n);
  for (int i = 0; i < n; ++i) { x[i] = 1e-6 * pow(1e12, (double)i / n) * (1 + 0.31 * sin(i * 0.7)); b[i] = 0.2 + 4.8 * fabs(sin(i * 1.3)); }
  double *dx, *db, *dy, *ds, *dp;
  CK(hipMalloc(&dx, n * 8)); CK(hipMalloc(&db, n * 8)); CK(hipMalloc(&dy, n * 8)); CK(hipMalloc(&ds, n * 8)); CK(hipMalloc(&dp, n * 8));
  CK(hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice));
  k<<<n / 256, 256>>>(dx, db, dy, ds, dp, n);
  CK(hipGetLastError()); CK(hipDeviceSynchronize());
  CK(hipMemcpy(y.data(), dy, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(s.data(), ds, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(p.data(), dp, n * 8, hipMemcpyDeviceToHost));
  double wy = 0, ws = 0, wp = 0, wl = 0;
  for (int i = 0; i < n; ++i) {
    const long double t = 1.0L / sqrtl((long double)x[i]);
    wy = fmax(wy, (double)fabsl((y[i] - t) / t)); ws = fmax(ws, (double)fabsl((s[i] - t) / t));
    if (x[i] >= 1e-4 && x[i] <= 1e4) {
      const long double q = powl((long double)x[i], (long double)b[i]);
      wp = fmax(wp, (double)fabsl((p[i] - q) / q));
      wl = fmax(wl, (double)fabsl((exp(b[i] * log(x[i])) - q) / q));
    }
  }
  printf("frsq max relative error over [1e-6, 1e6]: %.3e (v_rsq_f64 seed alone: %.3e)\n", wy, ws);
  printf("pow_pos max relative error, x in [1e-4, 1e4], b in [0.2, 5]: %.3e (host libm exp(b log x): %.3e)\n", wp, wl);
  return (wy > 1e-15 || wp > 1e-14) ? 1 : 0;
}
