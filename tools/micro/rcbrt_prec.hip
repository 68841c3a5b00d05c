// accuracy of fs::rcbrt_pos (x^(-1/3)) over 12 decades
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "../../include/flowsim_abi.h"
#include "../../flow-sim_amd/csrc/fs_device.hpp"
__global__ void k(const double *x, double *y, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) y[i] = fs::rcbrt_pos(x[i]); }
int main() {
  const int n = 1 << 20; std::vector<double> x(n), y(n);
  for (int i = 0; i < n; ++i) x[i] = 1e-6 * pow(1e12, (double)i / n) * (1 + 0.31 * sin(i * 0.7));
  double *dx, *dy; (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&dy, n * 8);
  (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, dy, n);
  (void)hipMemcpy(y.data(), dy, n * 8, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int i = 0; i < n; ++i) { const long double t = 1.0L / cbrtl((long double)x[i]); worst = fmax(worst, (double)fabsl((y[i] - t) / t)); }
  printf("rcbrt_pos max relative error over [1e-6, 1e6]: %.3e\n", worst);
  return worst > 1e-15;
}
