// precision of v_rcp_f64 / v_rsq_f64 seeds on gfx950 (how many Newton steps the kernel needs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2, double* q0, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  double v = x[i];
  double r = __builtin_amdgcn_rcp(v); r0[i] = r;
  double e = __builtin_fma(-v, r, 1.0); r = __builtin_fma(r, e, r); r1[i] = r;
  e = __builtin_fma(-v, r, 1.0); r = __builtin_fma(r, e, r); r2[i] = r;
  q0[i] = __builtin_amdgcn_rsq(v);
}
int main() {
  const int n = 1 << 20; std::vector<double> x(n), a(n), b(n), c(n), d(n);
  for (int i = 0; i < n; ++i) x[i] = 1e-3 * pow(1e7, (double)i / n) * (1 + 0.37 * sin(i));
  double *dx, *d0, *d1, *d2, *d3; hipMalloc(&dx, n*8); hipMalloc(&d0, n*8); hipMalloc(&d1, n*8); hipMalloc(&d2, n*8); hipMalloc(&d3, n*8);
  hipMemcpy(dx, x.data(), n*8, hipMemcpyHostToDevice);
  k<<<n/256, 256>>>(dx, d0, d1, d2, d3, n);
  hipMemcpy(a.data(), d0, n*8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n*8, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), d2, n*8, hipMemcpyDeviceToHost); hipMemcpy(d.data(), d3, n*8, hipMemcpyDeviceToHost);
  double e0=0,e1=0,e2=0,e3=0;
  for (int i = 0; i < n; ++i) { double t = 1.0/x[i];
    e0 = fmax(e0, fabs(a[i]-t)/t); e1 = fmax(e1, fabs(b[i]-t)/t); e2 = fmax(e2, fabs(c[i]-t)/t);
    double s = 1.0/sqrt(x[i]); e3 = fmax(e3, fabs(d[i]-s)/s); }
  printf("v_rcp_f64 max rel err %.3e (2^%.1f); +1 NR %.3e; +2 NR %.3e; v_rsq_f64 %.3e (2^%.1f)\n", e0, log2(e0), e1, e2, e3, log2(e3));
  return 0;
}
