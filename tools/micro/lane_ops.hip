// validates the DPP / permlane-swap cross-lane fetches used by the in-wave tree against __shfl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../include/flowsim_abi.h"
#include "../../flow-sim_amd/csrc/fs_device.hpp"
using namespace fs;
__global__ void k(double *out) {
  const int lane = threadIdx.x;
  const double v = 1000.0 + lane * 1.25;
  int c = 0;
  auto put = [&](double x) { out[(c++) * 64 + lane] = x; };
  put(fetch_left<2>(v, lane));  put(__shfl_up(v, 2, 64));
  put(fetch_left<4>(v, lane));  put(__shfl_up(v, 4, 64));
  put(fetch_left<8>(v, lane));  put(__shfl_up(v, 8, 64));
  put(fetch_left<16>(v, lane)); put(__shfl_up(v, 16, 64));
  put(fetch_left<32>(v, lane)); put(__shfl_up(v, 32, 64));
  put(tree_from_above<16>(v));  put(__shfl_down(v, 16, 64));
  put(tree_from_above<32>(v));  put(__shfl_down(v, 32, 64));
  put(wave_shr1(v));            put(__shfl_up(v, 1, 64));
  double s = v; for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  put(wave_sum(v)); put(s);
}
int main() {
  double *d; hipMalloc(&d, 18 * 64 * 8);
  k<<<1, 64>>>(d); hipDeviceSynchronize();
  std::vector<double> h(18 * 64); hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  const char *names[] = {"left2", "left4", "left8", "left16", "left32", "above16", "above32", "shr1", "sum"};
  const int need[] = {2, 4, 8, 16, 32, 16, 32, 1, 0};
  int bad = 0;
  for (int t = 0; t < 9; ++t) {
    int nb = 0;
    for (int l = 0; l < 64; ++l) {
      bool relevant;
      if (t < 5) relevant = (l & (need[t] - 1)) == need[t] - 1 && l >= need[t];
      else if (t < 7) relevant = (l & (2 * need[t] - 1)) == need[t] - 1;
      else if (t == 7) relevant = l >= 1;
      else relevant = true;
      const double a = h[(2 * t) * 64 + l], b = h[(2 * t + 1) * 64 + l];
      if (relevant && (t == 8 ? fabs(a - b) > 1e-9 * fabs(b) : a != b)) { if (nb < 3) printf("  %s lane %d: got %g want %g\n", names[t], l, a, b); ++nb; }
    }
    printf("%s: %s\n", names[t], nb ? "MISMATCH" : "ok"); bad += nb;
  }
  return bad != 0;
}
