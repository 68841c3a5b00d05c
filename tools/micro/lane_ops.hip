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
  put(tree_from_below<1>(v));  put(__shfl_up(v, 1, 64));
  put(tree_from_below<2>(v));  put(__shfl_up(v, 2, 64));
  put(tree_from_below<4>(v));  put(__shfl_up(v, 4, 64));
  put(tree_from_below<8>(v));  put(__shfl_up(v, 8, 64));
  put(tree_from_below<16>(v)); put(__shfl_up(v, 16, 64));
  put(tree_from_below<32>(v)); put(__shfl_up(v, 32, 64));
  put(dpp_mov<0xF5>(v));       put(__shfl(v, lane | 1, 64));       // quad_perm:[1,1,3,3]
  double s = v; for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  put(wave_sum(v)); put(s);
}
int main() {
  double *d; (void)hipMalloc(&d, 16 * 64 * 8);
  k<<<1, 64>>>(d); (void)hipDeviceSynchronize();
  std::vector<double> h(16 * 64); (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  const char *names[] = {"below1", "below2", "below4", "below8", "below16", "below32", "pair", "sum"};
  const int need[] = {1, 2, 4, 8, 16, 32, 0, 0};
  int bad = 0;
  for (int t = 0; t < 8; ++t) {
    int nb = 0;
    for (int l = 0; l < 64; ++l) {
      // tree_from_below<D>: defined for lanes whose low log2(2D) bits are all ones (and that have a lane - D)
      const bool relevant = t < 6 ? ((l & (2 * need[t] - 1)) == 2 * need[t] - 1) : true;
      const double a = h[(2 * t) * 64 + l], b = h[(2 * t + 1) * 64 + l];
      if (relevant && (t == 7 ? fabs(a - b) > 1e-9 * fabs(b) : a != b)) { if (nb < 3) printf("  %s lane %d: got %g want %g\n", names[t], l, a, b); ++nb; }
    }
    printf("%s: %s\n", names[t], nb ? "MISMATCH" : "ok"); bad += nb;
  }
  return bad != 0;
}
