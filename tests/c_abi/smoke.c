/* The C ABI from plain C: include/flowsim_abi.h compiles as C99, the library links, and one small
 * reach (rectangular channel, flow hydrograph in, normal depth out) steps on the GPU when there is one.
 * Built and run by tests/test_abi_c.py.  Exit codes: 0 ok, 2 no device (CPU box), 1 failure. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "flowsim_abi.h"

#define CHECK(x) do { if ((x) != 0) { fprintf(stderr, "%s failed: %s\n", #x, fs_last_error()); return 1; } } while (0)

int main(void) {
  enum { N = 50, NT = 6 };
  if (fs_abi_version() != FS_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
  fs_batch_desc d = {1, N, FS_F64, FS_SEC_RECT_UNIFORM, 0, NT, FS_FLAG_HISTORY, 0};
  if (fs_device_count() == 0) {
    /* no CPU path: creation must fail and say why */
    fs_batch *none = fs_batch_create(&d);
    if (none != NULL) { fprintf(stderr, "fs_batch_create succeeded without a device\n"); return 1; }
    printf("no device: %s\n", fs_last_error());
    return 2;
  }
  fs_batch *b = fs_batch_create(&d);
  if (!b) { fprintf(stderr, "create: %s\n", fs_last_error()); return 1; }
  const double width = 120.0, n = 0.03, S0 = 4e-4, dx = 400.0, Q0 = 300.0;
  const double geo[FS_RU_NPARAM] = {width, n, S0 * dx * (N - 1), 0.0};
  /* normal depth by bisection (Manning) */
  double lo = 1e-6, hi = 50.0;
  for (int i = 0; i < 200; ++i) {
    const double h = 0.5 * (lo + hi), A = width * h, P = width + 2 * h;
    if (A * pow(A / P, 2.0 / 3.0) / n * sqrt(S0) < Q0) lo = h; else hi = h;
  }
  const double hn = 0.5 * (lo + hi);
  double target[NT];
  for (int k = 0; k < NT; ++k) target[k] = Q0 * (1.0 + 0.1 * k);
  const double nd[2] = {S0, 0.0};
  CHECK(fs_batch_set_scheme(b, 0.6, 600.0, dx, 1e-6, 100));
  CHECK(fs_batch_set_geometry_uniform(b, geo));
  CHECK(fs_batch_set_bc(b, FS_UPSTREAM, FS_BC_FLOW_HYDROGRAPH, NULL, 0, 0, target));
  CHECK(fs_batch_set_bc(b, FS_DOWNSTREAM, FS_BC_NORMAL_DEPTH, nd, 2, 0, NULL));
  CHECK(fs_batch_set_state_uniform(b, &hn, &Q0));
  CHECK(fs_batch_step(b, NT - 1));
  int32_t status = -1, iters[NT];
  double h[N], Q[N];
  CHECK(fs_batch_get_status(b, &status));
  CHECK(fs_batch_get_iterations(b, 0, NT, iters));
  CHECK(fs_batch_get_state(b, h, Q));
  fs_batch_destroy(b);
  if (status != FS_OK) { fprintf(stderr, "status %d\n", status); return 1; }
  /* the upstream discharge follows the hydrograph, depths stay near normal depth, Newton converged quickly */
  if (fabs(Q[0] - target[NT - 1]) > 1e-6 * target[NT - 1]) { fprintf(stderr, "Q[0] = %g\n", Q[0]); return 1; }
  for (int i = 0; i < N; ++i) if (!(h[i] > 0.5 * hn && h[i] < 2.0 * hn)) { fprintf(stderr, "h[%d] = %g\n", i, h[i]); return 1; }
  for (int k = 1; k < NT; ++k) if (iters[k] < 1 || iters[k] > 10) { fprintf(stderr, "iters[%d] = %d\n", k, iters[k]); return 1; }
  printf("ok: h0 %.6f hN %.6f Q0 %.3f QN %.3f\n", h[0], h[N - 1], Q[0], Q[N - 1]);
  return 0;
}
