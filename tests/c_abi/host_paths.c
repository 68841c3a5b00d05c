/* Host-side paths of the C ABI that need no GPU: the dispatch-table accessors, descriptor validation, NULL handles, error
 * texts.  Built by tests/test_sanitizers.py against a host-only AddressSanitizer + UBSan build of fs_abi.hip (clang) and
 * run in the build container: any out-of-bounds access, use after free or undefined behaviour in these paths aborts it.
 * Exit code 0 = every call answered as the header says. */
#include <stdio.h>
#include <string.h>
#include "flowsim_abi.h"

#define EXPECT(cond) do { if (!(cond)) { fprintf(stderr, "line %d: %s\n  last error: %s\n", __LINE__, #cond, fs_last_error()); return 1; } } while (0)

int main(void) {
  EXPECT(fs_abi_version() == FS_ABI_VERSION);
  EXPECT(fs_device_count() >= 0);
  const int n = fs_kernel_table_size();
  EXPECT(n > 0);
  for (int i = 0; i < n; ++i) {
    int32_t e[8];
    EXPECT(fs_kernel_table_entry(i, e) == 0);
    EXPECT((e[0] == FS_F64 || e[0] == FS_F32) && e[1] >= FS_SEC_RECT_UNIFORM && e[1] <= FS_SEC_IRREGULAR);
    EXPECT(e[2] >= 2 && e[3] >= 1 && e[3] <= 8 && (e[4] == 0 || e[4] == 1) && e[5] >= -1 && (e[6] == 0 || e[6] == 1) && (e[7] == 0 || e[7] == 1));
  }
  int32_t e[8];
  EXPECT(fs_kernel_table_entry(-1, e) < 0 && strlen(fs_last_error()) > 0);
  EXPECT(fs_kernel_table_entry(n, e) < 0);
  EXPECT(fs_kernel_table_entry(0, NULL) < 0);

  EXPECT(fs_batch_create(NULL) == NULL && strstr(fs_last_error(), "null descriptor"));
  fs_batch_desc d = {1, 50, FS_F64, FS_SEC_RECT_UNIFORM, 0, 6, 0, 0};
  fs_batch_desc bad = d;
  bad.n_nodes = 1;      EXPECT(fs_batch_create(&bad) == NULL && strstr(fs_last_error(), "n_nodes"));
  bad = d; bad.n_reaches = 0; EXPECT(fs_batch_create(&bad) == NULL);
  bad = d; bad.max_levels = 1; EXPECT(fs_batch_create(&bad) == NULL);
  bad = d; bad.dtype = 7;     EXPECT(fs_batch_create(&bad) == NULL && strstr(fs_last_error(), "dtype"));
  bad = d; bad.section_mode = 9; EXPECT(fs_batch_create(&bad) == NULL && strstr(fs_last_error(), "section_mode"));
  bad = d; bad.section_mode = FS_SEC_IRREGULAR; bad.dtype = FS_F32; EXPECT(fs_batch_create(&bad) == NULL && strstr(fs_last_error(), "fp64"));
  bad = d; bad.device = -1;   EXPECT(fs_batch_create(&bad) == NULL);
  if (fs_device_count() == 0) {
    EXPECT(fs_batch_create(&d) == NULL && strstr(fs_last_error(), "no HIP device"));
    bad = d; bad.n_nodes = 1 << 20; EXPECT(fs_batch_create(&bad) == NULL);
  }

  double x = 0.0; int32_t k = 0;
  EXPECT(fs_batch_set_scheme(NULL, 0.6, 1.0, 1.0, 1e-6, 10) < 0);
  EXPECT(fs_batch_set_geometry_uniform(NULL, &x) < 0);
  EXPECT(fs_batch_set_geometry_table(NULL, &x, NULL) < 0);
  EXPECT(fs_batch_set_geometry_irregular(NULL, &x, &k, 2, &x, &x, &x, NULL) < 0);
  EXPECT(fs_batch_set_bc(NULL, FS_UPSTREAM, FS_BC_FLOW_HYDROGRAPH, NULL, 0, 0, &x) < 0);
  EXPECT(fs_batch_set_state(NULL, &x, &x) < 0);
  EXPECT(fs_batch_set_state_uniform(NULL, &x, &x) < 0);
  EXPECT(fs_batch_step(NULL, 1) < 0 && fs_batch_sync(NULL) < 0 && fs_batch_iterate(NULL, &k) < 0);
  EXPECT(fs_batch_set_host_rows(NULL, FS_UPSTREAM, &x) < 0 && fs_batch_get_boundary_iterate(NULL, &x) < 0);
  EXPECT(fs_batch_restart(NULL, 0, &x, &x, &x, &x, NULL) < 0);
  EXPECT(fs_batch_level(NULL) == -1);
  EXPECT(fs_batch_get_state(NULL, &x, &x) < 0 && fs_batch_get_guess(NULL, &x, &x) < 0);
  EXPECT(fs_batch_get_hydrographs(NULL, 0, 1, &x) < 0 && fs_batch_get_iterations(NULL, 0, 1, &k) < 0 && fs_batch_get_status(NULL, &k) < 0);
  EXPECT(fs_batch_get_history(NULL, 0, 1, &x, &x) < 0 && fs_batch_get_residual_trace(NULL, 0, 1, &x) < 0);
  EXPECT(fs_batch_get_storage_stage(NULL, &x) < 0 && fs_batch_get_storage_stages(NULL, 0, 1, &x) < 0);
  EXPECT(fs_batch_derive(NULL, 0, 1, &x, NULL, NULL, NULL, NULL, NULL, NULL, NULL) < 0 && fs_batch_derive_device(NULL, 0, 1, 1) < 0);
  EXPECT(fs_batch_derived_device_ptr(NULL, 0) == NULL && fs_batch_hydrograph_device_ptr(NULL) == NULL && fs_batch_stream(NULL) == NULL);
  EXPECT(fs_batch_last_step_ms(NULL) < 0 && fs_batch_last_launch_count(NULL) == 0 && fs_batch_kernel_index(NULL) == -1);
  EXPECT(fs_batch_kernel_info(NULL, &k, &k, &k, &k) < 0);
  fs_batch_destroy(NULL);
  printf("ok: %d dispatch-table entries, host paths answered\n", n);
  return 0;
}
