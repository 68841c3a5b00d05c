// Kernel instantiation lists (X-macros): one line per instantiation, X(R, DT, SEC, M, W, FULL, BCK).
//   M = rows of the scalar system per lane (>= 2), W = waves per reach; FULL == 1: no per-row selects, valid only for N = 64*W*M
//   BCK: boundary-kind class the kernel is compiled for (fs_kernel.hpp): -1 any, 0 any but FS_BC_STORAGE_CURVE,
//        1 RECT_UNIFORM with bc_is_light() kinds on both ends, 2 + k flow hydrograph upstream and kind k downstream
// fs_abi.hip builds its dispatch table from them; the fs_part_*.hip translation units instantiate them (compiled in
// parallel by the Makefile: one translation unit with all ~130 kernels takes 2.5 minutes).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include "fs_kernel.hpp"

#define FS_BCK(kind) (2 + (kind))

typedef void (*FsLaunchFn)(const void *args, int B, hipStream_t st);

template <typename R, int SEC, int M, int W, bool RAGGED, int BCK, bool DIAG = true, int TAIL = -1>
void fs_launch(const void *args, int B, hipStream_t st) {
  const fs::KernelArgs<R> &a = *static_cast<const fs::KernelArgs<R> *>(args);
  hipLaunchKernelGGL((fs::preissmann_step_kernel<R, SEC, M, W, RAGGED, BCK, DIAG, TAIL>), dim3(B), dim3(64 * W), 0, st, a);
}

template <typename R, int SEC, int M, int W, int BCK>
void fs_launch_long(const void *args, int B, hipStream_t st) {
  const fs::KernelArgs<R> &a = *static_cast<const fs::KernelArgs<R> *>(args);
  hipLaunchKernelGGL((fs::preissmann_long_kernel<R, SEC, M, W, BCK>), dim3(B), dim3(64 * W), 0, st, a);
}

// reaches longer than one lane grid as a team of workgroups (fs_kernel.hpp, TEAM): X(R, DT, SEC, M, W, BCK); 64 W M rows per member, up to 64 / W
// members; general form (ragged, diagnostics compiled in), uniform section modes
// (measured, profiles/round4/team_kernel.txt: 16 rows per lane - the fewest members - wins at every length; an (8, 4) shape with two
// workgroups per CU, one computing while the other waits for its team, ties at 8 192 nodes and loses beyond: twice the members to wait for.
// A TABLE (8, 4) team - 2 048 rows per member, 512 registers + 992 B of scratch - gains 3 - 7 % on cases/gerd_roseires at 25 m and 10 m
// (1.91e5 against 1.84e5, 8.2e4 against 7.7e4): a compound-section reach is bound by its section evaluations, not by the passes' traffic;
// not kept)
// X(R, DT, SEC, M, W, FULL, BCK, DIAG); FULL: N a whole number of lane grids (every row a cell but the very last one).  The DIAG = 0 ones are the
// benchmark shapes of bench.py --workload long (flow hydrograph in, normal depth out, no history), as the flagship has them
#define FS_LIST_TEAM(X) \
  X(double, FS_F64, FS_SEC_RECT_UNIFORM, 16, 4, 0, 1, 1) \
  X(double, FS_F64, FS_SEC_RECT_UNIFORM, 16, 4, 0, 0, 1) \
  X(double, FS_F64, FS_SEC_TRAP_UNIFORM, 16, 4, 0, 0, 1) \
  X(double, FS_F64, FS_SEC_RECT_UNIFORM, 16, 4, 0, FS_BCK(FS_BC_NORMAL_DEPTH), 0) \
  X(double, FS_F64, FS_SEC_RECT_UNIFORM, 16, 4, 1, FS_BCK(FS_BC_NORMAL_DEPTH), 0)
template <typename R, int SEC, int M, int W, bool RAGGED, int BCK, bool DIAG>
void fs_launch_team(const void *args, int B, hipStream_t st) {
  const fs::KernelArgs<R> &a = *static_cast<const fs::KernelArgs<R> *>(args);
  // FS_TEAM_TEST_DROP=1 (tests only): one workgroup too few, so that the last reach's team waits for a member that never comes - the bounded
  // wait of the exchange must then end that reach with FS_TEAM_STALL and leave the others alone (tests/test_gpu_ragged_batches.py)
  const int grid = B * a.team_size - (std::getenv("FS_TEAM_TEST_DROP") ? 1 : 0);
  hipLaunchKernelGGL((fs::preissmann_step_kernel<R, SEC, M, W, RAGGED, BCK, DIAG, -1, true>), dim3(grid), dim3(64 * W), 0, st, a);
}
#define FS_INSTANTIATE_TEAM(R, DT, SEC, M, W, FULL, BCK, DIAG)                                                                          \
  template __global__ void fs::preissmann_step_kernel<R, SEC, M, W, !(FULL), (int)(BCK), (DIAG) != 0, -1, true>(const fs::KernelArgs<R>); \
  template void fs_launch_team<R, SEC, M, W, !(FULL), (int)(BCK), (DIAG) != 0>(const void *, int, hipStream_t);
#define FS_DECLARE_TEAM(R, DT, SEC, M, W, FULL, BCK, DIAG)                                                                                     \
  extern template __global__ void fs::preissmann_step_kernel<R, SEC, M, W, !(FULL), (int)(BCK), (DIAG) != 0, -1, true>(const fs::KernelArgs<R>); \
  extern template void fs_launch_team<R, SEC, M, W, !(FULL), (int)(BCK), (DIAG) != 0>(const void *, int, hipStream_t);

// reaches longer than one lane grid (fs_long.hpp): X(R, DT, SEC, M, W, BCK); capacity 64 M rows per wave slot x 64 slots
#define FS_LIST_LONG(X) \
  X(double, FS_F64, FS_SEC_RECT_UNIFORM, 8, 4, 0) \
  X(double, FS_F64, FS_SEC_TRAP_UNIFORM, 8, 4, 0) \
  X(double, FS_F64, FS_SEC_TABLE, 4, 4, -1) \
  X(double, FS_F64, FS_SEC_IRREGULAR, 4, 4, -1) \
  X(float, FS_F32, FS_SEC_RECT_UNIFORM, 8, 4, 0) \
  X(float, FS_F32, FS_SEC_TRAP_UNIFORM, 8, 4, 0) \
  X(float, FS_F32, FS_SEC_TABLE, 4, 4, -1)
#define FS_INSTANTIATE_LONG(R, DT, SEC, M, W, BCK)                                                       \
  template __global__ void fs::preissmann_long_kernel<R, SEC, M, W, (int)(BCK)>(const fs::KernelArgs<R>); \
  template void fs_launch_long<R, SEC, M, W, (int)(BCK)>(const void *, int, hipStream_t);
#define FS_DECLARE_LONG(R, DT, SEC, M, W, BCK)                                                                  \
  extern template __global__ void fs::preissmann_long_kernel<R, SEC, M, W, (int)(BCK)>(const fs::KernelArgs<R>); \
  extern template void fs_launch_long<R, SEC, M, W, (int)(BCK)>(const void *, int, hipStream_t);

#define FS_LIST_RECT(X, R, DT) \
  X(R, DT, FS_SEC_RECT_UNIFORM, 2, 1, 0, 0) \
  X(R, DT, FS_SEC_RECT_UNIFORM, 4, 1, 0, 0) \
  X(R, DT, FS_SEC_RECT_UNIFORM, 8, 1, 0, 0) \
  X(R, DT, FS_SEC_RECT_UNIFORM, 16, 1, 0, 0) \
  X(R, DT, FS_SEC_RECT_UNIFORM, 16, 2, 0, 0) \
  X(R, DT, FS_SEC_RECT_UNIFORM, 16, 4, 0, 0) \
  X(R, DT, FS_SEC_RECT_UNIFORM, 8, 1, 0, 1) \
  X(R, DT, FS_SEC_RECT_UNIFORM, 2, 1, 0, 1) \
  X(R, DT, FS_SEC_RECT_UNIFORM, 4, 1, 0, 1) \
  X(R, DT, FS_SEC_RECT_UNIFORM, 16, 4, 0, 1) \
  X(R, DT, FS_SEC_RECT_UNIFORM, 8, 1, 1, 1) \
  X(R, DT, FS_SEC_RECT_UNIFORM, 16, 4, 1, 1) \
  X(R, DT, FS_SEC_RECT_UNIFORM, 8, 4, 1, 1) \
  X(R, DT, FS_SEC_RECT_UNIFORM, 8, 4, 0, 1) \
  X(R, DT, FS_SEC_RECT_UNIFORM, 8, 8, 1, 1) \
  X(R, DT, FS_SEC_RECT_UNIFORM, 16, 1, 1, 1) \
  X(R, DT, FS_SEC_RECT_UNIFORM, 16, 2, 1, 1) \
  X(R, DT, FS_SEC_RECT_UNIFORM, 16, 1, 0, 1) \
  X(R, DT, FS_SEC_RECT_UNIFORM, 16, 2, 0, 1) \
  X(R, DT, FS_SEC_RECT_UNIFORM, 16, 4, 1, FS_BCK(FS_BC_NORMAL_DEPTH)) \
  X(R, DT, FS_SEC_RECT_UNIFORM, 16, 4, 0, FS_BCK(FS_BC_NORMAL_DEPTH)) \
  X(R, DT, FS_SEC_RECT_UNIFORM, 16, 2, 1, FS_BCK(FS_BC_NORMAL_DEPTH)) \
  X(R, DT, FS_SEC_RECT_UNIFORM, 16, 1, 1, FS_BCK(FS_BC_NORMAL_DEPTH)) \
  X(R, DT, FS_SEC_RECT_UNIFORM, 8, 1, 1, FS_BCK(FS_BC_NORMAL_DEPTH)) \
  X(R, DT, FS_SEC_RECT_UNIFORM, 8, 1, 0, FS_BCK(FS_BC_NORMAL_DEPTH))

#define FS_LIST_TRAP(X, R, DT) \
  X(R, DT, FS_SEC_TRAP_UNIFORM, 2, 1, 0, 0) \
  X(R, DT, FS_SEC_TRAP_UNIFORM, 4, 1, 0, 0) \
  X(R, DT, FS_SEC_TRAP_UNIFORM, 8, 1, 0, 0) \
  X(R, DT, FS_SEC_TRAP_UNIFORM, 8, 1, 1, 0) \
  X(R, DT, FS_SEC_TRAP_UNIFORM, 16, 4, 0, 0) \
  X(R, DT, FS_SEC_TRAP_UNIFORM, 8, 1, 1, FS_BCK(FS_BC_RATING_POWER)) \
  X(R, DT, FS_SEC_TRAP_UNIFORM, 8, 1, 0, FS_BCK(FS_BC_RATING_POWER))

#define FS_LIST_TABLE(X, R, DT) \
  X(R, DT, FS_SEC_TABLE, 2, 1, 0, 0) \
  X(R, DT, FS_SEC_TABLE, 4, 1, 0, 0) \
  X(R, DT, FS_SEC_TABLE, 8, 1, 0, 0) \
  X(R, DT, FS_SEC_TABLE, 8, 2, 0, 0) \
  X(R, DT, FS_SEC_TABLE, 8, 4, 0, 0) \
  X(R, DT, FS_SEC_TABLE, 2, 1, 0, -1) \
  X(R, DT, FS_SEC_TABLE, 4, 1, 0, -1) \
  X(R, DT, FS_SEC_TABLE, 8, 1, 0, -1) \
  X(R, DT, FS_SEC_TABLE, 8, 2, 0, -1) \
  X(R, DT, FS_SEC_TABLE, 8, 4, 0, -1)

// polyline sections: fp64 only, a few shapes (the section evaluation dominates, not the elimination)
#define FS_LIST_IRREGULAR(X) \
  X(double, FS_F64, FS_SEC_IRREGULAR, 2, 1, 0, 0) \
  X(double, FS_F64, FS_SEC_IRREGULAR, 8, 1, 0, 0) \
  X(double, FS_F64, FS_SEC_IRREGULAR, 8, 4, 0, 0) \
  X(double, FS_F64, FS_SEC_IRREGULAR, 2, 1, 0, -1) \
  X(double, FS_F64, FS_SEC_IRREGULAR, 8, 1, 0, -1) \
  X(double, FS_F64, FS_SEC_IRREGULAR, 8, 4, 0, -1)

// the hot shapes once more without the history / residual-trace stores (DIAG = false), for batches created without those flags
#define FS_LIST_NODIAG(X) \
  X(double, FS_F64, FS_SEC_RECT_UNIFORM, 16, 4, 1, FS_BCK(FS_BC_NORMAL_DEPTH)) \
  X(double, FS_F64, FS_SEC_RECT_UNIFORM, 16, 4, 0, FS_BCK(FS_BC_NORMAL_DEPTH)) \
  X(double, FS_F64, FS_SEC_RECT_UNIFORM, 16, 2, 1, FS_BCK(FS_BC_NORMAL_DEPTH)) \
  X(double, FS_F64, FS_SEC_RECT_UNIFORM, 16, 1, 1, FS_BCK(FS_BC_NORMAL_DEPTH)) \
  X(double, FS_F64, FS_SEC_RECT_UNIFORM, 8, 1, 1, FS_BCK(FS_BC_NORMAL_DEPTH)) \
  X(double, FS_F64, FS_SEC_TRAP_UNIFORM, 8, 1, 1, FS_BCK(FS_BC_RATING_POWER)) \
  X(float, FS_F32, FS_SEC_TRAP_UNIFORM, 8, 1, 1, FS_BCK(FS_BC_RATING_POWER)) \
  X(double, FS_F64, FS_SEC_TABLE, 2, 1, 0, 0) \
  X(double, FS_F64, FS_SEC_TABLE, 2, 1, 0, FS_BCK(FS_BC_RATING_BLEND)) \
  X(double, FS_F64, FS_SEC_IRREGULAR, 2, 1, 0, 0) \
  X(double, FS_F64, FS_SEC_IRREGULAR, 2, 1, 0, FS_BCK(FS_BC_NORMAL_DEPTH))

// The ensemble shape of BASELINE configs[3] (cases/gerd_roseires: 121 nodes in a 128-row lane grid, gate curve downstream) in its
// tail-only form (fs_kernel.hpp, TAIL): X(R, DT, SEC, M, W, BCK, TAIL), TAIL = (N - 1) mod M = the local row of the boundary row;
// ragged, no diagnostics; a batch with per-reach node counts takes the general ragged kernel instead
#define FS_LIST_TAIL(X) \
  X(double, FS_F64, FS_SEC_TABLE, 2, 1, FS_BCK(FS_BC_RATING_BLEND), 0) \
  X(double, FS_F64, FS_SEC_TABLE, 2, 1, FS_BCK(FS_BC_RATING_BLEND), 1)
#define FS_INSTANTIATE_TAIL(R, DT, SEC, M, W, BCK, TAIL)                                                                    \
  template __global__ void fs::preissmann_step_kernel<R, SEC, M, W, true, (int)(BCK), false, TAIL>(const fs::KernelArgs<R>); \
  template void fs_launch<R, SEC, M, W, true, (int)(BCK), false, TAIL>(const void *, int, hipStream_t);
#define FS_DECLARE_TAIL(R, DT, SEC, M, W, BCK, TAIL)                                                                               \
  extern template __global__ void fs::preissmann_step_kernel<R, SEC, M, W, true, (int)(BCK), false, TAIL>(const fs::KernelArgs<R>); \
  extern template void fs_launch<R, SEC, M, W, true, (int)(BCK), false, TAIL>(const void *, int, hipStream_t);

// explicit instantiation (fs_part_*.hip) / extern declaration (fs_abi.hip) of one entry
#define FS_INSTANTIATE(R, DT, SEC, M, W, FULL, BCK)                                                            \
  template __global__ void fs::preissmann_step_kernel<R, SEC, M, W, !(FULL), (int)(BCK)>(const fs::KernelArgs<R>); \
  template void fs_launch<R, SEC, M, W, !(FULL), (int)(BCK)>(const void *, int, hipStream_t);
#define FS_DECLARE(R, DT, SEC, M, W, FULL, BCK)                                                                       \
  extern template __global__ void fs::preissmann_step_kernel<R, SEC, M, W, !(FULL), (int)(BCK)>(const fs::KernelArgs<R>); \
  extern template void fs_launch<R, SEC, M, W, !(FULL), (int)(BCK)>(const void *, int, hipStream_t);
#define FS_INSTANTIATE_NODIAG(R, DT, SEC, M, W, FULL, BCK)                                                            \
  template __global__ void fs::preissmann_step_kernel<R, SEC, M, W, !(FULL), (int)(BCK), false>(const fs::KernelArgs<R>); \
  template void fs_launch<R, SEC, M, W, !(FULL), (int)(BCK), false>(const void *, int, hipStream_t);
#define FS_DECLARE_NODIAG(R, DT, SEC, M, W, FULL, BCK)                                                                       \
  extern template __global__ void fs::preissmann_step_kernel<R, SEC, M, W, !(FULL), (int)(BCK), false>(const fs::KernelArgs<R>); \
  extern template void fs_launch<R, SEC, M, W, !(FULL), (int)(BCK), false>(const void *, int, hipStream_t);
