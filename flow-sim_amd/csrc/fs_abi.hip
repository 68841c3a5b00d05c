// fs_abi.hip - host side of libflowsim_hip.so: the C ABI declared in include/flowsim_abi.h.
//
// Owns the device buffers of a batch (SoA, reach-major state [B][N]; per-level tables [level][B]),
// converts caller float64 host arrays to the batch dtype on upload, picks the kernel instantiation
// (cells per lane M, waves per reach W) from N, and launches the fused step kernel on the
// handle's HIP stream.  No CPU compute path exists here: without a HIP device every entry point
// that needs one fails and says so.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "fs_entries.hpp"

// ROCTx ranges around the phases a system trace should show (rocprofv3 --marker-trace): uploads, downloads, the step launch,
// the post-processing launch.  Without a tool attached a push / pop is a few nanoseconds.
#include <rocprofiler-sdk-roctx/roctx.h>
namespace {
struct TraceRange {
  explicit TraceRange(const char *name) { roctxRangePushA(name); }
  ~TraceRange() { roctxRangePop(); }
};
}  // namespace

#ifndef FS_MINIMAL
// the full library: the kernels are instantiated in the fs_part_*.hip translation units
FS_LIST_RECT(FS_DECLARE, double, FS_F64) FS_LIST_RECT(FS_DECLARE, float, FS_F32)
FS_LIST_TRAP(FS_DECLARE, double, FS_F64) FS_LIST_TRAP(FS_DECLARE, float, FS_F32)
FS_LIST_TABLE(FS_DECLARE, double, FS_F64) FS_LIST_TABLE(FS_DECLARE, float, FS_F32)
FS_LIST_IRREGULAR(FS_DECLARE)
FS_LIST_NODIAG(FS_DECLARE_NODIAG)
FS_LIST_TAIL(FS_DECLARE_TAIL)
FS_LIST_LONG(FS_DECLARE_LONG)
FS_LIST_TEAM(FS_DECLARE_TEAM)
#endif

namespace {

thread_local std::string g_err;

int fail(const std::string &m) { g_err = m; return -1; }

#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_));   \
  } while (0)

struct Shape { int M, W; };

// Every entry point that allocates, copies, launches or frees runs with the batch's device current and puts the
// caller's device back on the way out (two batches on different ordinals in one process, calls from another thread).
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define FS_ON_DEVICE(b)                                                                     \
  DeviceGuard guard_((b)->d.device);                                                        \
  if (!guard_.ok) return fail("hipSetDevice(" + std::to_string((b)->d.device) + ") failed")

typedef FsLaunchFn LaunchFn;
typedef const void *KernelPtr;

// full   == 1: no per-row selects, valid only for N = 64*W*M
// bck: boundary-kind class the kernel is compiled for (fs_kernel.hpp): -1 any, 0 any but FS_BC_STORAGE_CURVE,
//      1 RECT_UNIFORM with bc_is_light() kinds on both ends, 2 + k flow hydrograph upstream and kind k downstream
struct Entry { int dtype, sec, M, W, full, bck, diag; LaunchFn fn; KernelPtr kp; int longk; int tail = -1; int team = 0; };      // team: a reach as a team of workgroups (fs_kernel.hpp)   // diag == 0: no history / trace stores; longk: fs_long.hpp; tail >= 0: tail-only form, (N - 1) mod M == tail
#define FS_TABLE_ROW(R, DT, SEC, M, W, FULL, BCK)                                             \
  { DT, SEC, M, W, FULL, (int)(BCK), 1, &fs_launch<R, SEC, M, W, !(FULL), (int)(BCK)>,          \
    (KernelPtr)&fs::preissmann_step_kernel<R, SEC, M, W, !(FULL), (int)(BCK)>, 0 },
#define FS_TABLE_ROW_LONG(R, DT, SEC, M, W, BCK)                                                \
  { DT, SEC, M, W, 0, (int)(BCK), 1, &fs_launch_long<R, SEC, M, W, (int)(BCK)>,                  \
    (KernelPtr)&fs::preissmann_long_kernel<R, SEC, M, W, (int)(BCK)>, 1 },
#define FS_TABLE_ROW_NODIAG(R, DT, SEC, M, W, FULL, BCK)                                       \
  { DT, SEC, M, W, FULL, (int)(BCK), 0, &fs_launch<R, SEC, M, W, !(FULL), (int)(BCK), false>,    \
    (KernelPtr)&fs::preissmann_step_kernel<R, SEC, M, W, !(FULL), (int)(BCK), false>, 0 },
#define FS_TABLE_ROW_TAIL(R, DT, SEC, M, W, BCK, TAIL)                                            \
  { DT, SEC, M, W, 0, (int)(BCK), 0, &fs_launch<R, SEC, M, W, true, (int)(BCK), false, TAIL>,      \
    (KernelPtr)&fs::preissmann_step_kernel<R, SEC, M, W, true, (int)(BCK), false, TAIL>, 0, TAIL },
#define FS_TABLE_ROW_TEAM(R, DT, SEC, M, W, FULL, BCK, DIAG)                                               \
  { DT, SEC, M, W, FULL, (int)(BCK), DIAG, &fs_launch_team<R, SEC, M, W, !(FULL), (int)(BCK), (DIAG) != 0>,  \
    (KernelPtr)&fs::preissmann_step_kernel<R, SEC, M, W, !(FULL), (int)(BCK), (DIAG) != 0, -1, true>, 0, -1, 1 },
#define FS_ENTRY_X(R, DT, SEC, M, W, FULL, BCK) FS_TABLE_ROW(R, DT, SEC, M, W, FULL, BCK)
#define FS_ENTRY(R, DT, SEC, M, W) FS_TABLE_ROW(R, DT, SEC, M, W, 0, 0)

#if defined(FS_MINIMAL) && FS_MINIMAL == 2   // experiment builds: shapes for 512-node trapezoid reaches
const Entry kEntries[] = {FS_ENTRY_X(float, FS_F32, FS_SEC_TRAP_UNIFORM, 8, 1, 1, false)
                          FS_ENTRY_X(float, FS_F32, FS_SEC_TRAP_UNIFORM, 4, 2, 1, false)
                          FS_ENTRY_X(float, FS_F32, FS_SEC_TRAP_UNIFORM, 2, 4, 1, false)
                          FS_ENTRY_X(double, FS_F64, FS_SEC_TRAP_UNIFORM, 8, 1, 1, false)
                          FS_ENTRY_X(double, FS_F64, FS_SEC_TRAP_UNIFORM, 4, 2, 1, false)
                          FS_ENTRY_X(double, FS_F64, FS_SEC_TRAP_UNIFORM, 2, 4, 1, false)};
#elif defined(FS_MINIMAL) && FS_MINIMAL == 3   // experiment builds: the polyline kernels
const Entry kEntries[] = {FS_LIST_IRREGULAR(FS_TABLE_ROW)};
#elif defined(FS_MINIMAL)   // experiment builds: just the flagship shapes
const Entry kEntries[] = {FS_TABLE_ROW_NODIAG(double, FS_F64, FS_SEC_RECT_UNIFORM, 16, 4, 1, FS_BCK(FS_BC_NORMAL_DEPTH))
                          FS_TABLE_ROW_NODIAG(double, FS_F64, FS_SEC_RECT_UNIFORM, 8, 8, 1, FS_BCK(FS_BC_NORMAL_DEPTH))
                          FS_TABLE_ROW_NODIAG(double, FS_F64, FS_SEC_TABLE, 2, 1, 0, FS_BCK(FS_BC_RATING_BLEND))
#ifndef FS_NO_TAIL
                          FS_LIST_TAIL(FS_TABLE_ROW_TAIL)
#endif
                          FS_TABLE_ROW_NODIAG(double, FS_F64, FS_SEC_IRREGULAR, 2, 1, 0, FS_BCK(FS_BC_NORMAL_DEPTH))
                          FS_ENTRY_X(double, FS_F64, FS_SEC_RECT_UNIFORM, 16, 4, 1, FS_BCK(FS_BC_NORMAL_DEPTH))
                          FS_ENTRY_X(double, FS_F64, FS_SEC_RECT_UNIFORM, 16, 4, 1, true)
                          FS_ENTRY_X(double, FS_F64, FS_SEC_RECT_UNIFORM, 16, 4, 0, true)
                          FS_ENTRY_X(double, FS_F64, FS_SEC_RECT_UNIFORM, 8, 4, 1, true)
                          FS_ENTRY_X(double, FS_F64, FS_SEC_RECT_UNIFORM, 8, 8, 1, true)
                          FS_ENTRY_X(double, FS_F64, FS_SEC_RECT_UNIFORM, 4, 4, 1, true)
                          FS_ENTRY_X(double, FS_F64, FS_SEC_RECT_UNIFORM, 8, 1, 0, true)
                          FS_ENTRY_X(double, FS_F64, FS_SEC_RECT_UNIFORM, 4, 1, 0, true) FS_ENTRY_X(double, FS_F64, FS_SEC_RECT_UNIFORM, 2, 1, 0, true)
                          FS_ENTRY(double, FS_F64, FS_SEC_RECT_UNIFORM, 4, 1) FS_ENTRY(double, FS_F64, FS_SEC_RECT_UNIFORM, 2, 1)
                          FS_ENTRY(double, FS_F64, FS_SEC_TABLE, 2, 1) FS_ENTRY(double, FS_F64, FS_SEC_IRREGULAR, 2, 1)
                          FS_ENTRY_X(double, FS_F64, FS_SEC_TRAP_UNIFORM, 8, 1, 1, false)
                          FS_ENTRY_X(float, FS_F32, FS_SEC_TRAP_UNIFORM, 8, 1, 1, false)
                          FS_ENTRY_X(float, FS_F32, FS_SEC_TRAP_UNIFORM, 8, 1, 1, FS_BCK(FS_BC_RATING_POWER))
                          FS_TABLE_ROW_NODIAG(double, FS_F64, FS_SEC_TRAP_UNIFORM, 8, 1, 1, FS_BCK(FS_BC_RATING_POWER))
                          FS_ENTRY_X(float, FS_F32, FS_SEC_RECT_UNIFORM, 8, 1, 1, true) FS_ENTRY_X(float, FS_F32, FS_SEC_RECT_UNIFORM, 8, 1, 0, true)
                          FS_ENTRY(double, FS_F64, FS_SEC_TRAP_UNIFORM, 4, 1) FS_ENTRY(double, FS_F64, FS_SEC_TABLE, 4, 1)
                          FS_TABLE_ROW_LONG(double, FS_F64, FS_SEC_RECT_UNIFORM, 8, 4, 0)
                          FS_TABLE_ROW_TEAM(double, FS_F64, FS_SEC_RECT_UNIFORM, 16, 4, 0, 1, 1)
                          FS_TABLE_ROW_TEAM(double, FS_F64, FS_SEC_RECT_UNIFORM, 16, 4, 0, FS_BCK(FS_BC_NORMAL_DEPTH), 0)
                          FS_TABLE_ROW_TEAM(double, FS_F64, FS_SEC_RECT_UNIFORM, 16, 4, 1, FS_BCK(FS_BC_NORMAL_DEPTH), 0)
#ifdef FS_TEAM_8X4
                          FS_TABLE_ROW_TEAM(double, FS_F64, FS_SEC_RECT_UNIFORM, 8, 4, 0, 1, 1)
#endif
                          };
#else
const Entry kEntries[] = {FS_LIST_RECT(FS_TABLE_ROW, double, FS_F64) FS_LIST_TRAP(FS_TABLE_ROW, double, FS_F64)
                          FS_LIST_TABLE(FS_TABLE_ROW, double, FS_F64) FS_LIST_RECT(FS_TABLE_ROW, float, FS_F32)
                          FS_LIST_TRAP(FS_TABLE_ROW, float, FS_F32) FS_LIST_TABLE(FS_TABLE_ROW, float, FS_F32)
                          FS_LIST_IRREGULAR(FS_TABLE_ROW) FS_LIST_NODIAG(FS_TABLE_ROW_NODIAG) FS_LIST_LONG(FS_TABLE_ROW_LONG)
                          FS_LIST_TAIL(FS_TABLE_ROW_TAIL) FS_LIST_TEAM(FS_TABLE_ROW_TEAM)};
#endif

// usk / dsk: boundary kinds of the batch.  FS_KERNEL_SHAPE="M,W" and FS_KERNEL_GENERAL=1 (environment) narrow the
// choice for experiments and tests.
constexpr int kNumEntries = (int)(sizeof(kEntries) / sizeof(kEntries[0]));

// hetero: bit 0 = per-reach node counts (ragged kernels only), bit 1 = per-reach scheme or boundary kinds (kernels that read
// them: boundary classes 0 and -1)
bool entry_fits(const Entry &e, int dtype, int sec, int N, int usk, int dsk, bool need_diag, bool need_any, int hetero = 0) {
  if ((hetero & 1) && (e.full || e.tail >= 0)) return false;      // per-reach node counts: the boundary row's place differs from reach to reach
  if (e.tail >= 0 && (N - 1) % e.M != e.tail) return false;
  if ((hetero & 2) && e.bck > 0) return false;
  const bool light = fs::bc_is_light(usk) && fs::bc_is_light(dsk);
  const bool beyond0 = usk >= FS_BC_STORAGE_CURVE || dsk >= FS_BC_STORAGE_CURVE;      // general storage / host rows: class -1 only
  if (e.dtype != dtype || e.sec != sec) return false;
  // rows of the scalar system: N - 1 cells + the downstream boundary row; a long-reach kernel makes up to 64 / W passes
  const long cap = 64L * e.W * e.M * ((e.longk || e.team) ? 64 / e.W : 1);
  if (cap < N || N > 32768) return false;
  if (e.team && (N <= 4096 || N <= 64L * e.W * e.M || std::getenv("FS_NO_TEAM"))) return false;      // a team only where one workgroup does not hold the reach (FS_NO_TEAM=1: the multi-pass kernel instead)
  if (e.team) { if (const char *tm = std::getenv("FS_TEAM_M")) { if (std::atoi(tm) != e.M) return false; } }      // experiments: rows per lane of the team kernel
  if (e.longk && ((need_any && e.bck != -1) || (e.bck == 0 && beyond0))) return false;      // iteration budget / host rows: class -1 (tables, polylines)
  if (e.full && (e.team ? N % (64L * e.W * e.M) != 0 : N != cap)) return false;      // (a team's: a whole number of lane grids)
  if (!e.diag && need_diag) return false;
  if (need_any && e.bck != -1) return false;
  if (e.bck == 0 && beyond0) return false;
  if (e.bck == 1 && (!light || sec != FS_SEC_RECT_UNIFORM)) return false;
  if (e.bck >= 2 && (usk != FS_BC_FLOW_HYDROGRAPH || dsk != e.bck - 2)) return false;
  return true;
}

// need_any: the caller needs a kernel of boundary class -1 (iteration budget, host rows)
const Entry *pick_kernel(int dtype, int sec, int N, int usk, int dsk, bool need_diag, std::string *why, bool need_any = false,
                         bool honour_index = true, int hetero = 0) {
  if (const char *env = honour_index ? std::getenv("FS_KERNEL_INDEX") : nullptr) {      // tests: one specific instantiation or nothing
    const int i = std::atoi(env);
    if (i >= 0 && i < kNumEntries && entry_fits(kEntries[i], dtype, sec, N, usk, dsk, need_diag, need_any, hetero)) return &kEntries[i];
    if (why) *why = "FS_KERNEL_INDEX=" + std::string(env) + " does not fit this batch";
    return nullptr;
  }
  int wantM = 0, wantW = 0;
  if (const char *env = std::getenv("FS_KERNEL_SHAPE")) std::sscanf(env, "%d,%d", &wantM, &wantW);
  const char *gen = std::getenv("FS_KERNEL_GENERAL");
  const bool general_only = gen && gen[0] == '1';
  const Entry *best = nullptr;
  for (const Entry &e : kEntries) {
    if (!entry_fits(e, dtype, sec, N, usk, dsk, need_diag, need_any, hetero)) continue;
    if (general_only && (!e.diag || e.bck >= 2)) continue;
    if (wantM && (e.M != wantM || e.W != wantW)) continue;
    // smallest capacity first; on ties prefer fewer waves per reach, then the more specific variant
    auto rank = [](const Entry &x) { return x.full + (x.bck >= 2 ? 4 : x.bck == 1 ? 2 : x.bck == 0 ? 1 : 0) + (x.diag ? 0 : 8) + (x.tail >= 0 ? 16 : 0); };
    const int spec = rank(e), bspec = best ? rank(*best) : 0;
    // a kernel that keeps the reach on chip whenever one fits: one workgroup, else a team of them, else the multi-pass kernel
    const int tier = e.longk ? 2 : (e.team ? 1 : 0), btier = best ? (best->longk ? 2 : (best->team ? 1 : 0)) : 0;
    if (best && tier != btier) {
      if (tier < btier) best = &e;
      continue;
    }
    if (!best || e.M * e.W < best->M * best->W || (e.M * e.W == best->M * best->W && e.W < best->W) ||
        (e.M == best->M && e.W == best->W && spec > bspec))
      best = &e;
  }
  if (!best && why) *why = "no kernel instantiation for N=" + std::to_string(N) + " (supported: 2..32768 nodes for the uniform section "
                           "modes, 2..16384 for tables and polylines)";
  return best;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// Host <-> device staging.  Large transfers go through a ring of pinned chunks (hipHostMalloc): while the DMA engine moves
// chunk i, a few host threads copy (and, for fp32 batches, convert) chunk i + 1 between the caller's pageable buffer and
// the next pinned chunk - the host copy, the page faults of a freshly allocated destination and the PCIe transfer overlap
// instead of adding up.  (A pageable hipMemcpy device -> host measured 22.9 GB/s on this box, profiles/round2/pcie.json.)
// ---------------------------------------------------------------------------------------------
namespace {

class CopyPool {                       // a handful of persistent host threads: parallel_for over slices of a chunk
 public:
  static CopyPool &get() { static CopyPool p; return p; }
  int size() const { return (int)workers_.size() + 1; }
  void run(size_t n, const std::function<void(size_t, size_t)> &fn) {        // fn(begin, end) over [0, n)
    const int parts = size();
    if (n < (size_t)1 << 16 || parts == 1) { fn(0, n); return; }
    std::lock_guard<std::mutex> one_at_a_time(run_m_);        // handles on different host threads share the pool
    {
      std::lock_guard<std::mutex> lk(m_);
      fn_ = &fn; n_ = n; parts_ = parts; pending_ = parts - 1; ++epoch_;
    }
    cv_.notify_all();
    slice(parts - 1);
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [&] { return pending_ == 0; });
  }

 private:
  CopyPool() {
    const char *env = std::getenv("FS_COPY_THREADS");
    int want = env ? std::atoi(env) : (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency() / 2));
    want = std::max(1, std::min(want, 32));
    for (int i = 0; i + 1 < want; ++i) workers_.emplace_back([this, i] { loop(i); });
  }
  ~CopyPool() {
    { std::lock_guard<std::mutex> lk(m_); stop_ = true; ++epoch_; }
    cv_.notify_all();
    for (auto &t : workers_) t.join();
  }
  void slice(int part) {
    const size_t per = (n_ + parts_ - 1) / parts_, a = std::min(n_, per * part), b = std::min(n_, a + per);
    if (a < b) (*fn_)(a, b);
  }
  void loop(int id) {
    unsigned long seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> lk(m_);
      cv_.wait(lk, [&] { return epoch_ != seen; });
      seen = epoch_;
      if (stop_) return;
      lk.unlock();
      slice(id);
      lk.lock();
      if (--pending_ == 0) done_.notify_one();
    }
  }
  std::vector<std::thread> workers_;
  std::mutex m_, run_m_;
  std::condition_variable cv_, done_;
  const std::function<void(size_t, size_t)> *fn_ = nullptr;
  size_t n_ = 0;
  int parts_ = 1, pending_ = 0;
  unsigned long epoch_ = 0;
  bool stop_ = false;
};

constexpr size_t kStageChunk = (size_t)32 << 20;      // bytes per pinned chunk
constexpr int kStageSlots = 3;
constexpr size_t kStageMin = (size_t)8 << 20;         // smaller transfers: one plain copy

struct Staging {
  void *buf[kStageSlots] = {nullptr, nullptr, nullptr};
  hipEvent_t ev[kStageSlots] = {nullptr, nullptr, nullptr};
  bool ready = false;
  int init() {
    if (ready) return 0;
    for (int i = 0; i < kStageSlots; ++i) {
      if (hipHostMalloc(&buf[i], kStageChunk, hipHostMallocDefault) != hipSuccess) return -1;
      if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) return -1;
    }
    ready = true;
    return 0;
  }
  void release() {
    for (int i = 0; i < kStageSlots; ++i) {
      if (buf[i]) (void)hipHostFree(buf[i]);
      if (ev[i]) (void)hipEventDestroy(ev[i]);
      buf[i] = nullptr; ev[i] = nullptr;
    }
    ready = false;
  }
};

}  // namespace

struct fs_batch {
  fs_batch_desc d;
  size_t esz;                 // sizeof(real)
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  bool iterating = false;     // a level opened by fs_batch_iterate has not been closed yet
  int launches = 0;
  int level = 0;
  int restart_level = 0;      // > 0: the batch was re-seeded at this level (fs_batch_restart); history rows 1 .. restart_level-1 are empty
  const Entry *kern = nullptr;
  // scheme
  double theta = 0.6, dt = 0, dx = 0, tol = 1e-4;
  int max_iter = 100;
  bool have_scheme = false, have_geo = false, have_state = false, have_bc[2] = {false, false};
  // device buffers
  void *hk = nullptr, *Qk = nullptr, *hg = nullptr, *Qg = nullptr;
  void *geo_uniform = nullptr, *geo_table = nullptr, *n_override = nullptr;
  void *poly_x = nullptr, *poly_z = nullptr, *poly_lim = nullptr;
  int32_t *poly_n = nullptr;
  void *poly_tz = nullptr;              // stage tables of the polylines (fs_poly.hpp)
  int poly_K = 0;
  size_t geo_reach_stride = 0, poly_reach_stride = 0;     // per-reach geometry (elements between the tables of two reaches), 0: shared
  int32_t *reach_nodes = nullptr;       // [B] per-reach node counts (heterogeneous batch) or nullptr
  std::vector<int32_t> reach_nodes_host;      // its host copy (empty: every reach has n_nodes): where a reach's last node is
  bool any_storage[2] = {false, false}; // some reach's boundary on this side is a storage kind (per-reach kinds: OR over the reaches)
  void *reach_scheme = nullptr;         // [5][B] per-reach theta, dt, dx, tolerance, max_iter or nullptr
  std::vector<double> rs_host[5];       // what the caller set per reach (empty: the batch-wide value of fs_batch_set_scheme)
  int32_t *reach_kinds = nullptr;       // [2][B] per-reach boundary kinds or nullptr
  bool kinds_per_reach[2] = {false, false};
  bool some_host_rows[2] = {false, false};      // per-reach kinds with FS_BC_HOST_ROW among them: fs_batch_set_host_rows writes those reaches' rows only
  void *rows_stage = nullptr;          // [3][B] what the caller handed to fs_batch_set_host_rows, before the masked merge into the parameters
  void *bc_params[2] = {nullptr, nullptr}, *bc_target[2] = {nullptr, nullptr};
  int bc_kind[2] = {0, 0}, bc_stride[2] = {0, 0};
  void *Yprev = nullptr, *stage_hist = nullptr, *trace = nullptr, *hydro = nullptr, *hist_h = nullptr, *hist_Q = nullptr;
  int32_t *iters = nullptr, *status = nullptr;
  int32_t *it_done = nullptr;          // [B] Newton iterations spent on the open level (fs_batch_iterate)
  double *ends_dev = nullptr;          // [4][B] scratch of fs_batch_get_boundary_iterate (allocated on first use)
  double *ends_pin = nullptr;          // its pinned host mirror
  int32_t *open_dev = nullptr, *open_pin = nullptr;      // fs_batch_iterate: count of the reaches still iterating
  void *derived[8] = {nullptr};        // device results of the last derive call, kept and reused
  size_t derived_cap[8] = {0};         // their capacities in elements
  unsigned long long *dbg = nullptr;
  Staging stage;                       // pinned chunks for large host <-> device transfers
  void *kc_scratch = nullptr;          // long reaches: level constants [B][4][passes * 64 W M]
  size_t kc_scratch_elems = 0;
  int passes = 0;
  void *team_mail = nullptr;           // reaches stepped by teams of workgroups: mailboxes [B][2][G W + 1][kTeamWords]
  size_t team_mail_elems = 0;
  unsigned long long *team_sync = nullptr;      // [1 + B] ticket counter + per-reach post counters, zeroed before every launch
  int team_size = 0;
  uint32_t team_epoch = 0;             // team launches made on this handle (the high half of the tagged mailbox's tags)
};

namespace {

int upload(fs_batch *b, void **dst, const double *src, size_t n) {
  TraceRange range_("flowsim:upload");
  if (!*dst) HIP_TRY(hipMalloc(dst, n * b->esz));
  const bool f64 = b->d.dtype == FS_F64;
  if (f64 || n * b->esz < kStageMin || b->stage.init() != 0) {
    // fp64: the runtime's own pageable path (it pins and pipelines; ~55 GB/s here); small fp32 arrays: convert, then one copy
    if (f64) {
      HIP_TRY(hipMemcpyAsync(*dst, src, n * sizeof(double), hipMemcpyHostToDevice, b->stream));
      HIP_TRY(hipStreamSynchronize(b->stream));
    } else {
      std::vector<float> tmp(n);
      for (size_t i = 0; i < n; ++i) tmp[i] = (float)src[i];
      HIP_TRY(hipMemcpyAsync(*dst, tmp.data(), n * sizeof(float), hipMemcpyHostToDevice, b->stream));
      HIP_TRY(hipStreamSynchronize(b->stream));
    }
    return 0;
  }
  // fp32, large: convert chunk i + 1 into a pinned slot (host threads) while chunk i is on the bus
  const size_t per = kStageChunk / sizeof(float);
  int slot = 0;
  for (size_t off = 0; off < n; off += per, slot = (slot + 1) % kStageSlots) {
    const size_t cnt = std::min(per, n - off);
    HIP_TRY(hipEventSynchronize(b->stage.ev[slot]));              // the DMA that last read this slot is through
    float *pin = (float *)b->stage.buf[slot];
    const double *sp = src + off;
    CopyPool::get().run(cnt, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) pin[i] = (float)sp[i]; });
    HIP_TRY(hipMemcpyAsync((char *)*dst + off * sizeof(float), pin, cnt * sizeof(float), hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipEventRecord(b->stage.ev[slot], b->stream));
  }
  HIP_TRY(hipStreamSynchronize(b->stream));
  return 0;
}

int download(fs_batch *b, double *dst, const void *src, size_t off_elems, size_t n) {
  TraceRange range_("flowsim:download");
  HIP_TRY(hipStreamSynchronize(b->stream));
  const bool f64 = b->d.dtype == FS_F64;
  const char *dev = (const char *)src + off_elems * b->esz;
  if (n * b->esz < kStageMin || b->stage.init() != 0) {
    if (f64) {
      HIP_TRY(hipMemcpy(dst, dev, n * 8, hipMemcpyDeviceToHost));
    } else {
      std::vector<float> tmp(n);
      HIP_TRY(hipMemcpy(tmp.data(), dev, n * 4, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < n; ++i) dst[i] = tmp[i];
    }
    return 0;
  }
  // chunk i + 1 comes over the bus into a pinned slot while host threads copy (fp32: widen) chunk i into the caller's buffer
  const size_t per = kStageChunk / b->esz, chunks = (n + per - 1) / per;
  auto issue = [&](size_t c) -> hipError_t {
    const int slot = (int)(c % kStageSlots);
    const size_t off = c * per, cnt = std::min(per, n - off);
    hipError_t e = hipMemcpyAsync(b->stage.buf[slot], dev + off * b->esz, cnt * b->esz, hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipEventRecord(b->stage.ev[slot], b->stream);
    return e;
  };
  for (size_t c = 0; c < std::min<size_t>(chunks, kStageSlots - 1); ++c) HIP_TRY(issue(c));
  for (size_t c = 0; c < chunks; ++c) {
    const int slot = (int)(c % kStageSlots);
    const size_t off = c * per, cnt = std::min(per, n - off);
    if (c + kStageSlots - 1 < chunks) HIP_TRY(issue(c + kStageSlots - 1));   // its slot was drained in the iteration before
    HIP_TRY(hipEventSynchronize(b->stage.ev[slot]));
    double *dp = dst + off;
    if (f64) {
      const double *pin = (const double *)b->stage.buf[slot];
      CopyPool::get().run(cnt, [&](size_t lo, size_t hi) { std::memcpy(dp + lo, pin + lo, (hi - lo) * sizeof(double)); });
    } else {
      const float *pin = (const float *)b->stage.buf[slot];
      CopyPool::get().run(cnt, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) dp[i] = pin[i]; });
    }
  }
  return 0;
}

// The caller's TrapezoidalSection table [FS_GEO_NPARAM][N] plus the rows of what follows from it alone
// (fs_device.hpp FS_GEOX_*: side-slope roots, bankfull geometry, reciprocal / -1.5-power roughnesses), computed here
// once in double so that no node evaluation of any Newton iteration has to.
std::vector<double> extend_table(const double *t, size_t N) {
  std::vector<double> x((size_t)fs::FS_GEOX_NROWS * N, 0.0);
  std::memcpy(x.data(), t, (size_t)FS_GEO_NPARAM * N * sizeof(double));
  auto in = [&](int row, size_t i) { return t[(size_t)row * N + i]; };
  for (size_t i = 0; i < N; ++i) {
    const double b = in(FS_GEO_B_MAIN, i), m = in(FS_GEO_M_MAIN, i), hbf = in(FS_GEO_H_BANKFULL, i), mfp = in(FS_GEO_M_FP, i);
    const double sm = std::sqrt(1.0 + m * m), Tb = b + 2.0 * m * hbf;
    const double nm = in(FS_GEO_N_MAIN, i), nl = in(FS_GEO_N_LEFT, i), nr = in(FS_GEO_N_RIGHT, i);
    auto put = [&](int row, double v) { x[(size_t)row * N + i] = v; };
    put(fs::FS_GEOX_SM, sm); put(fs::FS_GEOX_SFP, std::sqrt(1.0 + mfp * mfp)); put(fs::FS_GEOX_TB, Tb);
    put(fs::FS_GEOX_AM, (b + Tb) / 2.0 * hbf); put(fs::FS_GEOX_PM, b + 2.0 * hbf * sm);
    put(fs::FS_GEOX_RNM, nm > 0 ? 1.0 / nm : 0.0); put(fs::FS_GEOX_KM15, nm > 0 ? std::pow(nm, -1.5) : 0.0);
    put(fs::FS_GEOX_KL15, nl > 0 ? std::pow(nl, -1.5) : 0.0); put(fs::FS_GEOX_KR15, nr > 0 ? std::pow(nr, -1.5) : 0.0);
  }
  return x;
}

// the Newton vector at the two ends of every reach, as doubles: out[4][B] = h_0, Q_0, h_last, Q_last (fs_batch_get_boundary_iterate)
template <typename R>
__global__ void gather_boundary_iterate(const R *hg, const R *Qg, const int32_t *reach_nodes, double *out, size_t B, size_t N) {
  const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= B) return;
  const size_t last = (reach_nodes ? (size_t)reach_nodes[r] : N) - 1;
  out[r] = (double)hg[r * N]; out[B + r] = (double)Qg[r * N];
  out[2 * B + r] = (double)hg[r * N + last]; out[3 * B + r] = (double)Qg[r * N + last];
}

// the downstream half of the level-0 hydrograph row from the state on the device: node reach_nodes[r] - 1 of every reach
// (fs_batch_set_reach_nodes after fs_batch_set_state: the row was filled from column N - 1)
template <typename R>
__global__ void refresh_level0_downstream(const R *hk, const R *Qk, const int32_t *reach_nodes, R *hydro, size_t B, size_t N) {
  const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= B) return;
  const size_t last = (reach_nodes ? (size_t)reach_nodes[r] : N) - 1;
  hydro[2 * B + r] = hk[r * N + last]; hydro[3 * B + r] = Qk[r * N + last];
}

// reaches whose open level still iterates (fs_batch_iterate): not yet accepted and not failed
__global__ void count_open_reaches(const int32_t *done, const int32_t *status, int32_t *out, size_t B) {
  const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool open = r < B && done[r] >= 0 && (status[r] == FS_OK || status[r] == FS_ILL_CONDITIONED);   // (a warning, not a failure)
  const unsigned long long m = __ballot(open);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(out, (int32_t)__popcll(m));
}

// fs_batch_set_host_rows on a side with per-reach kinds: the rows of the host-evaluated reaches go into parameter rows 0..2, the
// parameters of the reaches whose kind the device evaluates stay what fs_batch_set_bc_per_reach stored
template <typename R>
__global__ void merge_host_rows(R *params, const R *rows, const int32_t *kinds, size_t B) {
  const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= B || kinds[r] != FS_BC_HOST_ROW) return;
  for (int i = 0; i < 3; ++i) params[(size_t)i * B + r] = rows[(size_t)i * B + r];
}

// level 0 of every array from one (h, Q) pair per reach
template <typename R>
__global__ void broadcast_state(const R *h, const R *Q, R *hk, R *Qk, R *hg, R *Qg, R *hist_h, R *hist_Q, R *hydro,
                                size_t B, size_t N) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * N) return;
  const size_t r = i / N, node = i - r * N;
  const R hv = h[r], Qv = Q[r];
  hk[i] = hv; Qk[i] = Qv; hg[i] = hv; Qg[i] = Qv;
  if (hist_h) { hist_h[i] = hv; hist_Q[i] = Qv; }
  if (node == 0) { hydro[0 * B + r] = hv; hydro[1 * B + r] = Qv; hydro[2 * B + r] = hv; hydro[3 * B + r] = Qv; }
}

template <typename R> void fill_args(const fs_batch *b, int n_steps, fs::KernelArgs<R> &a) {
  a.B = b->d.n_reaches; a.N = b->d.n_nodes; a.n_steps = n_steps; a.level0 = b->level; a.max_iter = b->max_iter;
  a.theta = (R)b->theta; a.dt = (R)b->dt; a.dx = (R)b->dx; a.tol = (R)b->tol;
  a.hk = (R *)b->hk; a.Qk = (R *)b->Qk; a.hg = (R *)b->hg; a.Qg = (R *)b->Qg;
  a.geo_uniform = (const R *)b->geo_uniform; a.geo_table = (const R *)b->geo_table;
  a.n_override = (const R *)b->n_override;
  a.poly_x = (const R *)b->poly_x; a.poly_z = (const R *)b->poly_z; a.poly_lim = (const R *)b->poly_lim; a.poly_n = b->poly_n;
  a.poly_tz = (const R *)b->poly_tz; a.poly_K = b->poly_K;
  a.geo_reach_stride = (int64_t)b->geo_reach_stride; a.poly_reach_stride = (int64_t)b->poly_reach_stride;
  a.reach_nodes = b->reach_nodes; a.reach_scheme = (const R *)b->reach_scheme;
  a.reach_kinds = (b->kinds_per_reach[0] || b->kinds_per_reach[1]) ? b->reach_kinds : nullptr;
  fs::BCDesc<R> *bc[2] = {&a.us, &a.ds};
  for (int s = 0; s < 2; ++s) {
    bc[s]->kind = b->bc_kind[s]; bc[s]->stride = b->bc_stride[s];
    bc[s]->params = (const R *)b->bc_params[s]; bc[s]->target = (const R *)b->bc_target[s]; bc[s]->tgt = R(0);
  }
  a.Yprev = (R *)b->Yprev; a.stage_hist = (R *)b->stage_hist; a.trace = (R *)b->trace; a.hydro = (R *)b->hydro; a.iters = b->iters; a.status = b->status;
  a.hist_h = (R *)b->hist_h; a.hist_Q = (R *)b->hist_Q;
  a.dbg = b->dbg;
  a.kc_scratch = (R *)b->kc_scratch; a.passes = b->passes;
  a.team_size = b->team_size; a.team_mail = (R *)b->team_mail; a.team_sync = b->team_sync; a.team_epoch = b->team_epoch;
  a.iter_budget = 0; a.it_done = b->it_done;
}

// picks the instantiation for the batch as it is now (the boundary kinds are known) and launches it on the handle's stream
int launch_steps(fs_batch *b, int n_steps, int iter_budget) {
  TraceRange range_(iter_budget > 0 ? "flowsim:iterate" : "flowsim:step");
  std::string why;
  const int hetero = (b->reach_nodes ? 1 : 0) | ((b->reach_scheme || b->kinds_per_reach[0] || b->kinds_per_reach[1]) ? 2 : 0);
  const Entry *k = pick_kernel(b->d.dtype, b->d.section_mode, b->d.n_nodes, b->bc_kind[0], b->bc_kind[1],
                               (b->d.flags & (FS_FLAG_HISTORY | FS_FLAG_TRACE | FS_FLAG_MONITOR)) != 0, &why, iter_budget > 0, true, hetero);
  if (!k && why.rfind("FS_KERNEL_INDEX", 0) == 0) return fail("fs_batch_step: " + why);
  if (!k && (b->bc_kind[0] == FS_BC_STORAGE_CURVE || b->bc_kind[1] == FS_BC_STORAGE_CURVE))
    return fail("fs_batch_step: FS_BC_STORAGE_CURVE needs section mode FS_SEC_TABLE or FS_SEC_IRREGULAR");
  if (!k) return fail("fs_batch_step: no kernel instantiation for this boundary kind at this size");
  b->kern = k;
  b->passes = 0;
  if (k->longk) {      // a reach longer than one lane grid: passes of 64 W M rows, level constants in a scratch of the batch's own
    const size_t chunk = (size_t)64 * k->W * k->M;
    b->passes = (int)((b->d.n_nodes + chunk - 1) / chunk);
    // (uniform sections recompute their level constants, fs_long.hpp: no scratch)
    const bool recompute = FS_LONG_RECOMPUTE && (b->d.section_mode == FS_SEC_RECT_UNIFORM || b->d.section_mode == FS_SEC_TRAP_UNIFORM);
    const size_t need = recompute ? 0 : (size_t)b->d.n_reaches * 4 * b->passes * chunk;
    if (b->kc_scratch_elems < need) {
      if (b->kc_scratch) { (void)hipFree(b->kc_scratch); b->kc_scratch = nullptr; b->kc_scratch_elems = 0; }
      HIP_TRY(hipMalloc(&b->kc_scratch, need * b->esz));
      b->kc_scratch_elems = need;
    }
  }
  b->team_size = 0;
  if (k->team) {       // G workgroups per reach: their mailboxes and counters (the counters start every launch at zero)
    const size_t chunk = (size_t)64 * k->W * k->M, B = b->d.n_reaches;
    b->team_size = (int)((b->d.n_nodes + chunk - 1) / chunk);
    // (16 bytes per word: the tagged form posts (value, tag) pairs; zeroed once - a tag is never 0, launches count from 1)
    const size_t need = B * 2 * ((size_t)b->team_size * k->W + 1) * fs::kTeamWords * 2;
    if (b->team_mail_elems < need) {
      if (b->team_mail) { (void)hipFree(b->team_mail); b->team_mail = nullptr; b->team_mail_elems = 0; }
      HIP_TRY(hipMalloc(&b->team_mail, need * sizeof(double)));
      HIP_TRY(hipMemsetAsync(b->team_mail, 0, need * sizeof(double), b->stream));
      b->team_mail_elems = need;
    }
    ++b->team_epoch;
    if (!b->team_sync) HIP_TRY(hipMalloc((void **)&b->team_sync, (1 + B) * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(b->team_sync, 0, (1 + B) * sizeof(unsigned long long), b->stream));
  }
  HIP_TRY(hipEventRecord(b->ev0, b->stream));
  if (b->d.dtype == FS_F64) {
    fs::KernelArgs<double> a; fill_args(b, n_steps, a);
    a.iter_budget = iter_budget;
    b->kern->fn(&a, b->d.n_reaches, b->stream);
  } else {
    fs::KernelArgs<float> a; fill_args(b, n_steps, a);
    a.iter_budget = iter_budget;
    b->kern->fn(&a, b->d.n_reaches, b->stream);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(b->ev1, b->stream));
  b->timed = true; b->launches = 1;
  return 0;
}

}  // namespace

extern "C" {

int fs_abi_version(void) { return FS_ABI_VERSION; }

int fs_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char *fs_last_error(void) { return g_err.c_str(); }

fs_batch *fs_batch_create(const fs_batch_desc *desc) {
  if (!desc) { fail("fs_batch_create: null descriptor"); return nullptr; }
  if (desc->n_reaches < 1 || desc->n_nodes < 2 || desc->max_levels < 2) {
    fail("fs_batch_create: need n_reaches >= 1, n_nodes >= 2, max_levels >= 2"); return nullptr;
  }
  if (desc->dtype != FS_F64 && desc->dtype != FS_F32) { fail("fs_batch_create: bad dtype"); return nullptr; }
  if (desc->section_mode != FS_SEC_RECT_UNIFORM && desc->section_mode != FS_SEC_TRAP_UNIFORM &&
      desc->section_mode != FS_SEC_TABLE && desc->section_mode != FS_SEC_IRREGULAR) {
    fail("fs_batch_create: bad section_mode"); return nullptr;
  }
  if (desc->section_mode == FS_SEC_IRREGULAR && desc->dtype != FS_F64) {
    fail("fs_batch_create: FS_SEC_IRREGULAR is fp64 only"); return nullptr;
  }
  if (fs_device_count() <= desc->device || desc->device < 0) {
    fail("fs_batch_create: no HIP device " + std::to_string(desc->device) +
         " (this library has no CPU path; it needs an MI355X)");
    return nullptr;
  }
  std::string why;
  const Entry *k = pick_kernel(desc->dtype, desc->section_mode, desc->n_nodes, FS_BC_FLOW_HYDROGRAPH, FS_BC_FLOW_HYDROGRAPH, true, &why,
                               false, false);
  if (!k) { fail("fs_batch_create: " + why); return nullptr; }
  fs_batch *b = new fs_batch();
  b->d = *desc;
  b->esz = desc->dtype == FS_F64 ? 8 : 4;
  b->kern = k;
  int prev_dev = -1;
  if (hipGetDevice(&prev_dev) != hipSuccess) prev_dev = -1;
  auto bad = [&](const char *what, hipError_t e) {
    fail(std::string("fs_batch_create: ") + what + ": " + hipGetErrorString(e));
    fs_batch_destroy(b);
    if (prev_dev >= 0) (void)hipSetDevice(prev_dev);
    return (fs_batch *)nullptr;
  };
  hipError_t e;
  if ((e = hipSetDevice(desc->device)) != hipSuccess) return bad("hipSetDevice", e);
  if ((e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking)) != hipSuccess) return bad("hipStreamCreate", e);
  if ((e = hipEventCreate(&b->ev0)) != hipSuccess) return bad("hipEventCreate", e);
  if ((e = hipEventCreate(&b->ev1)) != hipSuccess) return bad("hipEventCreate", e);
  const size_t B = desc->n_reaches, N = desc->n_nodes, L = desc->max_levels;
  void **state[4] = {&b->hk, &b->Qk, &b->hg, &b->Qg};
  for (void **p : state)
    if ((e = hipMalloc(p, B * N * b->esz)) != hipSuccess) return bad("hipMalloc(state)", e);
  if ((e = hipMalloc(&b->hydro, L * 4 * B * b->esz)) != hipSuccess) return bad("hipMalloc(hydro)", e);
  if ((e = hipMalloc((void **)&b->iters, L * B * 4)) != hipSuccess) return bad("hipMalloc(iters)", e);
  if ((e = hipMalloc((void **)&b->status, B * 4)) != hipSuccess) return bad("hipMalloc(status)", e);
  if ((e = hipMalloc(&b->Yprev, B * b->esz)) != hipSuccess) return bad("hipMalloc(Yprev)", e);
  if ((e = hipMalloc(&b->stage_hist, L * B * b->esz)) != hipSuccess) return bad("hipMalloc(stage_hist)", e);
  if ((e = hipMalloc((void **)&b->it_done, B * 4)) != hipSuccess) return bad("hipMalloc(it_done)", e);
  if ((e = hipMemsetAsync(b->stage_hist, 0, L * B * b->esz, b->stream)) != hipSuccess) return bad("hipMemsetAsync", e);
  if ((e = hipMemsetAsync(b->it_done, 0, B * 4, b->stream)) != hipSuccess) return bad("hipMemsetAsync", e);
  if (desc->flags & FS_FLAG_TRACE) {
    if ((e = hipMalloc(&b->trace, L * FS_TRACE_CAP * B * b->esz)) != hipSuccess) return bad("hipMalloc(trace)", e);
    if ((e = hipMemsetAsync(b->trace, 0, L * FS_TRACE_CAP * B * b->esz, b->stream)) != hipSuccess) return bad("hipMemsetAsync", e);
  }
  if (desc->flags & FS_FLAG_HISTORY) {
    if ((e = hipMalloc(&b->hist_h, L * B * N * b->esz)) != hipSuccess) return bad("hipMalloc(history)", e);
    if ((e = hipMalloc(&b->hist_Q, L * B * N * b->esz)) != hipSuccess) return bad("hipMalloc(history)", e);
  }
#ifdef FS_STAMP
  if ((e = hipMalloc((void **)&b->dbg, B * 16 * 12 * 8)) != hipSuccess) return bad("hipMalloc(dbg)", e);
  hipMemsetAsync(b->dbg, 0, B * 16 * 12 * 8, b->stream);
#endif
  if ((e = hipMemsetAsync(b->hydro, 0, L * 4 * B * b->esz, b->stream)) != hipSuccess) return bad("hipMemsetAsync", e);
  if ((e = hipMemsetAsync(b->iters, 0, L * B * 4, b->stream)) != hipSuccess) return bad("hipMemsetAsync", e);
  if ((e = hipMemsetAsync(b->status, 0, B * 4, b->stream)) != hipSuccess) return bad("hipMemsetAsync", e);
  if ((e = hipMemsetAsync(b->Yprev, 0, B * b->esz, b->stream)) != hipSuccess) return bad("hipMemsetAsync", e);
  if ((e = hipStreamSynchronize(b->stream)) != hipSuccess) return bad("hipStreamSynchronize", e);
  if (B * N * b->esz >= kStageMin) (void)b->stage.init();        // large batch: the pinned chunks exist before the first transfer is timed
  (void)hipSetDevice(prev_dev >= 0 ? prev_dev : desc->device);
  return b;
}

void fs_batch_destroy(fs_batch *b) {
  if (!b) return;
  DeviceGuard guard_(b->d.device);
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  void *bufs[] = {b->hk, b->Qk, b->hg, b->Qg, b->geo_uniform, b->geo_table, b->n_override, b->bc_params[0],
                  b->bc_params[1], b->bc_target[0], b->bc_target[1], b->Yprev, b->stage_hist, b->trace, b->hydro, b->hist_h, b->hist_Q,
                  b->iters, b->status, b->poly_x, b->poly_z, b->poly_lim, b->poly_n, b->it_done, b->dbg, b->reach_nodes,
                  b->reach_scheme, b->reach_kinds, b->kc_scratch, b->poly_tz, b->ends_dev, b->open_dev, b->team_mail, b->team_sync};
  for (void *p : bufs) if (p) (void)hipFree(p);
  if (b->rows_stage) (void)hipFree(b->rows_stage);
  if (b->ends_pin) (void)hipHostFree(b->ends_pin);
  if (b->open_pin) (void)hipHostFree(b->open_pin);
  for (void *p : b->derived) if (p) (void)hipFree(p);
  b->stage.release();
  if (b->ev0) (void)hipEventDestroy(b->ev0);
  if (b->ev1) (void)hipEventDestroy(b->ev1);
  if (b->stream) (void)hipStreamDestroy(b->stream);
  delete b;
}

static int rebuild_reach_scheme(fs_batch *b);

int fs_batch_set_scheme(fs_batch *b, double theta, double dt, double dx, double tolerance, int32_t max_iter) {
  if (!b) return fail("null handle");
  if (!(dt > 0) || !(dx > 0) || !(tolerance > 0) || max_iter < 1) return fail("fs_batch_set_scheme: dt, dx, tolerance > 0 and max_iter >= 1 required");
  b->theta = theta; b->dt = dt; b->dx = dx; b->tol = tolerance; b->max_iter = max_iter;
  b->have_scheme = true;
  if (b->reach_scheme) {      // per-reach values stand; the rows the caller left to the batch take the new numbers
    FS_ON_DEVICE(b);
    return rebuild_reach_scheme(b);
  }
  return 0;
}

int fs_batch_set_geometry_uniform(fs_batch *b, const double *params) {
  if (!b || !params) return fail("fs_batch_set_geometry_uniform: null argument");
  const bool trap = b->d.section_mode == FS_SEC_TRAP_UNIFORM;
  FS_ON_DEVICE(b);
  if (b->d.section_mode != FS_SEC_RECT_UNIFORM && !trap) return fail("fs_batch_set_geometry_uniform: batch was created with another section_mode");
  const size_t B = b->d.n_reaches;
  for (size_t i = 0; i < B; ++i) {
    if (!(params[FS_RU_WIDTH * B + i] > 0) || !(params[FS_RU_MANNING * B + i] > 0))
      return fail("fs_batch_set_geometry_uniform: width and Manning n must be positive");
    if (trap && !(params[FS_TU_SIDE_SLOPE * B + i] >= 0)) return fail("fs_batch_set_geometry_uniform: side slope must be >= 0");
  }
  if (upload(b, &b->geo_uniform, params, (size_t)(trap ? FS_TU_NPARAM : FS_RU_NPARAM) * B)) return -1;
  b->have_geo = true;
  return 0;
}

int fs_batch_set_geometry_table(fs_batch *b, const double *table, const double *n_main_override) {
  if (!b || !table) return fail("fs_batch_set_geometry_table: null argument");
  if (b->d.section_mode != FS_SEC_TABLE) return fail("fs_batch_set_geometry_table: batch was created with another section_mode");
  FS_ON_DEVICE(b);
  {
    const std::vector<double> x = extend_table(table, b->d.n_nodes);
    if (b->geo_reach_stride) { (void)hipFree(b->geo_table); b->geo_table = nullptr; b->geo_reach_stride = 0; }
    if (upload(b, &b->geo_table, x.data(), x.size())) return -1;
  }
  if (n_main_override) {
    if (upload(b, &b->n_override, n_main_override, b->d.n_reaches)) return -1;
  } else if (b->n_override) {
    (void)hipFree(b->n_override); b->n_override = nullptr;
  }
  b->have_geo = true;
  return 0;
}

// polylines of one channel: validated and transposed into the vertex-major device layout ([P][N]; unused slots repeat the last
// vertex so that no lane ever reads NaN).  Returns an error text or nullptr.
// Stage table of one polyline (fs_poly.hpp): breakpoints = the distinct vertex elevations; for each interval between two of
// them the polynomial coefficients of A, P, T and of the three roughness strips' (A, P) in u = stage - lower breakpoint, and
// the number of wetted runs of >= 2 vertices.  Built per node as [KP] breakpoints + [P][FS_PT_BLOCK] (each interval's entry also carries its bounds and the node's constants);
// pack_polylines lays the entries out node-minor for the device.
static void build_stage_table(const double *xs, const double *zs, int c, double liml, double limr, int P, double *blk,
                              const double node_const[5] /* n_left, n_main, n_right, curvature, z_min */) {
  std::vector<double> lev(zs, zs + c);
  std::sort(lev.begin(), lev.end());
  lev.erase(std::unique(lev.begin(), lev.end()), lev.end());
  const int K = (int)lev.size(), KP = fs::poly_table_bp(P);
  const double inf = std::numeric_limits<double>::infinity();
  for (int j = 0; j < KP; ++j) blk[j] = j < K ? lev[j] : inf;
  const double xa = xs[0], xb = xs[c - 1];
  for (int k = 0; k < P; ++k) {
    double *co = blk + KP + (size_t)k * fs::FS_PT_BLOCK;
    for (int q = 0; q < fs::FS_PT_BLOCK; ++q) co[q] = 0.0;
    int runs = 0;
    if (k < K) {
      const double z0k = lev[k];
      auto wet = [&](int v) { return zs[v] <= z0k; };          // wet for every stage of the open interval above lev[k]
      for (int e = 0; e + 1 < c; ++e) {
        const double x0 = xs[e], x1 = xs[e + 1], za = zs[e], zb = zs[e + 1];
        const double dx = x1 - x0, dz = zb - za, len = std::sqrt(dx * dx + dz * dz);
        const bool w0 = wet(e), w1 = wet(e + 1);
        double a0 = 0, a1 = 0, a2 = 0, p0 = 0, p1 = 0, t0 = 0, t1 = 0;
        if (w0 && w1) {                                        // A = dx (s - zmid) = dx (z0k - zmid) + dx u
          a0 = dx * (z0k - 0.5 * (za + zb)); a1 = dx; p0 = len; t0 = dx;
        } else if (w0 != w1) {                                 // water's edge: (dx / 2|dz|) (s - zw)^2, (len / |dz|) (s - zw), (dx / |dz|) (s - zw)
          const double zw = w0 ? za : zb, adz = std::fabs(dz), d = z0k - zw;
          const double cA = 0.5 * dx / adz, cP = len / adz, cT = dx / adz;
          a0 = cA * d * d; a1 = 2.0 * cA * d; a2 = cA; p0 = cP * d; p1 = cP; t0 = cT * d; t1 = cT;
        } else {
          continue;
        }
        co[fs::FS_PT_A0] += a0; co[fs::FS_PT_A1] += a1; co[fs::FS_PT_A2] += a2; co[fs::FS_PT_P0] += p0; co[fs::FS_PT_P1] += p1;
        co[fs::FS_PT_T0] += t0; co[fs::FS_PT_T1] += t1;
        const bool in[3] = {x0 >= xa && x1 <= liml, x0 >= liml && x1 <= limr, x0 >= limr && x1 <= xb};      // cross_section.py:459
        for (int sidx = 0; sidx < 3; ++sidx)
          if (in[sidx]) {
            double *o = co + fs::FS_PT_STRIP + 5 * sidx;
            o[0] += a0; o[1] += a1; o[2] += a2; o[3] += p0; o[4] += p1;
          }
      }
      int run = 0;
      for (int v = 0; v < c; ++v) {
        if (wet(v)) ++run;
        if (!wet(v) || v == c - 1) { runs += run >= 2; run = 0; }
      }
    }
    co[fs::FS_PT_NSUB] = (double)runs;
    // what an evaluation that starts from this interval needs besides the coefficients (fs_poly.hpp: node_terms_poly_hinted)
    co[fs::FS_PT_ZLO] = k < K ? lev[k] : inf; co[fs::FS_PT_ZHI] = k + 1 < K ? lev[k + 1] : inf;
    for (int q = 0; q < 5; ++q) co[fs::FS_PT_NL + q] = node_const[q];
  }
}

static const char *pack_polylines(const double *table, const int32_t *n_pts, int32_t max_pts, const double *x, const double *z,
                                  const double *limits, size_t N, double *xt, double *zt, double *lim, double *tz) {
  const size_t P = max_pts;
  for (size_t i = 0; i < N; ++i) {
    const int c = n_pts[i];
    if (c == 0) continue;
    if (c < 2 || c > max_pts) return "fs_batch_set_geometry_irregular: n_pts must be 0 or 2..max_pts";
    double zmin = z[i * P];
    for (int j = 0; j < c; ++j) {
      const double xv = x[i * P + j], zv = z[i * P + j];
      if (!(xv == xv) || !(zv == zv)) return "x and z must have the same shape";            // cross_section.py:222 (NaN padding inside the count)
      if (j && xv < x[i * P + j - 1]) return "fs_batch_set_geometry_irregular: x must be ascending (IrregularSection sorts it, cross_section.py:231)";
      zmin = zv < zmin ? zv : zmin;
    }
    if (table[(size_t)FS_GEO_Z_BED * N + i] != zmin)
      return "fs_batch_set_geometry_irregular: table row Z_BED must hold min(z) of a polyline node (IrregularSection.z_min)";
    for (size_t j = 0; j < P; ++j) {
      const size_t src = i * P + (j < (size_t)c ? j : (size_t)c - 1);
      xt[j * N + i] = x[src]; zt[j * N + i] = z[src];
    }
    lim[i] = limits[2 * i]; lim[N + i] = limits[2 * i + 1];
    const double node_const[5] = {table[(size_t)FS_GEO_N_LEFT * N + i], table[(size_t)FS_GEO_N_MAIN * N + i],
                                  table[(size_t)FS_GEO_N_RIGHT * N + i], table[(size_t)FS_GEO_CURVATURE * N + i], zmin};
    if (!tz) continue;                 // no stage tables for this batch (set_irregular): the kernels walk the edges
    // the node's table, then into the device layout: breakpoints [N][KP], intervals [P][FS_PT_BLOCK / 2][N] pairs (fs_poly.hpp)
    const size_t KP = fs::poly_table_bp(max_pts);
    std::vector<double> blk(fs::poly_table_stride(max_pts));
    build_stage_table(x + i * P, z + i * P, c, limits[2 * i], limits[2 * i + 1], max_pts, blk.data(), node_const);
    std::memcpy(tz + i * KP, blk.data(), KP * sizeof(double));
    double *co = tz + N * KP;
    for (size_t q = 0; q < P * fs::FS_PT_BLOCK; q += 2) {
      co[((q / 2) * N + i) * 2] = blk[KP + q]; co[((q / 2) * N + i) * 2 + 1] = blk[KP + q + 1];
    }
  }
  return nullptr;
}

// n_sets = 1: one channel shared by the batch; n_sets = B: one per reach (tables [B][NPARAM][N], n_pts [B][N], x / z [B][N][P],
// limits [B][N][2])
static int set_irregular(fs_batch *b, const double *table, const int32_t *n_pts, int32_t max_pts, const double *x, const double *z,
                         const double *limits, const double *n_main_override, size_t n_sets) {
  if (!b || !table || !n_pts || !x || !z || !limits) return fail("fs_batch_set_geometry_irregular: null argument");
  if (b->d.section_mode != FS_SEC_IRREGULAR) return fail("fs_batch_set_geometry_irregular: batch was created with another section_mode");
  if (max_pts < 2) return fail("fs_batch_set_geometry_irregular: max_pts must be >= 2");
  FS_ON_DEVICE(b);
  const size_t N = b->d.n_nodes, P = max_pts, per = (size_t)fs::FS_GEOX_NROWS * N;
  std::vector<double> xt(n_sets * P * N, 0.0), zt(n_sets * P * N, 0.0), lim(n_sets * 2 * N, 0.0), tabs(n_sets * per);
  // Stage tables: poly_table_stride(P) numbers per node (about 10 KB at 40 stations) - times N, times one set per reach for
  // per-reach channels.  Beyond a bound (FS_POLY_TABLE_MAX_BYTES, default 8 GiB: staged once on the host, then resident in HBM)
  // the batch gets no tables and every evaluation walks its polyline's edges (fs_poly.hpp: poly_K = 0, the round-2 path - same
  // results, about five times the instructions); FS_POLY_WALK=1 forces that path.  fs_batch_poly_tables() says which one it is.
  const size_t tstride = (size_t)fs::poly_table_stride(max_pts);
  size_t table_cap = (size_t)8 << 30;
  if (const char *env = std::getenv("FS_POLY_TABLE_MAX_BYTES")) table_cap = (size_t)std::strtoull(env, nullptr, 10);
  const double table_bytes = (double)n_sets * (double)N * (double)tstride * sizeof(double);
  const bool walk = std::getenv("FS_POLY_WALK") != nullptr || table_bytes > (double)table_cap;
  std::vector<double> tz;
  try {
    if (!walk) tz.assign(n_sets * N * tstride, std::numeric_limits<double>::infinity());
  } catch (const std::bad_alloc &) {
    return fail("fs_batch_set_geometry_irregular: no host memory to stage " + std::to_string((size_t)(table_bytes / (1 << 20))) +
                " MiB of stage tables (lower FS_POLY_TABLE_MAX_BYTES to fall back to the edge walk)");
  }
  for (size_t r = 0; r < n_sets; ++r) {
    const double *tab_r = table + r * FS_GEO_NPARAM * N;
    if (const char *err = pack_polylines(tab_r, n_pts + r * N, max_pts, x + r * N * P, z + r * N * P, limits + r * 2 * N, N,
                                         xt.data() + r * P * N, zt.data() + r * P * N, lim.data() + r * 2 * N,
                                         walk ? nullptr : tz.data() + r * N * tstride))
      return fail(err);
    const std::vector<double> ext = extend_table(tab_r, N);
    std::memcpy(tabs.data() + r * per, ext.data(), per * sizeof(double));
  }
  void **old[] = {&b->geo_table, &b->poly_x, &b->poly_z, &b->poly_lim, &b->poly_tz};
  for (void **q : old) if (*q) { (void)hipFree(*q); *q = nullptr; }
  if (b->poly_n) { (void)hipFree(b->poly_n); b->poly_n = nullptr; }
  b->poly_K = 0;
  if (!walk) {
    if (upload(b, &b->poly_tz, tz.data(), tz.size()))
      return fail("fs_batch_set_geometry_irregular: " + std::to_string((size_t)(table_bytes / (1 << 20))) + " MiB of stage tables do not fit the "
                  "device (" + g_err + "); lower FS_POLY_TABLE_MAX_BYTES to fall back to the edge walk");
    b->poly_K = (int)P;
  }
  if (upload(b, &b->geo_table, tabs.data(), tabs.size()) || upload(b, &b->poly_x, xt.data(), xt.size()) ||
      upload(b, &b->poly_z, zt.data(), zt.size()) || upload(b, &b->poly_lim, lim.data(), lim.size())) return -1;
  HIP_TRY(hipMalloc((void **)&b->poly_n, n_sets * N * 4));
  HIP_TRY(hipMemcpy(b->poly_n, n_pts, n_sets * N * 4, hipMemcpyHostToDevice));
  b->geo_reach_stride = n_sets > 1 ? per : 0;
  b->poly_reach_stride = n_sets > 1 ? P * N : 0;
  if (n_main_override) {
    if (upload(b, &b->n_override, n_main_override, b->d.n_reaches)) return -1;
  } else if (b->n_override) {
    (void)hipFree(b->n_override); b->n_override = nullptr;
  }
  b->have_geo = true;
  return 0;
}

int fs_batch_set_geometry_irregular(fs_batch *b, const double *table, const int32_t *n_pts, int32_t max_pts,
                                    const double *x, const double *z, const double *limits,
                                    const double *n_main_override) {
  return set_irregular(b, table, n_pts, max_pts, x, z, limits, n_main_override, 1);
}

int fs_batch_set_geometry_irregular_per_reach(fs_batch *b, const double *tables, const int32_t *n_pts, int32_t max_pts,
                                              const double *x, const double *z, const double *limits,
                                              const double *n_main_override) {
  if (!b) return fail("fs_batch_set_geometry_irregular: null argument");
  if (b->d.n_reaches == 1) return set_irregular(b, tables, n_pts, max_pts, x, z, limits, n_main_override, 1);
  return set_irregular(b, tables, n_pts, max_pts, x, z, limits, n_main_override, (size_t)b->d.n_reaches);
}

// One TrapezoidalSection table per reach: tables[B][FS_GEO_NPARAM][N] on the host -> [B][FS_GEOX_NROWS][N] on the device (a
// reach's workgroup reads its own table, lanes along the nodes: coalesced; once per launch in the two-rows-per-lane kernels)
int fs_batch_set_geometry_table_per_reach(fs_batch *b, const double *tables, const double *n_main_override) {
  if (!b || !tables) return fail("fs_batch_set_geometry_table_per_reach: null argument");
  if (b->d.section_mode != FS_SEC_TABLE) return fail("fs_batch_set_geometry_table_per_reach: batch was created with another section_mode");
  FS_ON_DEVICE(b);
  const size_t B = b->d.n_reaches, N = b->d.n_nodes, per = (size_t)fs::FS_GEOX_NROWS * N;
  std::vector<double> all(B * per);
  for (size_t r = 0; r < B; ++r) {
    const std::vector<double> x = extend_table(tables + r * FS_GEO_NPARAM * N, N);
    std::memcpy(all.data() + r * per, x.data(), per * sizeof(double));
  }
  if (b->geo_table) { (void)hipFree(b->geo_table); b->geo_table = nullptr; }
  if (upload(b, &b->geo_table, all.data(), all.size())) return -1;
  b->geo_reach_stride = per;
  if (n_main_override) {
    if (upload(b, &b->n_override, n_main_override, B)) return -1;
  } else if (b->n_override) {
    (void)hipFree(b->n_override); b->n_override = nullptr;
  }
  b->have_geo = true;
  return 0;
}

int fs_batch_set_reach_nodes(fs_batch *b, const int32_t *n_nodes) {
  if (!b) return fail("null handle");
  FS_ON_DEVICE(b);
  const size_t B = b->d.n_reaches;
  // the level-0 hydrograph row holds each reach's LAST node (fs_batch_set_state): when the counts change under a state that is
  // already on the device and has not been stepped, the row follows them
  auto refresh_row0 = [&]() -> int {
    if (!b->have_state || b->level != 0) return 0;
    const dim3 grid((unsigned)((B + 255) / 256));
    if (b->d.dtype == FS_F64)
      hipLaunchKernelGGL((refresh_level0_downstream<double>), grid, dim3(256), 0, b->stream, (const double *)b->hk, (const double *)b->Qk,
                         b->reach_nodes, (double *)b->hydro, B, (size_t)b->d.n_nodes);
    else
      hipLaunchKernelGGL((refresh_level0_downstream<float>), grid, dim3(256), 0, b->stream, (const float *)b->hk, (const float *)b->Qk,
                         b->reach_nodes, (float *)b->hydro, B, (size_t)b->d.n_nodes);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(b->stream));
    return 0;
  };
  if (!n_nodes) {
    if (b->reach_nodes) { (void)hipFree(b->reach_nodes); b->reach_nodes = nullptr; }
    b->reach_nodes_host.clear();
    return refresh_row0();
  }
  for (size_t r = 0; r < B; ++r)
    if (n_nodes[r] < 2 || n_nodes[r] > b->d.n_nodes) return fail("fs_batch_set_reach_nodes: every reach needs 2 <= nodes <= n_nodes of the batch");
  if (!b->reach_nodes) HIP_TRY(hipMalloc((void **)&b->reach_nodes, B * 4));
  HIP_TRY(hipMemcpy(b->reach_nodes, n_nodes, B * 4, hipMemcpyHostToDevice));
  b->reach_nodes_host.assign(n_nodes, n_nodes + B);
  return refresh_row0();
}

// the [5][B] array the kernels of boundary classes 0 and -1 read: per-reach values where the caller gave them, the batch's elsewhere
static int rebuild_reach_scheme(fs_batch *b) {
  const size_t B = b->d.n_reaches;
  if (b->reach_scheme) { (void)hipFree(b->reach_scheme); b->reach_scheme = nullptr; }
  bool any = false;
  for (auto &v : b->rs_host) any = any || !v.empty();
  if (!any) return 0;
  const double wide[5] = {b->theta, b->dt, b->dx, b->tol, (double)b->max_iter};
  std::vector<double> v(5 * B);
  for (int i = 0; i < 5; ++i)
    for (size_t r = 0; r < B; ++r) v[i * B + r] = b->rs_host[i].empty() ? wide[i] : b->rs_host[i][r];
  return upload(b, &b->reach_scheme, v.data(), v.size());
}

int fs_batch_set_reach_scheme(fs_batch *b, const double *theta, const double *dt, const double *dx) {
  if (!b) return fail("null handle");
  FS_ON_DEVICE(b);
  const size_t B = b->d.n_reaches;
  if ((theta || dt || dx) && !b->have_scheme)
    return fail("fs_batch_set_reach_scheme: call fs_batch_set_scheme first (tolerance, max_iter and the values of the arrays left NULL)");
  for (size_t r = 0; r < B; ++r)
    if ((dt && !(dt[r] > 0)) || (dx && !(dx[r] > 0))) return fail("fs_batch_set_reach_scheme: dt and dx must be positive");
  const double *src[3] = {theta, dt, dx};
  for (int i = 0; i < 3; ++i) {
    if (src[i]) b->rs_host[i].assign(src[i], src[i] + B); else b->rs_host[i].clear();
  }
  return rebuild_reach_scheme(b);
}

int fs_batch_set_reach_tolerance(fs_batch *b, const double *tolerance, const int32_t *max_iter) {
  if (!b) return fail("null handle");
  FS_ON_DEVICE(b);
  const size_t B = b->d.n_reaches;
  if ((tolerance || max_iter) && !b->have_scheme) return fail("fs_batch_set_reach_tolerance: call fs_batch_set_scheme first");
  for (size_t r = 0; r < B; ++r)
    if ((tolerance && !(tolerance[r] > 0)) || (max_iter && max_iter[r] < 1)) return fail("fs_batch_set_reach_tolerance: tolerance > 0 and max_iter >= 1 required");
  if (tolerance) b->rs_host[3].assign(tolerance, tolerance + B); else b->rs_host[3].clear();
  if (max_iter) b->rs_host[4].assign(max_iter, max_iter + B); else b->rs_host[4].clear();
  return rebuild_reach_scheme(b);
}

int fs_batch_set_bc_per_reach(fs_batch *b, int32_t side, const int32_t *kinds, const double *params, const double *target) {
  return fs_batch_set_bc_per_reach_wide(b, side, kinds, params, FS_BC_MAX_PARAMS, target);
}

int fs_batch_set_bc_per_reach_wide(fs_batch *b, int32_t side, const int32_t *kinds, const double *params, int32_t n_params,
                                   const double *target) {
  if (!b || !kinds || !params) return fail("fs_batch_set_bc_per_reach: null argument");
  if (side != FS_UPSTREAM && side != FS_DOWNSTREAM) return fail("fs_batch_set_bc_per_reach: side must be FS_UPSTREAM or FS_DOWNSTREAM");
  if (n_params < FS_BC_MAX_PARAMS) return fail("fs_batch_set_bc_per_reach: at least FS_BC_MAX_PARAMS parameter rows");
  FS_ON_DEVICE(b);
  const size_t B = b->d.n_reaches;
  const bool tables = b->d.section_mode == FS_SEC_TABLE || b->d.section_mode == FS_SEC_IRREGULAR;
  bool need_target = false;
  for (size_t r = 0; r < B; ++r) {
    if (kinds[r] == FS_BC_STORAGE_CURVE) {   // a general reservoir behind THIS reach: its FS_SC_* rows, its own area curve of its own length
      if (!tables) return fail("fs_batch_set_bc_per_reach: FS_BC_STORAGE_CURVE needs section mode FS_SEC_TABLE or FS_SEC_IRREGULAR");
      if (side != FS_DOWNSTREAM) return fail("fs_batch_set_bc: the storage boundary is downstream only");
      auto at = [&](int i) { return params[(size_t)i * B + r]; };
      const int nc = n_params > FS_SC_N_CURVE ? (int)at(FS_SC_N_CURVE) : -1;
      if (nc < 0 || nc == 1 || FS_SC_NFIXED + 2 * nc > n_params)
        return fail("fs_batch_set_bc_per_reach: an FS_BC_STORAGE_CURVE reach needs FS_SC_NFIXED + 2*n_curve parameter rows (n_curve 0 or >= 2)");
      for (int j = 0; j + 1 < nc; ++j)
        if (!(at(FS_SC_NFIXED + j + 1) > at(FS_SC_NFIXED + j))) return fail("fs_batch_set_bc: area-curve stages must be increasing");
      if (nc == 0 && !(at(FS_SC_SURFACE_AREA) > 0)) return fail("Insufficient arguments for boundary condition.");
      continue;
    }
    if (kinds[r] == FS_BC_HOST_ROW) {      // a plugin without a device form on THIS reach (round 4: until then a batch-wide kind only)
      if (!tables) return fail("fs_batch_set_bc_per_reach: FS_BC_HOST_ROW needs section mode FS_SEC_TABLE or FS_SEC_IRREGULAR");
      continue;
    }
    if (kinds[r] < 0 || kinds[r] > FS_BC_STORAGE) return fail("Invalid boundary condition.");        // boundary.py:33
    if (kinds[r] == FS_BC_STORAGE && side != FS_DOWNSTREAM) return fail("fs_batch_set_bc: the storage boundary is downstream only");
    need_target = need_target || kinds[r] == FS_BC_FLOW_HYDROGRAPH || kinds[r] == FS_BC_STAGE_HYDROGRAPH;
  }
  if (need_target && !target) return fail("Insufficient arguments for boundary condition.");                     // boundary.py:87
  if (b->bc_params[side]) { (void)hipFree(b->bc_params[side]); b->bc_params[side] = nullptr; }
  if (b->bc_target[side]) { (void)hipFree(b->bc_target[side]); b->bc_target[side] = nullptr; }
  if (upload(b, &b->bc_params[side], params, (size_t)n_params * B)) return -1;
  if (target && upload(b, &b->bc_target[side], target, (size_t)b->d.max_levels * B)) return -1;
  if (!b->reach_kinds) {
    HIP_TRY(hipMalloc((void **)&b->reach_kinds, 2 * B * 4));
    HIP_TRY(hipMemset(b->reach_kinds, 0, 2 * B * 4));
  }
  HIP_TRY(hipMemcpy(b->reach_kinds + (size_t)side * B, kinds, B * 4, hipMemcpyHostToDevice));
  // the other side, if it was set for the whole batch, keeps its one kind in every slot
  const int other = 1 - side;
  if (b->have_bc[other] && !b->kinds_per_reach[other]) {
    std::vector<int32_t> same(B, b->bc_kind[other]);
    HIP_TRY(hipMemcpy(b->reach_kinds + (size_t)other * B, same.data(), B * 4, hipMemcpyHostToDevice));
  }
  b->bc_kind[side] = kinds[0]; b->bc_stride[side] = 1; b->kinds_per_reach[side] = true;
  b->any_storage[side] = false; b->some_host_rows[side] = false;
  for (size_t r = 0; r < B; ++r) {
    b->any_storage[side] = b->any_storage[side] || fs::bc_is_storage(kinds[r]);
    b->some_host_rows[side] = b->some_host_rows[side] || kinds[r] == FS_BC_HOST_ROW;
  }
  // one host-evaluated reach makes the batch one that advances with fs_batch_iterate on the kernels of boundary class -1, one general
  // reservoir makes it one for those kernels too: the side's representative kind (what the dispatch and fs_batch_step look at) says so
  for (size_t r = 0; r < B; ++r) if (kinds[r] == FS_BC_STORAGE_CURVE) b->bc_kind[side] = FS_BC_STORAGE_CURVE;
  if (b->some_host_rows[side]) b->bc_kind[side] = FS_BC_HOST_ROW;
  b->have_bc[side] = true;
  return 0;
}

int fs_batch_set_bc(fs_batch *b, int32_t side, int32_t kind, const double *params, int32_t n_params,
                    int32_t per_reach, const double *target) {
  if (!b) return fail("null handle");
  if (side != FS_UPSTREAM && side != FS_DOWNSTREAM) return fail("fs_batch_set_bc: side must be FS_UPSTREAM or FS_DOWNSTREAM");
  FS_ON_DEVICE(b);
  static const int need[] = {0, 1, 1, 2, 4, 5, 10, 5};
  if (kind < 0 || kind > FS_BC_HOST_ROW) return fail("Invalid boundary condition.");        // boundary.py:33
  if (kind == FS_BC_HOST_ROW) {
    if (b->d.section_mode != FS_SEC_TABLE && b->d.section_mode != FS_SEC_IRREGULAR)
      return fail("fs_batch_set_bc: FS_BC_HOST_ROW needs section mode FS_SEC_TABLE or FS_SEC_IRREGULAR");
    if (n_params != 3 || !per_reach) return fail("fs_batch_set_bc: FS_BC_HOST_ROW takes params[3][B] (per_reach = 1) or NULL");
    const size_t B = b->d.n_reaches;
    if (b->bc_params[side]) { (void)hipFree(b->bc_params[side]); b->bc_params[side] = nullptr; }
    if (b->bc_target[side]) { (void)hipFree(b->bc_target[side]); b->bc_target[side] = nullptr; }
    if (params) { if (upload(b, &b->bc_params[side], params, 3 * B)) return -1; }
    else {
      HIP_TRY(hipMalloc(&b->bc_params[side], 3 * B * b->esz));
      HIP_TRY(hipMemsetAsync(b->bc_params[side], 0, 3 * B * b->esz, b->stream));
    }
    b->bc_kind[side] = kind; b->bc_stride[side] = 1; b->have_bc[side] = true;
    b->kinds_per_reach[side] = false; b->any_storage[side] = false; b->some_host_rows[side] = false;
    if (b->reach_kinds) {          // the other side has per-reach kinds: this side's one kind goes into every slot
      std::vector<int32_t> same(B, kind);
      HIP_TRY(hipMemcpy(b->reach_kinds + (size_t)side * B, same.data(), B * 4, hipMemcpyHostToDevice));
    }
    return 0;
  }
  if (kind == FS_BC_STORAGE_CURVE) {
    // shared by the batch (params[n_params]) or - round 4 - one reservoir per reach (per_reach = 1: params[n_params][B], every reach its own
    // scalars, area curve and outflow rating curve; the curves of a batch have the same number of points)
    if (!params || n_params < FS_SC_NFIXED) return fail("Insufficient arguments for boundary condition.");
    const size_t Bn = per_reach ? (size_t)b->d.n_reaches : 1;
    auto at = [&](int i, size_t r) { return per_reach ? params[(size_t)i * Bn + r] : params[i]; };
    for (size_t r = 0; r < Bn; ++r) {
      const int nc = (int)at(FS_SC_N_CURVE, r);
      if (nc < 0 || nc == 1 || n_params != FS_SC_NFIXED + 2 * nc)
        return fail("fs_batch_set_bc: FS_BC_STORAGE_CURVE needs FS_SC_NFIXED + 2*n_curve parameters (n_curve 0 or >= 2; per reach: the same n_curve for all)");
      for (int j = 0; j + 1 < nc; ++j)
        if (!(at(FS_SC_NFIXED + j + 1, r) > at(FS_SC_NFIXED + j, r)))
          return fail("fs_batch_set_bc: area-curve stages must be increasing");
      if (nc == 0 && !(at(FS_SC_SURFACE_AREA, r) > 0)) return fail("Insufficient arguments for boundary condition.");
    }
  } else if (n_params != need[kind]) return fail("Insufficient arguments for boundary condition.");      // boundary.py:83
  if (n_params > 0 && !params) return fail("Insufficient arguments for boundary condition.");
  if ((kind == FS_BC_FLOW_HYDROGRAPH || kind == FS_BC_STAGE_HYDROGRAPH) && !target)
    return fail("Insufficient arguments for boundary condition.");                                // boundary.py:87
  if (fs::bc_is_storage(kind) && side != FS_DOWNSTREAM) return fail("fs_batch_set_bc: the storage boundary is downstream only");
  const size_t B = b->d.n_reaches;
  if (b->bc_params[side]) { (void)hipFree(b->bc_params[side]); b->bc_params[side] = nullptr; }
  if (b->bc_target[side]) { (void)hipFree(b->bc_target[side]); b->bc_target[side] = nullptr; }
  if (n_params > 0 && upload(b, &b->bc_params[side], params, per_reach ? n_params * B : (size_t)n_params)) return -1;
  if (target && upload(b, &b->bc_target[side], target, (size_t)b->d.max_levels * B)) return -1;
  b->bc_kind[side] = kind; b->bc_stride[side] = per_reach ? 1 : 0;
  b->have_bc[side] = true;
  b->kinds_per_reach[side] = false; b->any_storage[side] = fs::bc_is_storage(kind); b->some_host_rows[side] = false;
  if (b->reach_kinds) {          // the other side has per-reach kinds: this side's one kind goes into every slot
    std::vector<int32_t> same(B, kind);
    HIP_TRY(hipMemcpy(b->reach_kinds + (size_t)side * B, same.data(), B * 4, hipMemcpyHostToDevice));
  }
  return 0;
}

int fs_batch_set_state(fs_batch *b, const double *h, const double *Q) {
  if (!b || !h || !Q) return fail("fs_batch_set_state: null argument");
  FS_ON_DEVICE(b);
  const size_t B = b->d.n_reaches, N = b->d.n_nodes;
  // each array crosses the bus once; the Newton start vector and level 0 of the history are device-to-device copies
  if (upload(b, &b->hk, h, B * N) || upload(b, &b->Qk, Q, B * N)) return -1;
  HIP_TRY(hipMemcpyAsync(b->hg, b->hk, B * N * b->esz, hipMemcpyDeviceToDevice, b->stream));
  HIP_TRY(hipMemcpyAsync(b->Qg, b->Qk, B * N * b->esz, hipMemcpyDeviceToDevice, b->stream));
  if (b->hist_h) {   // level 0 of the history = initial conditions (solver.py:61-63)
    HIP_TRY(hipMemcpyAsync(b->hist_h, b->hk, B * N * b->esz, hipMemcpyDeviceToDevice, b->stream));
    HIP_TRY(hipMemcpyAsync(b->hist_Q, b->Qk, B * N * b->esz, hipMemcpyDeviceToDevice, b->stream));
  }
  std::vector<double> row(4 * B);
  for (size_t r = 0; r < B; ++r) {
    const size_t last = (b->reach_nodes_host.empty() ? N : (size_t)b->reach_nodes_host[r]) - 1;     // the reach's own last node, not the caller's padding
    row[0 * B + r] = h[r * N]; row[1 * B + r] = Q[r * N];
    row[2 * B + r] = h[r * N + last]; row[3 * B + r] = Q[r * N + last];
  }
  void *p = b->hydro;
  if (upload(b, &p, row.data(), 4 * B)) return -1;
  HIP_TRY(hipMemsetAsync(b->status, 0, B * 4, b->stream));
  HIP_TRY(hipMemsetAsync(b->iters, 0, (size_t)b->d.max_levels * B * 4, b->stream));
  HIP_TRY(hipMemsetAsync(b->Yprev, 0, B * b->esz, b->stream));
  if (b->trace) HIP_TRY(hipMemsetAsync(b->trace, 0, (size_t)b->d.max_levels * FS_TRACE_CAP * B * b->esz, b->stream));
  HIP_TRY(hipMemsetAsync(b->it_done, 0, B * 4, b->stream));
  b->level = 0; b->iterating = false; b->restart_level = 0;
  b->have_state = true;
  return 0;
}

int fs_batch_set_state_uniform(fs_batch *b, const double *h, const double *Q) {
  if (!b || !h || !Q) return fail("fs_batch_set_state_uniform: null argument");
  FS_ON_DEVICE(b);
  const size_t B = b->d.n_reaches, N = b->d.n_nodes;
  void *dh = nullptr, *dQ = nullptr;
  struct Tmp { void *&a, *&c; ~Tmp() { if (a) (void)hipFree(a); if (c) (void)hipFree(c); } } tmp_{dh, dQ};
  if (upload(b, &dh, h, B) || upload(b, &dQ, Q, B)) return -1;
  const dim3 grid((unsigned)((B * N + 255) / 256));
  if (b->d.dtype == FS_F64)
    hipLaunchKernelGGL((broadcast_state<double>), grid, dim3(256), 0, b->stream, (const double *)dh, (const double *)dQ,
                       (double *)b->hk, (double *)b->Qk, (double *)b->hg, (double *)b->Qg, (double *)b->hist_h,
                       (double *)b->hist_Q, (double *)b->hydro, B, N);
  else
    hipLaunchKernelGGL((broadcast_state<float>), grid, dim3(256), 0, b->stream, (const float *)dh, (const float *)dQ,
                       (float *)b->hk, (float *)b->Qk, (float *)b->hg, (float *)b->Qg, (float *)b->hist_h,
                       (float *)b->hist_Q, (float *)b->hydro, B, N);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemsetAsync(b->status, 0, B * 4, b->stream));
  HIP_TRY(hipMemsetAsync(b->iters, 0, (size_t)b->d.max_levels * B * 4, b->stream));
  HIP_TRY(hipMemsetAsync(b->Yprev, 0, B * b->esz, b->stream));
  HIP_TRY(hipMemsetAsync(b->it_done, 0, B * 4, b->stream));
  if (b->trace) HIP_TRY(hipMemsetAsync(b->trace, 0, (size_t)b->d.max_levels * FS_TRACE_CAP * B * b->esz, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  b->level = 0; b->iterating = false; b->restart_level = 0;
  b->have_state = true;
  return 0;
}

int fs_batch_step(fs_batch *b, int32_t n_steps) {
  if (!b) return fail("null handle");
  if (!b->have_scheme || !b->have_geo || !b->have_state || !b->have_bc[0] || !b->have_bc[1])
    return fail("fs_batch_step: scheme, geometry, both boundaries and the initial state must be set first");
  if (n_steps < 1) return fail("fs_batch_step: n_steps must be >= 1");
  if (b->level + n_steps >= b->d.max_levels) return fail("fs_batch_step: would run past max_levels");
  if (b->bc_kind[0] == FS_BC_HOST_ROW || b->bc_kind[1] == FS_BC_HOST_ROW)
    return fail("fs_batch_step: a batch with FS_BC_HOST_ROW boundaries advances with fs_batch_iterate (the caller evaluates the rows "
                "before every Newton iteration)");
  if (b->iterating) return fail("fs_batch_step: a level opened with fs_batch_iterate must be closed with it first");
  FS_ON_DEVICE(b);
  if (launch_steps(b, n_steps, 0)) return -1;
  b->level += n_steps;
  return 0;
}

int fs_batch_iterate(fs_batch *b, int32_t *n_open) {
  if (!b) return fail("null handle");
  if (!b->have_scheme || !b->have_geo || !b->have_state || !b->have_bc[0] || !b->have_bc[1])
    return fail("fs_batch_iterate: scheme, geometry, both boundaries and the initial state must be set first");
  if (b->level + 1 >= b->d.max_levels) return fail("fs_batch_iterate: would run past max_levels");
  if (b->d.section_mode != FS_SEC_TABLE && b->d.section_mode != FS_SEC_IRREGULAR)
    return fail("fs_batch_iterate: section mode FS_SEC_TABLE or FS_SEC_IRREGULAR required");
  FS_ON_DEVICE(b);
  if (launch_steps(b, 1, 1)) return -1;
  b->iterating = true;
  // how many reaches are still open: counted on the device, four bytes come back (until round 3: two B-sized downloads per iteration)
  const size_t B = b->d.n_reaches;
  if (!b->open_dev) {
    HIP_TRY(hipMalloc((void **)&b->open_dev, sizeof(int32_t)));
    HIP_TRY(hipHostMalloc((void **)&b->open_pin, sizeof(int32_t), hipHostMallocDefault));
  }
  HIP_TRY(hipMemsetAsync(b->open_dev, 0, sizeof(int32_t), b->stream));
  hipLaunchKernelGGL(count_open_reaches, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, b->stream, b->it_done, b->status, b->open_dev, B);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(b->open_pin, b->open_dev, sizeof(int32_t), hipMemcpyDeviceToHost, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  const int32_t open = *b->open_pin;
  if (open == 0) {      // every reach has accepted the level (or failed on it): next level, counters back to zero
    b->level += 1;
    b->iterating = false;
    HIP_TRY(hipMemsetAsync(b->it_done, 0, B * 4, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
  }
  if (n_open) *n_open = open;
  return 0;
}

int fs_batch_set_host_rows(fs_batch *b, int32_t side, const double *rows) {
  if (!b || !rows) return fail("fs_batch_set_host_rows: null argument");
  if (side != FS_UPSTREAM && side != FS_DOWNSTREAM) return fail("fs_batch_set_host_rows: side must be FS_UPSTREAM or FS_DOWNSTREAM");
  if (!b->have_bc[side] || b->bc_kind[side] != FS_BC_HOST_ROW) return fail("fs_batch_set_host_rows: this side is not an FS_BC_HOST_ROW boundary");
  FS_ON_DEVICE(b);
  const size_t B = b->d.n_reaches;
  if (!b->kinds_per_reach[side]) return upload(b, &b->bc_params[side], rows, 3 * B);
  // per-reach kinds: entries of reaches whose boundary the device evaluates are ignored
  if (upload(b, &b->rows_stage, rows, 3 * B)) return -1;
  const dim3 grid((unsigned)((B + 255) / 256));
  if (b->d.dtype == FS_F64)
    hipLaunchKernelGGL((merge_host_rows<double>), grid, dim3(256), 0, b->stream, (double *)b->bc_params[side], (const double *)b->rows_stage,
                       b->reach_kinds + (size_t)side * B, B);
  else
    hipLaunchKernelGGL((merge_host_rows<float>), grid, dim3(256), 0, b->stream, (float *)b->bc_params[side], (const float *)b->rows_stage,
                       b->reach_kinds + (size_t)side * B, B);
  HIP_TRY(hipGetLastError());
  return 0;
}

int fs_batch_get_boundary_iterate(fs_batch *b, double *out) {
  if (!b || !out) return fail("fs_batch_get_boundary_iterate: null argument");
  if (!b->have_state) return fail("fs_batch_get_boundary_iterate: no state yet");
  FS_ON_DEVICE(b);
  // one gather kernel and one transfer through a pinned buffer per call (it is made once per Newton iteration): until round 3 this
  // was four strided hipMemcpy2D of one element per reach
  const size_t B = b->d.n_reaches, N = b->d.n_nodes;
  if (!b->ends_dev) {
    HIP_TRY(hipMalloc((void **)&b->ends_dev, 4 * B * sizeof(double)));
    HIP_TRY(hipHostMalloc((void **)&b->ends_pin, 4 * B * sizeof(double), hipHostMallocDefault));
  }
  const dim3 grid((unsigned)((B + 255) / 256));
  if (b->d.dtype == FS_F64)
    hipLaunchKernelGGL((gather_boundary_iterate<double>), grid, dim3(256), 0, b->stream, (const double *)b->hg, (const double *)b->Qg,
                       b->reach_nodes, b->ends_dev, B, N);
  else
    hipLaunchKernelGGL((gather_boundary_iterate<float>), grid, dim3(256), 0, b->stream, (const float *)b->hg, (const float *)b->Qg,
                       b->reach_nodes, b->ends_dev, B, N);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(b->ends_pin, b->ends_dev, 4 * B * sizeof(double), hipMemcpyDeviceToHost, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  std::memcpy(out, b->ends_pin, 4 * B * sizeof(double));
  return 0;
}

int fs_batch_restart(fs_batch *b, int32_t level, const double *h, const double *Q, const double *h_guess, const double *Q_guess,
                     const double *storage_stage) {
  if (!b || !h || !Q || !h_guess || !Q_guess) return fail("fs_batch_restart: null argument");
  if (level < 0 || level + 1 >= b->d.max_levels) return fail("fs_batch_restart: level out of range");
  if (level > 0 && !storage_stage && b->have_bc[FS_DOWNSTREAM] && b->any_storage[FS_DOWNSTREAM])      // (per-reach kinds: any reach)
    return fail("fs_batch_restart: a storage boundary continues from the reservoir stage of `level` (storage_stage[B], "
                "fs_batch_get_storage_stage); without it the run would go on from stage 0");
  if (fs_batch_set_state(b, h, Q)) return -1;
  FS_ON_DEVICE(b);
  const size_t B = b->d.n_reaches, N = b->d.n_nodes;
  if (upload(b, &b->hg, h_guess, B * N) || upload(b, &b->Qg, Q_guess, B * N)) return -1;
  if (storage_stage && upload(b, &b->Yprev, storage_stage, B)) return -1;
  if (level > 0) {     // the boundary row of `level` (fs_batch_set_state wrote it to row 0) moves to its own row
    HIP_TRY(hipMemcpyAsync((char *)b->hydro + (size_t)level * 4 * B * b->esz, b->hydro, 4 * B * b->esz, hipMemcpyDeviceToDevice, b->stream));
    if (b->hist_h) {
      HIP_TRY(hipMemcpyAsync((char *)b->hist_h + (size_t)level * B * N * b->esz, b->hist_h, B * N * b->esz, hipMemcpyDeviceToDevice, b->stream));
      HIP_TRY(hipMemcpyAsync((char *)b->hist_Q + (size_t)level * B * N * b->esz, b->hist_Q, B * N * b->esz, hipMemcpyDeviceToDevice, b->stream));
    }
    HIP_TRY(hipStreamSynchronize(b->stream));
  }
  b->level = level; b->restart_level = level;
  return 0;
}

int fs_batch_sync(fs_batch *b) {
  if (!b) return fail("null handle");
  FS_ON_DEVICE(b);
  TraceRange range_("flowsim:sync");
  HIP_TRY(hipStreamSynchronize(b->stream));
  return 0;
}

int32_t fs_batch_level(const fs_batch *b) { return b ? b->level : -1; }

int fs_batch_get_state(fs_batch *b, double *h, double *Q) {
  if (!b || !h || !Q) return fail("fs_batch_get_state: null argument");
  FS_ON_DEVICE(b);
  const size_t n = (size_t)b->d.n_reaches * b->d.n_nodes;
  return download(b, h, b->hk, 0, n) || download(b, Q, b->Qk, 0, n) ? -1 : 0;
}

int fs_batch_get_guess(fs_batch *b, double *h, double *Q) {
  if (!b || !h || !Q) return fail("fs_batch_get_guess: null argument");
  FS_ON_DEVICE(b);
  const size_t n = (size_t)b->d.n_reaches * b->d.n_nodes;
  return download(b, h, b->hg, 0, n) || download(b, Q, b->Qg, 0, n) ? -1 : 0;
}

int fs_batch_get_hydrographs(fs_batch *b, int32_t first, int32_t n, double *out) {
  if (!b || !out) return fail("fs_batch_get_hydrographs: null argument");
  FS_ON_DEVICE(b);
  if (first < 0 || n < 1 || first + n > b->d.max_levels) return fail("fs_batch_get_hydrographs: level range out of bounds");
  const size_t B = b->d.n_reaches;
  return download(b, out, b->hydro, (size_t)first * 4 * B, (size_t)n * 4 * B);
}

int fs_batch_get_iterations(fs_batch *b, int32_t first, int32_t n, int32_t *out) {
  if (!b || !out) return fail("fs_batch_get_iterations: null argument");
  FS_ON_DEVICE(b);
  if (first < 0 || n < 1 || first + n > b->d.max_levels) return fail("fs_batch_get_iterations: level range out of bounds");
  const size_t B = b->d.n_reaches;
  HIP_TRY(hipStreamSynchronize(b->stream));
  HIP_TRY(hipMemcpy(out, b->iters + (size_t)first * B, (size_t)n * B * 4, hipMemcpyDeviceToHost));
  return 0;
}

int fs_batch_get_status(fs_batch *b, int32_t *out) {
  if (!b || !out) return fail("fs_batch_get_status: null argument");
  FS_ON_DEVICE(b);
  HIP_TRY(hipStreamSynchronize(b->stream));
  HIP_TRY(hipMemcpy(out, b->status, (size_t)b->d.n_reaches * 4, hipMemcpyDeviceToHost));
  return 0;
}

int fs_batch_get_history(fs_batch *b, int32_t first, int32_t n, double *h, double *Q) {
  if (!b || !h || !Q) return fail("fs_batch_get_history: null argument");
  FS_ON_DEVICE(b);
  if (!b->hist_h) return fail("fs_batch_get_history: batch was created without FS_FLAG_HISTORY");
  if (first < 0 || n < 1 || first + n > b->d.max_levels) return fail("fs_batch_get_history: level range out of bounds");
  if (b->restart_level > 0 && first < b->restart_level)
    return fail("fs_batch_get_history: this batch was restarted at level " + std::to_string(b->restart_level) + "; the history before it was not restored");
  const size_t per = (size_t)b->d.n_reaches * b->d.n_nodes;
  return download(b, h, b->hist_h, first * per, n * per) || download(b, Q, b->hist_Q, first * per, n * per) ? -1 : 0;
}

int fs_batch_get_storage_stage(fs_batch *b, double *out) {
  if (!b || !out) return fail("fs_batch_get_storage_stage: null argument");
  FS_ON_DEVICE(b);
  return download(b, out, b->Yprev, 0, b->d.n_reaches);
}

int fs_batch_get_residual_trace(fs_batch *b, int32_t first, int32_t n, double *out) {
  if (!b || !out) return fail("fs_batch_get_residual_trace: null argument");
  FS_ON_DEVICE(b);
  if (!b->trace) return fail("fs_batch_get_residual_trace: batch was created without FS_FLAG_TRACE");
  if (first < 0 || n < 1 || first + n > b->d.max_levels) return fail("fs_batch_get_residual_trace: level range out of bounds");
  const size_t per = (size_t)FS_TRACE_CAP * b->d.n_reaches;
  return download(b, out, b->trace, first * per, n * per);
}

int fs_batch_get_storage_stages(fs_batch *b, int32_t first, int32_t n, double *out) {
  if (!b || !out) return fail("fs_batch_get_storage_stages: null argument");
  FS_ON_DEVICE(b);
  if (first < 0 || n < 1 || first + n > b->d.max_levels) return fail("fs_batch_get_storage_stages: level range out of bounds");
  const size_t B = b->d.n_reaches;
  return download(b, out, b->stage_hist, (size_t)first * B, (size_t)n * B);
}

int fs_batch_derive_device(fs_batch *b, int32_t first, int32_t n, int32_t fields) {
  if (!b) return fail("null handle");
  if (!b->hist_h) return fail("fs_batch_derive: batch was created without FS_FLAG_HISTORY");
  if (first < 0 || n < 1 || first + n > b->d.max_levels) return fail("fs_batch_derive: level range out of bounds");
  if ((fields & FS_DERIVE_ALL) == 0) return fail("fs_batch_derive: no field requested");
  if (b->restart_level > 0 && first < b->restart_level)
    return fail("fs_batch_derive: this batch was restarted at level " + std::to_string(b->restart_level) +
                "; the history before it was not restored (row 0 holds the restart state, which amplitudes then refer to)");
  FS_ON_DEVICE(b);
  TraceRange range_("flowsim:derive");
  const size_t BN = (size_t)b->d.n_reaches * b->d.n_nodes;
  void *dev[8] = {nullptr};
  for (int f = 0; f < 8; ++f) {
    if (!(fields & (1 << f))) continue;
    const size_t need = f == 7 ? BN : BN * n;
    if (b->derived_cap[f] < need) {          // grown, never shrunk: a later call of the same size allocates nothing
      if (b->derived[f]) { (void)hipFree(b->derived[f]); b->derived[f] = nullptr; b->derived_cap[f] = 0; }
      HIP_TRY(hipMalloc(&b->derived[f], need * b->esz));
      b->derived_cap[f] = need;
    }
    dev[f] = b->derived[f];
  }
  const size_t per_thread = 16 / b->esz;           // one 16-byte access per thread, level and field
  const dim3 grid((unsigned)(((BN + per_thread - 1) / per_thread + 255) / 256));
  HIP_TRY(hipEventRecord(b->ev0, b->stream));      // fs_batch_last_step_ms() then reports this kernel
  if (b->d.dtype == FS_F64) {
    fs::DeriveArgs<double> a{b->d.n_reaches, b->d.n_nodes, first, n, b->d.section_mode, (const double *)b->hist_h,
                             (const double *)b->hist_Q, (const double *)b->geo_uniform, (const double *)b->geo_table,
                             (const double *)b->poly_x, (const double *)b->poly_z, b->poly_n,
                             (int64_t)b->geo_reach_stride, (int64_t)b->poly_reach_stride,
                             (double *)dev[0], (double *)dev[1], (double *)dev[2], (double *)dev[3], (double *)dev[4],
                             (double *)dev[5], (double *)dev[6], (double *)dev[7], b->reach_nodes};
    hipLaunchKernelGGL((fs::derive_fields_kernel<double, 2>), grid, dim3(256), 0, b->stream, a);
  } else {
    fs::DeriveArgs<float> a{b->d.n_reaches, b->d.n_nodes, first, n, b->d.section_mode, (const float *)b->hist_h,
                            (const float *)b->hist_Q, (const float *)b->geo_uniform, (const float *)b->geo_table,
                            (const float *)b->poly_x, (const float *)b->poly_z, b->poly_n,
                            (int64_t)b->geo_reach_stride, (int64_t)b->poly_reach_stride,
                            (float *)dev[0], (float *)dev[1], (float *)dev[2], (float *)dev[3], (float *)dev[4],
                            (float *)dev[5], (float *)dev[6], (float *)dev[7], b->reach_nodes};
    hipLaunchKernelGGL((fs::derive_fields_kernel<float, 4>), grid, dim3(256), 0, b->stream, a);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(b->ev1, b->stream));
  b->timed = true; b->launches = 1;
  return 0;
}

void *fs_batch_derived_device_ptr(fs_batch *b, int32_t field_index) {
  return (b && field_index >= 0 && field_index < 8) ? b->derived[field_index] : nullptr;
}

int fs_batch_derive(fs_batch *b, int32_t first, int32_t n, double *level, double *area, double *top_width,
                    double *froude, double *velocity, double *celerity, double *amplitude, double *peak_amplitude) {
  if (!b) return fail("null handle");
  double *host[8] = {level, area, top_width, froude, velocity, celerity, amplitude, peak_amplitude};
  int fields = 0;
  for (int f = 0; f < 8; ++f) fields |= host[f] ? (1 << f) : 0;
  if (fs_batch_derive_device(b, first, n, fields)) return -1;
  FS_ON_DEVICE(b);
  const size_t BN = (size_t)b->d.n_reaches * b->d.n_nodes;
  for (int f = 0; f < 8; ++f)
    if (host[f] && download(b, host[f], b->derived[f], 0, f == 7 ? BN : BN * n)) return -1;
  return 0;
}

void *fs_batch_hydrograph_device_ptr(fs_batch *b) { return b ? b->hydro : nullptr; }
void *fs_batch_stream(fs_batch *b) { return b ? (void *)b->stream : nullptr; }

double fs_batch_last_step_ms(fs_batch *b) {
  if (!b || !b->timed) return -1.0;
  float ms = 0.f;
  if (hipEventSynchronize(b->ev1) != hipSuccess) return -1.0;
  if (hipEventElapsedTime(&ms, b->ev0, b->ev1) != hipSuccess) return -1.0;
  return (double)ms;
}

int32_t fs_batch_last_launch_count(fs_batch *b) { return b ? b->launches : 0; }

#ifdef FS_STAMP
// diagnostic builds only: out[B][16][12] cycle sums (waves beyond W are zero)
int fs_debug_stamps(fs_batch *b, unsigned long long *out) {
  if (!b || !out || !b->dbg) return fail("fs_debug_stamps: not available");
  HIP_TRY(hipStreamSynchronize(b->stream));
  HIP_TRY(hipMemcpy(out, b->dbg, (size_t)b->d.n_reaches * 16 * 12 * 8, hipMemcpyDeviceToHost));
  return 0;
}
#endif

int32_t fs_kernel_table_size(void) { return kNumEntries; }

int fs_kernel_table_entry(int32_t i, int32_t *out) {
  if (i < 0 || i >= kNumEntries || !out) return fail("fs_kernel_table_entry: index out of range");
  const Entry &e = kEntries[i];
  out[0] = e.dtype; out[1] = e.sec; out[2] = e.M; out[3] = e.W; out[4] = e.full; out[5] = e.bck; out[6] = e.diag; out[7] = e.longk;
  return 0;
}

int32_t fs_batch_kernel_index(fs_batch *b) { return (b && b->kern) ? (int32_t)(b->kern - kEntries) : -1; }

int32_t fs_kernel_table_entry_tail(int32_t i) { return (i >= 0 && i < kNumEntries) ? kEntries[i].tail : -2; }
int32_t fs_kernel_table_entry_team(int32_t i) { return (i >= 0 && i < kNumEntries) ? kEntries[i].team : -2; }

int32_t fs_batch_poly_tables(fs_batch *b) { return (b && b->poly_x) ? (b->poly_K > 0 ? 1 : 0) : -1; }

int fs_batch_kernel_info(fs_batch *b, int32_t *cells_per_thread, int32_t *waves_per_reach, int32_t *lds_bytes,
                         int32_t *vgprs) {
  if (!b) return fail("null handle");
  hipFuncAttributes at;
  HIP_TRY(hipFuncGetAttributes(&at, b->kern->kp));
  if (cells_per_thread) *cells_per_thread = b->kern->M;
  if (waves_per_reach) *waves_per_reach = b->kern->W;
  if (lds_bytes) *lds_bytes = (int32_t)at.sharedSizeBytes;
  if (vgprs) *vgprs = at.numRegs;
  return 0;
}

}  // extern "C"
