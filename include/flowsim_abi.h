/*
 * flowsim_abi.h - C ABI of the MI355X-native batched Preissmann stepper (libflowsim_hip.so).
 *
 * Drop-in boundary for ONE path of cve-mohd/flow-sim: the per-timestep Newton loop of
 * PreissmannSolver.run (reference: src/hydromodel/preissmann.py:101-163) together with everything
 * it calls per node (preissmann.py:61-99, :200-344, :407-798, :899-910; solver.py:244-296;
 * channel.py:53-105,:172-190; cross_section.py:114-175,:623-793; hydraulics.py:4-229;
 * boundary.py:56-247; rating_curve.py:32-63,:132-147; lumped_storage.py:24-45) and the sparse
 * solve it hands to scipy.sparse.linalg.spsolve (preissmann.py:146).
 *
 * The reference is pure Python and has no FFI of its own; the binding a maintainer would add is a
 * ctypes stub inside PreissmannSolver.run (shown in INTEGRATION.md).  Every entry point below
 * states which reference interface it stands in for.
 *
 * Conventions
 *   - plain C, no torch / numpy types; all host arrays are caller-owned, contiguous, float64
 *     (also for FS_F32 batches - converted on upload) unless stated otherwise;
 *   - B = reaches in the batch (independent channels / ensemble members), N = nodes per reach,
 *     level k = time level (k = 0 is the initial condition);
 *   - return 0 on success, <0 on error (text via fs_last_error(), thread-local);
 *   - a handle owns its device buffers and one HIP stream; it is not thread-safe;
 *   - calls are asynchronous on the handle's stream unless they copy to host memory.
 */
#ifndef FLOWSIM_ABI_H
#define FLOWSIM_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FS_ABI_VERSION 3

typedef struct fs_batch fs_batch;

/* arithmetic type of the path (reference: float64 everywhere, solver.py:43-44).  FS_F32 is a throughput mode for large
 * ensembles of short reaches (BASELINE configs[4]): it holds 5e-4 of the fp64 answer at tolerance 1e-3, cannot resolve
 * ||R|| below ~6e-8 |Q| sqrt(2N) (4 096-node reaches need a tolerance of ~2e-2) and carries no 1e-8 parity claim. */
enum { FS_F64 = 0, FS_F32 = 1 };

/* How node geometry reaches the kernel.
 *   RECT_UNIFORM : one rectangular prismatic channel per reach from per-reach scalars
 *                  (what Channel(width=, roughness=) builds, channel.py:282-294);
 *   TRAP_UNIFORM : one simple (non-compound) trapezoidal prismatic channel per reach from
 *                  per-reach scalars (two TrapezoidalSection(z_bank=None) end sections with equal
 *                  b_main / m_main / n_main, cross_section.py:641-645; BASELINE configs[4]);
 *   TABLE        : one [FS_GEO_NPARAM][N] table of TrapezoidalSection parameters at the nodes
 *                  (cross_section.py:569-613 after interpolation, channel.py:213-241), shared by
 *                  all reaches, optional per-reach main-channel Manning n
 *                  (cases/gerd_roseires/custom_functions.py:147).
 *   IRREGULAR    : TABLE plus polyline nodes: what Channel.xs_at_node holds when any input
 *                  section is an IrregularSection (cross_section.py:207-543; mixed interpolation
 *                  :932-968).  fp64 only (the reference's 1e-6 finite differences do not survive
 *                  single precision). */
enum { FS_SEC_RECT_UNIFORM = 0, FS_SEC_TRAP_UNIFORM = 1, FS_SEC_TABLE = 2, FS_SEC_IRREGULAR = 3 };

/* rows of the RECT_UNIFORM parameter block, each [B] */
enum { FS_RU_WIDTH = 0, FS_RU_MANNING = 1, FS_RU_Z_US = 2, FS_RU_Z_DS = 3, FS_RU_NPARAM = 4 };
/* rows of the TRAP_UNIFORM parameter block, each [B]: the four above + the side slope m (H:V) */
enum { FS_TU_SIDE_SLOPE = 4, FS_TU_NPARAM = 5 };

/* rows of the TABLE geometry block, each [N] (TrapezoidalSection attributes) */
enum {
  FS_GEO_Z_BED = 0, FS_GEO_B_MAIN, FS_GEO_M_MAIN, FS_GEO_N_MAIN, FS_GEO_N_LEFT, FS_GEO_N_RIGHT,
  FS_GEO_IS_COMPOUND, FS_GEO_H_BANKFULL, FS_GEO_B_FP_LEFT, FS_GEO_B_FP_RIGHT, FS_GEO_M_FP,
  FS_GEO_CURVATURE, FS_GEO_NPARAM
};

/* Boundary.condition (boundary.py:32) flattened; rating_curve split by RatingCurve.type
 * (rating_curve.py:10-31) plus the smooth Roseires gate curve of cases/gerd_roseires
 * (roseires_rating_curve.py:65-109,:202-208) and fixed_depth behind a constant-area
 * LumpedStorage (boundary.py:97-133, lumped_storage.py:24-45). */
enum {
  FS_BC_FLOW_HYDROGRAPH = 0,  /* params: -                      ; needs target table          */
  FS_BC_STAGE_HYDROGRAPH = 1, /* params: bed_level              ; needs target table          */
  FS_BC_FIXED_DEPTH = 2,      /* params: initial_depth                                        */
  FS_BC_NORMAL_DEPTH = 3,     /* params: bed_slope, bed_level                                 */
  FS_BC_RATING_POWER = 4,     /* params: a, b, stage_shift, bed_level                         */
  FS_BC_RATING_POLY = 5,      /* params: a, b, c, stage_shift, bed_level                      */
  FS_BC_RATING_BLEND = 6,     /* params: stage0, buffer, lo0, lo1, lo2, hi0, hi1, hi2, dY, bed_level
                                 Q = (1-s)*lo(z) + s*hi(z), lo/hi quadratics in z, s = smoothstep */
  FS_BC_STORAGE = 7,          /* params: surface_area, min_stage, Y_min, Y_max, bed_level (downstream only) */
  FS_BC_STORAGE_CURVE = 8,    /* general LumpedStorage (lumped_storage.py:8-179; downstream only): area curve,
                                 reservoir rating curve, entrance losses.  params[FS_SC_NFIXED + 2*n_curve]:
                                 the FS_SC_* scalars, then stage[n_curve], area[n_curve] of set_area_curve
                                 (n_curve = 0: constant surface_area).  The mass-balance root
                                 (lumped_storage.py:24-35) is found on the device with the algorithm of
                                 scipy.optimize.brentq and its default tolerances.  Section modes
                                 FS_SEC_TABLE and FS_SEC_IRREGULAR only (fs_batch_step refuses it in the
                                 uniform-geometry modes: describe such a channel as a table).  Shared by the
                                 batch (per_reach = 0) or one reservoir per reach (per_reach = 1:
                                 params[FS_SC_NFIXED + 2*n_curve][B], the same n_curve for every reach). */
  FS_BC_HOST_ROW = 9          /* a boundary whose plugin has no device form: any object with discharge(stage, time) /
                                 dQ_dz(stage, time) (rating_curve.py:32-63,:132-147; e.g. RoseiresRatingCurve(smooth=False),
                                 cases/gerd_roseires/roseires_rating_curve.py:65-140, whose gates move with time) or a
                                 LumpedStorage with a callable rating curve (lumped_storage.py:24-35).  params[3][B]
                                 (per_reach = 1) = (d/dh, d/dQ, residual) of the boundary equation at the current Newton
                                 vector, evaluated by the caller before every iteration (boundary.py:56-242) and handed over
                                 with fs_batch_set_host_rows; such a batch advances with fs_batch_iterate (one Newton
                                 iteration per launch), not fs_batch_step.  Section modes FS_SEC_TABLE / FS_SEC_IRREGULAR. */
};
/* scalar slots of FS_BC_STORAGE_CURVE */
enum {
  FS_SC_MIN_STAGE = 0, FS_SC_Y_MIN, FS_SC_Y_MAX, FS_SC_BED_LEVEL, FS_SC_SURFACE_AREA, FS_SC_ALPHA, FS_SC_BETA,
  FS_SC_N_CURVE,
  FS_SC_RC_TYPE,            /* 0: no outflow, 1: power a*(Y+shift)^b, 2: polynomial a*x^2 + b*x + c (rating_curve.py:10-63) */
  FS_SC_RC_A, FS_SC_RC_B, FS_SC_RC_C, FS_SC_RC_SHIFT,
  FS_SC_CAPTURE_LOSSES,     /* LumpedStorage.capture_losses (0/1) */
  FS_SC_RESERVOIR_LENGTH, FS_SC_K_Q,
  FS_SC_NFIXED
};
#define FS_BC_MAX_PARAMS 10   /* of the fixed-size kinds 0..7 */
enum { FS_UPSTREAM = 0, FS_DOWNSTREAM = 1 };

/* per-reach status after stepping (preissmann.py:124-126 raises ValueError; :135-137 NaN check) */
enum { FS_OK = 0, FS_MAX_ITER = 1, FS_NAN = 2, FS_STORAGE_RANGE = 3,
       /* A warning, not a failure (the reach keeps stepping and the status sticks): the linear systems of this reach are
        * ill-conditioned - in practice supercritical flow (v > c) over a long stretch, where one boundary condition per end
        * is not what the flow takes.  The on-chip elimination does not pivot; it watches how strongly the first unknown of any
        * segment of rows depends on the last one (a product of super-diagonal / pivot ratios that decays for subcritical
        * flow) and raises this status when that exceeds 2^10 in the iteration that accepts a level: results may then differ from another solver's (the reference's
        * SuperLU included) by more than the 1e-8 this library otherwise keeps.  The reference's counterpart is the
        * `diagnos` check of run(), ValueError("Jacobian is ill-conditioned (rcond too small)"), preissmann.py:139-144, and
        * the Froude diagnosis of check_criticality (:179-198).  Raised by the kernels compiled with diagnostics: every batch
        * with FS_FLAG_HISTORY, FS_FLAG_TRACE or FS_FLAG_MONITOR, and every batch no specialised kernel exists for. */
       FS_ILL_CONDITIONED = 4,
       /* A reach longer than one lane grid is advanced by a team of workgroups that meet once per Newton iteration (uniform section
        * modes beyond 4 096 nodes).  A member that waits for the others longer than about eight seconds gives the reach up with this
        * status instead of spinning on; it has not been seen to happen (a team's members are the workgroups that started first, so
        * none of them waits for one that cannot start) and would mean a lost or wedged workgroup.  The reach's state is that of the
        * last launch that completed. */
       FS_TEAM_STALL = 5 };

enum {
  FS_FLAG_HISTORY = 1,  /* keep depth/flow[level][B][N] on the device (solver.py:43-44); large batches that only need the
                           boundary hydrographs leave it (and FS_FLAG_TRACE) off: no [levels][B][N] arrays in HBM and
                           step kernels compiled without those stores */
  FS_FLAG_TRACE = 2,    /* keep ||R|| of every Newton iteration (what run(verbose=3) prints, preissmann.py:149-152) */
  FS_FLAG_MONITOR = 4   /* run the conditioning monitor (FS_ILL_CONDITIONED) also on a batch that keeps neither history nor trace:
                           selects the kernels compiled with diagnostics (measured: 1.2 % slower on the 4 096-node benchmark shape, 3 % on
                           the 121-node ensemble).  The Python binding sets it by default (PreissmannBatch(monitor=True), the reference's
                           per-run `diagnos` check stands behind it, preissmann.py:133-144); a caller that wants the fastest kernels and
                           knows its flow to be subcritical passes monitor=False, as bench.py does. */
};
#define FS_TRACE_CAP 64 /* iterations per level kept by FS_FLAG_TRACE */

typedef struct fs_batch_desc {
  int32_t n_reaches;     /* B */
  int32_t n_nodes;       /* N >= 2; up to 4 096 (uniform section modes) / 2 048 (tables, polylines) a reach stays on chip, longer ones -
                            to 32 768 / 16 384 nodes - run on the multi-pass kernel (state and level constants through L2) */
  int32_t dtype;         /* FS_F64 | FS_F32 */
  int32_t section_mode;  /* FS_SEC_* */
  int32_t device;        /* HIP device ordinal */
  int32_t max_levels;    /* capacity of per-level tables: time levels 0..max_levels-1 (solver.py:35) */
  int32_t flags;         /* FS_FLAG_* */
  int32_t reserved;
} fs_batch_desc;

int fs_abi_version(void);
int fs_device_count(void);                 /* 0 when no HIP device is visible */
const char *fs_last_error(void);

/* Solver.__init__ state arrays + PreissmannSolver.__init__ (solver.py:11-51, preissmann.py:23-46) */
fs_batch *fs_batch_create(const fs_batch_desc *desc);
void fs_batch_destroy(fs_batch *b);

/* theta / time_step / spatial_step of the solver and tolerance / max_iter of run()
 * (preissmann.py:23-46, :101; solver.py:32, :53-55) - shared by the whole batch */
int fs_batch_set_scheme(fs_batch *b, double theta, double dt, double dx, double tolerance, int32_t max_iter);

/* RECT_UNIFORM: params[FS_RU_NPARAM][B];  TRAP_UNIFORM: params[FS_TU_NPARAM][B] */
int fs_batch_set_geometry_uniform(fs_batch *b, const double *params);
/* TABLE: table[FS_GEO_NPARAM][N]; n_main_override[B] or NULL */
int fs_batch_set_geometry_table(fs_batch *b, const double *table, const double *n_main_override);
/* IRREGULAR: table[FS_GEO_NPARAM][N] as for TABLE.  Node i with n_pts[i] > 0 is the polyline
 * x[i][0..n_pts[i]), z[i][..] (rows of length max_pts, x ascending, >= 2 points; IrregularSection.x/.z
 * after its argsort, cross_section.py:231-233) with roughness strips split at limits[i][0..1]
 * (left_fp_limit, right_fp_limit, +-inf allowed; cross_section.py:105-111); of its table column only
 * Z_BED (= min z), N_MAIN, N_LEFT, N_RIGHT and CURVATURE are read.  n_pts[i] == 0: trapezoid-family
 * node described by the table column.  n_main_override[B] or NULL. */
int fs_batch_set_geometry_irregular(fs_batch *b, const double *table, const int32_t *n_pts, int32_t max_pts,
                                    const double *x, const double *z, const double *limits,
                                    const double *n_main_override);

/* Heterogeneous batches: every reach its own channel - what the reference builds per Channel / Solver object
 * (channel.py:213-241 node sections, cross_section.py:857-930 interpolation, solver.py:34-38,53-55 grid) - so that different
 * rivers, or a geometry Monte-Carlo of one river (bed levels, widths, bankfull depths), step in one launch.
 *   fs_batch_set_geometry_table_per_reach      tables[B][FS_GEO_NPARAM][N]: one TrapezoidalSection table per reach (FS_SEC_TABLE);
 *   fs_batch_set_geometry_irregular_per_reach  the same for FS_SEC_IRREGULAR: tables[B][FS_GEO_NPARAM][N], n_pts[B][N],
 *                                              x / z [B][N][max_pts], limits[B][N][2];
 *   fs_batch_set_reach_nodes                   n_nodes[B], 2 <= n_nodes[r] <= N of the batch (NULL: all N): reach r uses the
 *                                              first n_nodes[r] entries of its rows of every [B][N] array (state, tables - pad the
 *                                              rest with the last node's values -, history); entries beyond are left untouched;
 *   fs_batch_set_reach_scheme                  theta[B], dt[B], dx[B] (any may be NULL: the batch-wide value of
 *                                              fs_batch_set_scheme, which is called first);
 *   fs_batch_set_reach_tolerance               tolerance[B], max_iter[B] (either may be NULL: the batch-wide value): every run() of the reference
 *                                              has its own tolerance and iteration cap (preissmann.py:101);
 *   fs_batch_set_bc_per_reach                  kinds[B] (FS_BC_FLOW_HYDROGRAPH .. FS_BC_STORAGE, and FS_BC_HOST_ROW on the reaches whose
 *                                              plugin has no device form), params[FS_BC_MAX_PARAMS][B] (row i = parameter i of the
 *                                              reach's own kind, unused rows ignored), target[max_levels][B];
 *   fs_batch_set_bc_per_reach_wide             the same with params[n_params][B], n_params >= FS_BC_MAX_PARAMS: room for reaches of kind
 *                                              FS_BC_STORAGE_CURVE (a general reservoir behind some channels of the batch, each with
 *                                              its FS_SC_* rows and an area curve of its own length: FS_SC_NFIXED + 2*n_curve <= n_params).
 * Time level k of reach r is t = k * dt[r]: targets are sampled per reach.  These batches run on the general kernels (run-time
 * boundary switch, ragged node counts).  One FS_BC_HOST_ROW reach makes the batch one that advances with fs_batch_iterate (TABLE /
 * IRREGULAR section modes): the reference runs a RatingCurve subclass on one channel next to closed-form boundaries on the others
 * (boundary.py:56-141 is per Boundary object), and so does a batch. */
int fs_batch_set_geometry_table_per_reach(fs_batch *b, const double *tables, const double *n_main_override);
int fs_batch_set_geometry_irregular_per_reach(fs_batch *b, const double *tables, const int32_t *n_pts, int32_t max_pts,
                                              const double *x, const double *z, const double *limits,
                                              const double *n_main_override);
int fs_batch_set_reach_nodes(fs_batch *b, const int32_t *n_nodes);
int fs_batch_set_reach_scheme(fs_batch *b, const double *theta, const double *dt, const double *dx);
int fs_batch_set_reach_tolerance(fs_batch *b, const double *tolerance, const int32_t *max_iter);
int fs_batch_set_bc_per_reach(fs_batch *b, int32_t side, const int32_t *kinds, const double *params, const double *target);
int fs_batch_set_bc_per_reach_wide(fs_batch *b, int32_t side, const int32_t *kinds, const double *params, int32_t n_params,
                                   const double *target);

/* one boundary (Boundary.__init__, boundary.py:10-46).  params[n_params] when per_reach == 0,
 * params[n_params][B] otherwise.  target[max_levels][B] (value at t = level*dt, i.e.
 * Hydrograph.get_at pre-sampled, hydrograph.py:15-22) or NULL. */
int fs_batch_set_bc(fs_batch *b, int32_t side, int32_t kind, const double *params, int32_t n_params,
                    int32_t per_reach, const double *target);

/* Channel.initial_conditions (channel.py:107-138) -> depth[0], flow[0] and the Newton start
 * vector (solver.py:61-63, preissmann.py:48-59).  h, Q: [B][N].  Resets the level counter. */
int fs_batch_set_state(fs_batch *b, const double *h, const double *Q);

/* Same for the 'steady-state' initial condition of a prismatic reach (channel.py:296-305: one
 * normal depth and the initial flow at every node): h[B], Q[B], broadcast along the reach on
 * the device (avoids staging B*N host values for large batches). */
int fs_batch_set_state_uniform(fs_batch *b, const double *h, const double *Q);

/* THE HOT PATH: advances every reach by n_steps time levels (preissmann.py:108-161).  Results
 * stay on the device: boundary hydrographs, Newton counts, status, (history).  Asynchronous. */
int fs_batch_step(fs_batch *b, int32_t n_steps);
int fs_batch_sync(fs_batch *b);
int32_t fs_batch_level(const fs_batch *b);   /* current time level k */

/* The same loop opened up for boundaries evaluated on the host (FS_BC_HOST_ROW): ONE Newton iteration
 * (preissmann.py:122-156: residual, Jacobian, solve, update, norm test) of level fs_batch_level() + 1 for every reach
 * that has not converged on it yet.  *n_open (may be NULL) returns how many reaches still iterate; when it reaches 0
 * the level is complete for the whole batch and fs_batch_level() has advanced.  Synchronous.  Works for any TABLE /
 * IRREGULAR batch (device-evaluated kinds included). */
int fs_batch_iterate(fs_batch *b, int32_t *n_open);
/* rows[3][B] = (d/dh, d/dQ, residual) of the side's boundary equation at the current Newton vector
 * (Boundary.df_dh, df_dQ, condition_residual: boundary.py:143-242, :56-141).  On a side with per-reach kinds the entries of the
 * reaches whose kind the device evaluates are ignored (their parameters stay). */
int fs_batch_set_host_rows(fs_batch *b, int32_t side, const double *rows);
/* the current Newton vector at the two boundary nodes, out[4][B] = h[0], Q[0], h[N-1], Q[N-1]
 * (what preissmann.py:200-218, :303-320 pass to the boundary: depth_at / flow_at of the iterate) */
int fs_batch_get_boundary_iterate(fs_batch *b, double *out);

/* Restart (the reference keeps its whole history in memory and has no restart, solver.py:43-44):
 * everything a run needs to continue bit-exactly from time level `level`: depth/flow[level] (h, Q: what
 * fs_batch_get_state returned), the Newton start vector of level+1 (h_guess, Q_guess: fs_batch_get_guess; after the
 * first level it differs from the state) and, behind a storage boundary, the reservoir stage of `level`
 * (storage_stage[B]: fs_batch_get_storage_stage; required behind a storage boundary when level > 0, NULL otherwise).  All
 * [B][N] float64.  A batch that keeps a history (FS_FLAG_HISTORY) holds, after a restart, the rows from `level` on:
 * fs_batch_get_history / fs_batch_derive refuse ranges that begin before it, and the amplitude fields of fs_batch_derive
 * (solver.py:96-97: depth - depth[0]) then refer to the restart state.  The per-reach status starts from FS_OK again. */
int fs_batch_restart(fs_batch *b, int32_t level, const double *h, const double *Q, const double *h_guess,
                     const double *Q_guess, const double *storage_stage);

/* depth[k], flow[k] of the current level (what update_guesses stored, preissmann.py:166-177).
 * The kernel writes the accepted state to HBM at the last level of each fs_batch_step call (and at
 * every level into the history when FS_FLAG_HISTORY is set); for a reach whose status is not FS_OK
 * this is therefore the state at the end of the previous call. */
int fs_batch_get_state(fs_batch *b, double *h, double *Q);
/* the post-update Newton vector that seeds the next level (preissmann.py:146-147) */
int fs_batch_get_guess(fs_batch *b, double *h, double *Q);
/* out[n_levels][4][B] = depth[k,0], flow[k,0], depth[k,-1], flow[k,-1] for k = first..first+n-1
 * (what cases read from solver.depth / solver.flow, cases/gerd_roseires/model.py:107-111) */
int fs_batch_get_hydrographs(fs_batch *b, int32_t first_level, int32_t n_levels, double *out);
int fs_batch_get_iterations(fs_batch *b, int32_t first_level, int32_t n_levels, int32_t *out); /* [n][B] */
int fs_batch_get_status(fs_batch *b, int32_t *out);                                            /* [B]   */
/* FS_FLAG_HISTORY only: h, Q [n_levels][B][N] (solver.depth / solver.flow, solver.py:43-44) */
int fs_batch_get_history(fs_batch *b, int32_t first_level, int32_t n_levels, double *h, double *Q);
/* FS_FLAG_TRACE only: out[n_levels][FS_TRACE_CAP][B] residual norms, iteration i of level k at
 * [k][i-1][reach]; entries beyond the iteration count of a level are 0 */
int fs_batch_get_residual_trace(fs_batch *b, int32_t first_level, int32_t n_levels, double *out);
/* reservoir stage kept per level by the storage boundary (boundary.py:126-131): out[B] */
int fs_batch_get_storage_stage(fs_batch *b, double *out);
/* the same per time level (LumpedStorage.stage_hydrograph, boundary.py:126-131): out[n_levels][B],
 * rows of levels that have not been computed (and level 0) are 0 */
int fs_batch_get_storage_stages(fs_batch *b, int32_t first_level, int32_t n_levels, double *out);

/* Solver.prepare_results (solver.py:65-127) for levels [first, first+n) of the stored history
 * (FS_FLAG_HISTORY): level = depth + bed, area, top width, Froude number (hydraulics.py:155-168),
 * velocity Q/A, wave celerity V + sqrt(g A / T), amplitude = depth - depth[0]; each [n][B][N], and
 * peak_amplitude [B][N] = max over those levels.  Any output pointer may be NULL.  One elementwise,
 * HBM-bound kernel; results are copied to the caller's host arrays (the device buffers behind them are
 * kept by the handle and reused by later calls). */
int fs_batch_derive(fs_batch *b, int32_t first_level, int32_t n_levels, double *level, double *area,
                    double *top_width, double *froude, double *velocity, double *celerity,
                    double *amplitude, double *peak_amplitude);
/* The same kernel with the results left on the device (no PCIe traffic): fields = bit mask of FS_DERIVE_* to
 * compute; fs_batch_derived_device_ptr(field index 0..7) then points at [n_levels][B][N] (index 7,
 * peak amplitude: [B][N]) in the batch dtype, valid until the next derive call on this handle. */
enum { FS_DERIVE_LEVEL = 1, FS_DERIVE_AREA = 2, FS_DERIVE_TOP_WIDTH = 4, FS_DERIVE_FROUDE = 8, FS_DERIVE_VELOCITY = 16,
       FS_DERIVE_CELERITY = 32, FS_DERIVE_AMPLITUDE = 64, FS_DERIVE_PEAK_AMPLITUDE = 128, FS_DERIVE_ALL = 255 };
int fs_batch_derive_device(fs_batch *b, int32_t first_level, int32_t n_levels, int32_t fields);
void *fs_batch_derived_device_ptr(fs_batch *b, int32_t field_index);

/* zero-copy access for device-side consumers (RCCL gather of hydrographs): device pointer to the
 * [max_levels][4][B] hydrograph block in the batch dtype, and the handle's hipStream_t */
void *fs_batch_hydrograph_device_ptr(fs_batch *b);
void *fs_batch_stream(fs_batch *b);

/* measurement: HIP-event time of the kernels launched by the last fs_batch_step (ms, after
 * fs_batch_sync), their count, and the launch geometry chosen for this batch */
double fs_batch_last_step_ms(fs_batch *b);
int32_t fs_batch_last_launch_count(fs_batch *b);
int fs_batch_kernel_info(fs_batch *b, int32_t *cells_per_thread, int32_t *waves_per_reach,
                         int32_t *lds_bytes, int32_t *vgprs);
/* The dispatch table of the step kernel (what fs_batch_step chooses from): fs_kernel_table_size() entries,
 * entry i described by out[8] = dtype, section_mode, cells per lane M, waves per reach W, full (1: only
 * N == 64*W*M: the downstream boundary row takes the last row of the lane grid), boundary class (-1 any kind, 0 any but FS_BC_STORAGE_CURVE / FS_BC_HOST_ROW,
 * 1 closed-form rectangular rows, 2+k flow hydrograph upstream and kind k downstream), diag (0: compiled
 * without history / trace stores), long (1: the multi-pass kernel for reaches longer than one lane grid: capacity 64*M rows per
 * wave slot x 64 slots).  The environment
 * variable FS_KERNEL_INDEX=i makes fs_batch_step use entry i or fail if it does not fit the batch (tests:
 * every instantiation is checked against the oracle). fs_batch_kernel_index: the entry the last step used. */
int32_t fs_kernel_table_size(void);
int fs_kernel_table_entry(int32_t i, int32_t *out);
int32_t fs_batch_kernel_index(fs_batch *b);
/* entry i in its tail-only form (round 4: ragged, but compiled for the local row the downstream boundary row takes in its lane):
 * that row, (N - 1) mod M, or -1 for every other entry (-2: no such entry).  fs_batch_step only picks such an entry for a batch
 * whose one node count gives that row. */
int32_t fs_kernel_table_entry_tail(int32_t i);
/* 1: entry i advances a reach LONGER than one lane grid as a team of ceil(N / (64 W M)) workgroups that each keep 64 W M rows on chip and
 * meet once per Newton iteration through device memory (uniform section modes, 4 097 ... 32 768 nodes; round 4) - fs_batch_step prefers it
 * to the multi-pass entries (long = 1), which remain for tables, polylines, host rows and FS_NO_TEAM=1; 0 otherwise (-2: no such entry). */
int32_t fs_kernel_table_entry_team(int32_t i);
/* FS_SEC_IRREGULAR: 1 when the batch evaluates its polylines from stage tables, 0 when it walks their edges (the same results at
 * about five times the instructions): the tables take about 10 KB per node at 40 stations, per channel, and
 * fs_batch_set_geometry_irregular(_per_reach) leaves them out beyond FS_POLY_TABLE_MAX_BYTES (environment, default 8 GiB) or with
 * FS_POLY_WALK=1.  The reference evaluates every call by walking the polyline (cross_section.py:248-538). -1: no polylines set. */
int32_t fs_batch_poly_tables(fs_batch *b);

#ifdef __cplusplus
}
#endif
#endif /* FLOWSIM_ABI_H */
