/* tinyopt_amd.h — C-ABI of the MI355X-native Levenberg-Marquardt / Gauss-Newton hot path.
 *
 * This is the drop-in boundary beneath tinyopt's `Optimize(x, cost, options)` /
 * Accumulate-callback API (reference: include/tinyopt/optimize.h:16-77,
 * include/tinyopt/optimizers/optimizer.h:145-539).  The reference has NO FFI layer (it is a
 * header-only template library), so every entry point below cites the C++ interface it
 * replaces.  Plain pointers and sizes only; no torch / Eigen / HIP types in any signature.
 *
 * Conventions
 *  - Every `*_dev` pointer is a DEVICE pointer valid on the handle's GPU; `*_host` is host memory.
 *  - All functions return 0 on success, <0 on error (see TOA_E_*); `toa_last_error()` describes it.
 *    Numeric outcomes are never errors: they are per-problem `StopReason` integers with the SAME
 *    values as the reference enum (include/tinyopt/stop_reasons.h:14-43).
 *  - A handle is NOT thread-safe: one handle per host thread per GPU (reference: one stateful
 *    `Optimizer_` per thread, optimizers/optimizer.h:544-548).  All kernels are launched on the
 *    `stream` given at creation (a hipStream_t cast to void*, NULL = default stream) and are
 *    asynchronous w.r.t. the host unless stated.
 *  - Problems are independent; a "batch" is P problems of identical (n, m, dtype, model).
 */
#ifndef TINYOPT_AMD_H_
#define TINYOPT_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TOA_VERSION 1

/* error codes */
#define TOA_OK 0
#define TOA_E_ARG (-1)     /* invalid argument (reference: std::invalid_argument, optimize.h:47,55,75) */
#define TOA_E_HIP (-2)     /* HIP runtime error */
#define TOA_E_NOMEM (-3)   /* device allocation failed.  toa_lm_run itself mirrors the reference (bad_alloc -> kOutOfMemory,
                              optimizer.h:75-86): a workspace that cannot be allocated ends every problem with
                              TOA_STOP_OUT_OF_MEMORY, x untouched, and the call returns TOA_OK */
#define TOA_E_UNSUPPORTED (-4)
#define TOA_E_RCCL (-5)    /* RCCL (collective) error */

/* scalar type of x / H / g (the solver `Scalar`, solvers/lm.h:27) */
#define TOA_F32 0
#define TOA_F64 1

/* residual-model families compiled into the library (device functors).  Arbitrary host lambdas
 * cannot run on the GPU; see INTEGRATION.md for the "bring your own functor" template entry. */
#define TOA_MODEL_DENSE_ROW 1       /* r_i = a_i.x + 0.1 sin(a_i.x) - b_i ; J_i = (1+0.1cos(a_i.x)) a_i    */
#define TOA_MODEL_GAUSSIAN_PRIOR 2  /* r = (x-y)/sigma, m = n  (benchmarks/dense.cpp:53-66, losses/mahalanobis.h:124-136) */
#define TOA_MODEL_SQRT2 3           /* r = x*x - 2, n = m = 1   (tests/sqrt2.cpp:30-70) */
#define TOA_MODEL_SE3_REPROJ 4      /* pinhole reprojection of 3-D points, SE3 right-perturbation */
/* residual functors differentiated ON THE DEVICE with forward-mode dual numbers (csrc/jet.hpp = ceres::Jet on the GPU;
 * replaces OptimizeWithAutoDiff, include/tinyopt/diff/optimize_autodiff.h:21-169) */
#define TOA_MODEL_CIRCLE_FIT 5      /* r_i = ||p_i - c||^2 - radius^2, x = (cx, cy, radius)   (tests/circle.cpp:32-68) */
#define TOA_MODEL_DENSE_ROW_AD6 6   /* the DenseRow residual, n = 6, written without a hand-derived Jacobian */
/* the analytic functions of the reference's optimizer tests as manual Accumulate callbacks with exact Hessians
 * (tests/optimize_easy.cpp:35-221, tests/optimize_hard.cpp:34-102): data_dev = [1] function id
 * (0 Rosenbrock, 1 plateau, 2 Powell [n = 4], 3 Beale, 4 Himmelblau, 5 `x - 2` [n = 1, tests/basic.cpp:41-87]);
 * m = residual count reported in Cost (1, 1, 1, 3, 2, 1); x: [P][n] = a batch of starting points */
#define TOA_MODEL_TESTFN 7
/* Gaussian prior with a general covariance: res = U (x - y), U = upper Cholesky factor of the information matrix
 * (losses/mahalanobis.h:160-171 MahaWhitenedInfoU, tests/cov.cpp:91-146); m == n; data_dev: [P][n + n*n] = y, U row-major */
#define TOA_MODEL_MAHA_PRIOR 8
/* SE3 pose prior, the reference's manifold test (tests/sophus.cpp:26-44): residual = log(prior_inv * x) in R^6,
 * differentiated on the device by dual numbers over the right perturbation; n == m == 6; x: [P][12] poses;
 * data_dev: [P][12] = prior_inv (rotation matrix row-major, translation) */
#define TOA_MODEL_SE3_PRIOR 9
/* DenseRow beyond one wavefront (n up to 4096 — beyond 1024 every stage of a pass is the library's): rows in natural layout.  64 <= n <= 128: one persistent
 * workgroup-per-problem kernel (csrc/large_fused.hip).  Beyond: J^T J by a hand-written LDS-staged MFMA Gram (fp32; the
 * library GEMM for fp64), the solve by a one-workgroup blocked Cholesky (fp32 n <= 1024, fp64 n <= 512; rocSOLVER's batched
 * potrf + potrs beyond and for use_ldlt = 0), the LM state machine in small kernels between them (csrc/large_n.hip).
 * toa_lm_run only. */
#define TOA_MODEL_DENSE_ROW_NATURAL 10

/* The DenseRow residual written as r(x) ONLY and differentiated on the device for wide parameter blocks (13 <= n <= 63,
 * "chunked Jets": the 16 lanes of a row group evaluate the functor on Jet chunks seeded on the columns each feeds to the
 * matrix cores; csrc/kernels.hpp JetRowModel).  Instantiated for n = 12 and n = 50 (BASELINE shapes C3 / C4);
 * data_dev: [P][m][n + 1] = (a_i, b_i) rows in natural layout; x: [P][n]. */
#define TOA_MODEL_DENSE_ROW_AD 11

/* robust norms / M-estimators (include/tinyopt/losses/robust_norms.h:32-316) */
#define TOA_LOSS_L2 0
#define TOA_LOSS_TRUNCATED 1
#define TOA_LOSS_HUBER 2
#define TOA_LOSS_TUKEY 3
#define TOA_LOSS_ARCTAN 4
#define TOA_LOSS_CAUCHY 5
#define TOA_LOSS_GEMAN_MCCLURE 6
#define TOA_LOSS_BLAKE_ZISSERMAN 7

/* StopReason — identical values to include/tinyopt/stop_reasons.h:14-43 */
#define TOA_STOP_OUT_OF_MEMORY (-4)
#define TOA_STOP_SOLVER_FAILED (-3)
#define TOA_STOP_NAN_OR_INF (-2)
#define TOA_STOP_SKIPPED (-1)
#define TOA_STOP_NONE 0
#define TOA_STOP_MIN_ERROR 1
#define TOA_STOP_MIN_REL_ERROR 2
#define TOA_STOP_MIN_DELTA_NORM 3
#define TOA_STOP_MIN_GRAD_NORM 4
#define TOA_STOP_MAX_ITERS 5
#define TOA_STOP_MAX_NO_DECR 6
#define TOA_STOP_MAX_CONSEC_NO_DECR 7
#define TOA_STOP_TIMED_OUT 8
#define TOA_STOP_USER_STOPPED 9

/* POD mirror of tinyopt::Options (include/tinyopt/optimizers/options.h:18-156), numeric knobs only.
 * Not mirrored (host-side concerns, evaluated by the header adaptor between launches or unsupported
 * on the device path): log.*, stop_callback, stop_callback2, max_duration_ms. */
typedef struct toa_options {
  int32_t solver_type;          /* 0 LevenbergMarquardt, 1 GaussNewton            options.h:24-30 */
  int32_t max_iters;            /* uint16 in the reference, default 50           options.h:89   */
  float min_error;              /* 1e-12f                                        options.h:90   */
  float min_rerr_dec;           /* 1e-10f                                        options.h:91   */
  float min_step_norm2;         /* 1e-14f                                        options.h:92   */
  float min_grad_norm2;         /* 1e-18f                                        options.h:93   */
  int32_t max_total_failures;   /* uint8, 0 = unlimited                          options.h:94   */
  int32_t max_consec_failures;  /* uint8, default 5, 0 = unlimited               options.h:95   */
  float damping_init;           /* 1e-4f; 0 disables damping                     options.h:133  */
  float damping_min;            /* 1e-9f                                         options.h:136  */
  float damping_max;            /* 1e9f                                          options.h:136  */
  float good_factor;            /* 1/3                                           options.h:138  */
  float bad_factor;             /* 2                                             options.h:139  */
  float grad_clipping;          /* 0 = off                                       options.h:49   */
  float check_min_H_diag;       /* 0 = off                                       options.h:63   */
  uint8_t check_final_cost;     /* false                                         options.h:43   */
  uint8_t use_step_quality_approx; /* false                                      options.h:46   */
  uint8_t use_ldlt;             /* true                                          options.h:59   */
  uint8_t H_is_full;            /* true (device H is always symmetric-full)      options.h:61   */
  uint8_t save_last;            /* true: export final undamped H                 options.h:66   */
  uint8_t use_squared_norm;     /* true                                          options.h:76   */
  uint8_t downscale_by_2;       /* false                                         options.h:77   */
  uint8_t normalize;            /* false                                         options.h:79   */
} toa_options;

/* `Options{}` defaults (options.h:43-148). */
void toa_options_default(toa_options* o);
/* benchmarks/options.h:10-27 `CreateOptions()`. */
void toa_options_benchmark(toa_options* o);

/* Per-problem results of a batched solve — POD mirror of tinyopt::Output (include/tinyopt/output.h:26-145).
 * Every pointer is a DEVICE pointer to an array of P entries (or NULL to skip that output, except
 * stop_reason / num_iters / final_cost which are required). */
typedef struct toa_results {
  int32_t* stop_reason;        /* [P]  Output::stop_reason                     output.h:108 */
  int32_t* num_iters;          /* [P]  Output::num_iters (failed iters count)  output.h:115, optimizer.h:307 */
  int32_t* num_failures;       /* [P]  Output::num_failures                    output.h:116 */
  int32_t* num_consec_failures;/* [P]  Output::num_consec_failures             output.h:117 */
  double* final_cost;          /* [P]  Output::final_cost.cost                 output.h:104 */
  int32_t* final_num_residuals;/* [P]  Output::final_cost.num_resisuals                      */
  double* final_rerr_dec;      /* [P]  Output::final_rerr_dec                  output.h:105 */
  double* final_hessian;       /* [P][n*n] undamped final H as double (optimizer.h:313-316), col-major; NULL or !save_last: skipped */
  double* errs;                /* [P][hist_stride] Output::errs                output.h:140 */
  double* deltas2;             /* [P][hist_stride] Output::deltas2             output.h:141 */
  uint8_t* successes;          /* [P][hist_stride] Output::successes           output.h:142 */
  int32_t hist_stride;         /* >= max_iters+2 when history pointers are given */
  int32_t _pad;
  float* final_inlier_ratio;   /* [P]  Output::final_cost.inlier_ratio (cost.h:84-95); 1 for models without a robust loss; NULL: skipped */
} toa_results;

typedef struct toa_context* toa_handle;

/* ---- lifetime (replaces: `lm::Optimizer<H_t> optimizer(options)` construction, optimize.h:50-53) ---- */
int toa_create(toa_handle* out, int device, void* stream);

/* ---- ABI version.  4 (round 4): counters_dev arrays are [TOA_NUM_COUNTERS = 8] uint64 (they were [4] up to version 2 — a caller
 *      that still allocates 4 entries would be written out of bounds by the memo counter), toa_tuning / toa_jit_spec exist.
 *      The host mirrors (include/tinyopt_amd/tinyopt.hpp, tinyopt_amd/_capi.py) refuse a library whose version differs. */
#define TOA_ABI_VERSION 6
int toa_abi_version(void);

/* ---- tuning (per handle).  The arms of the A/B logs (profiles/r0N_ab_log.md) and of the bit-identity tests, as typed state
 *      instead of environment variables: the product path reads NO environment variable.  All-zero = the library's choices;
 *      toa_set_tuning(h, NULL) restores them.  Nothing here changes a result beyond round-off (memo, coop: not at all —
 *      tests/test_gpu_memo.py, tests/test_gpu_coop.py). */
typedef struct toa_tuning {
  int32_t memo_off;              /* lm_fused_kernel: do not park / read back the last accepted linearisation */
  int32_t coop_off;              /* lm_fused_kernel: one chunk per pass, no cooperative tail */
  int32_t coop_chunks;           /* 0 = ~1024 rows per chunk; else 2 .. 64 chunks per pass */
  int32_t max_workgroups;        /* 0 = every resident slot; else a cap on lm_fused_kernel's grid (experiments) */
  int32_t wide_no_autosplit;     /* never pick the row-split / team forms automatically (one wavefront per problem) */
  int32_t wide_multilaunch;      /* row-split form: one (partial, step) launch pair per iteration instead of the persistent kernel */
  int32_t wide_no_team;          /* row-split form: no one-workgroup-per-problem team kernel */
  int32_t wide_team_max_per_cu;  /* 0 = automatic; problems per CU up to which the team form is chosen */
  int32_t wide_graph;            /* launch-per-iteration form replayed from a hipGraph */
  int32_t large_row_split;       /* 64 <= n <= 128: the row-split data pass instead of the tile-split one */
  int32_t large_pipeline;        /* the launch-per-stage pipeline for every n > 63 */
  int32_t large_library_gram;    /* n > 128: rocBLAS batched GEMM instead of the hand-written MFMA Gram */
  int32_t large_library_solver;  /* rocSOLVER potrf / potrs wherever a solver of our own would run (also toa_solve_damped for n <= 63) */
  int32_t fail_workspace_alloc;  /* TEST HOOK: the n > 128 workspace request fails as on a full device (kOutOfMemory path) */
  int32_t large_one_lane;        /* n > 128, own kernels: 1 = the whole batch on one stream instead of two half-batch lanes (same bits); k > 1 = k lanes by name
                                    (0: two lanes from 16 problems on, one at 224 < n <= 256 where the operand-sharing Gram runs) */
  int32_t large_chol_no_lookahead; /* n > 128: the one-workgroup Cholesky without its look-ahead (A/B and the bit-identity test) */
  int32_t large_gram_plain_deal; /* 224 < n <= 256, own Gram: the tiles dealt round-robin to the waves instead of the operand-sharing deal (triangles of
                                    blocks; A/B and the bit-identity test: which wave computes a tile does not change its bits) */
  int32_t narrow_mfma_pass;      /* TOA_MODEL_DENSE_ROW: 1 = the routes of rounds 1-5 — sixteen lanes per row for narrow blocks, the
                                    launch-per-iteration form for batches with an M-estimator — instead of the narrow routes of round 6 (an item per
                                    lane with the Gram in registers: fp32 n <= 10, fp64 n <= 5; a row per lane through the LDS stage: fp32 n = 11; with
                                    toa_set_loss also n = 12, 50 and fp64 n = 6, 12, 50; profiles/r06_ab_log.md sections 7, 8, 11; A/B) */
  int32_t se3_reproj_header_l2;  /* TOA_MODEL_SE3_REPROJ: 1 = the caller guarantees that no problem's data header names a loss (header[3] == 0 everywhere):
                                    the kernels WITHOUT the M-estimator branch run (fp64: 308 -> 216 registers, one -> two waves per SIMD; round 6,
                                    profiles/r06_ab_log.md section 12).  A header that names one anyway is then IGNORED.  Both host mirrors set it
                                    for models constructed without a loss. */
  int32_t reserved[13];          /* (three of them were the team form of the fused kernel, round 5: removed in round 6, profiles/r06_pruned_arms.patch) */
} toa_tuning;
int toa_set_tuning(toa_handle h, const toa_tuning* t);
int toa_get_tuning(toa_handle h, toa_tuning* out);
/* debug: per-problem start / end stamps of every lm_fused_kernel launch appended to `path` as text (tools/timeline.py); NULL = off */
int toa_debug_timeline(toa_handle h, const char* path);
int toa_destroy(toa_handle h);
const char* toa_last_error(void);
/* GPUs visible to the process (a sharded host program creates one handle per device; tests use it to skip N > 1 cases). */
int toa_device_count(int* count);
/* Device properties the measurement needs (CU count, clock, name). */
int toa_device_info(toa_handle h, int* num_cus, int* clock_khz, char* name, size_t name_len);

/* Measured read ceiling of this GPU's HBM (SURVEY §8d asks for a STREAM-like figure next to the nominal 8 TB/s):
 * streams `bytes` from src_dev `reps` times with 16-byte loads and reports GB/s between two HIP events. */
int toa_hbm_read_probe(toa_handle h, const void* src_dev, size_t bytes, int reps, double* gb_per_s);
/* Measured read ceiling of the 256 MiB Infinity Cache for the access pattern of this library's kernels (a problem's rows re-read
 * by its own waves): `bytes` (<= ~200 MB to stay on-die; larger buffers measure HBM) as two contiguous slices per compute unit,
 * six waves per slice; the rate of the re-reads, GB/s.  ~7.7 TB/s on MI355X against 6.3 from HBM: the XCD <-> IO-die fabric
 * bounds it (profiles/r05_ab_log.md §1a). */
int toa_llc_read_probe(toa_handle h, const void* src_dev, size_t bytes, double* gb_per_s);

/* ---- device memory conveniences for non-torch callers (the C++ header adaptor) ---- */
int toa_malloc(toa_handle h, void** dev_ptr, size_t bytes);
int toa_free(toa_handle h, void* dev_ptr);
int toa_memcpy_h2d(toa_handle h, void* dst_dev, const void* src_host, size_t bytes);  /* synchronous */
int toa_memcpy_d2h(toa_handle h, void* dst_host, const void* src_dev, size_t bytes);  /* synchronous */
int toa_memset(toa_handle h, void* dst_dev, int value, size_t bytes);                 /* stream-ordered */
int toa_synchronize(toa_handle h);

/* ---- DenseRow problem data --------------------------------------------------------------------
 * HBM layout ("packed"): per problem a [m4][RS] array of T, m4 = round_up(m,4).  A row is
 * [ main : RSM elements ][ thin : THIN elements ], RS = RSM + THIN:
 *   main  the columns contracted on the matrix cores, NB blocks of 16; a wavefront reads 4 rows
 *         with one coalesced NB*sizeof(T)-byte load per lane, already in MFMA operand order;
 *   thin  when n = 16*NB + (0..3): the last n-16*NB Jacobian columns followed by b (THIN = 1..4),
 *         contracted on the VALU; THIN = 0: b is the last main element.
 * Padding rows/columns are zero.  n <= 63.  n = 50 fp32: RS = 48 + 3 = 51 floats = m(n+1)*4 bytes. */
int toa_dense_row_layout(int dtype, int n, int m, int* nb, int* thin, int* row_stride, int* rows_padded,
                         size_t* bytes_per_problem);
/* Pack natural arrays (A: [P][m][n] row-major, b: [P][m], device pointers) into the layout above. */
int toa_dense_row_pack(toa_handle h, int dtype, int n, int m, int64_t P,
                       const void* A_dev, const void* b_dev, void* packed_dev);
/* Generate the synthetic DenseRow batch of SURVEY §8(d) directly in HBM (counter-based RNG keyed by
 * (seed, problem id, element); identical to oracle/synth.hpp).  problem0 = id of the first problem
 * (rank offset for sharded runs).  x0_dev / xstar_dev: [P][n] of T (either may be NULL). */
int toa_dense_row_synth(toa_handle h, int dtype, int n, int m, int64_t P, uint64_t seed, int64_t problem0,
                        void* packed_dev, void* x0_dev, void* xstar_dev);

/* ---- other device models (same entry points, `model` selects the functor) ----------------------------
 * TOA_MODEL_GAUSSIAN_PRIOR  m == n; data_dev: [P][2][n] = y then sigma; x: [P][n].
 * TOA_MODEL_SQRT2           n == m == 1; data_dev ignored; x: [P][1].
 * TOA_MODEL_SE3_REPROJ      n == 6 (tangent, Sophus order upsilon, omega), m = 2 * points; x: [P][12] = rotation matrix
 *                           (row-major) + translation, updated by pose <- pose * exp(delta)
 *                           (include/tinyopt/3rdparty/traits/sophus.h:24-26); data_dev: [P][8 + 5*m/2] =
 *                           [f, cx, cy, loss, th2, 0,0,0 | x, y, z, u, v per point].  loss = TOA_LOSS_* (0 = plain
 *                           squared L2), th2 = squared threshold in px^2: each point's ||r||^2 goes through the
 *                           M-estimator (cost += l, the point's J^T J and J^T r are scaled by s; robust_norms.h:20-26);
 *                           a point is an inlier when ||r||^2 <= th2 (results: final_inlier_ratio).
 * TOA_MODEL_TESTFN          see the define above.
 * TOA_MODEL_CIRCLE_FIT      n == 3; data_dev: [P][m][2] observed points; x: [P][3].
 * TOA_MODEL_DENSE_ROW_AD6   n == 6; data_dev: [P][m][7] = (a_i, b_i) rows (natural layout); x: [P][6].
 * TOA_MODEL_DENSE_ROW_NATURAL  1 <= n <= 4096 (any P: the n > 128 pipeline takes a large batch 65 535 problems at a time); data_dev: per problem A row-major [m][n] then b [m]
 *                           (problem stride m (n + 1) elements); x: [P][n]. */

/* ---- K1/K2: Accumulate callback (replaces `acc(x, grad, H) -> Cost`, docs/API.md:37-57;
 *      SolverGN::Accumulate gn.h:108-113 / Evaluate gn.h:97-105; AD closure optimize_autodiff.h:91-166).
 * want_grad = 0  <=>  grad == nullptr (cost only).  g_dev: [P][n] T; H_dev: [P][n*n] T full symmetric
 * (assigned, not accumulated); cost_dev: [P] double (= ||r||^2, un-normalised); nres_dev: [P] int32.
 * TOA_MODEL_DENSE_ROW_NATURAL: every n the layout takes (1 .. 4096) — 64 <= n <= 128 without an M-estimator: the data pass + fold of the
 * workgroup-per-problem kernel, one launch; otherwise (round 6: n > 128, n < 64, toa_set_loss at any n) ONE data pass of the
 * launch-per-stage pipeline (rows kernel + Gram).  With a loss, cost = the sum of the losses l_i and g, H are the weighted ones. */
int toa_accumulate(toa_handle h, int model, int dtype, int n, int m, int64_t P,
                   const void* data_dev, const void* x_dev, int want_grad,
                   void* g_dev, void* H_dev, double* cost_dev, int32_t* nres_dev);

/* ---- K3: damped solve (replaces SolverLM::Build's damping lm.h:108-117 + SolverGN::Solve gn.h:150-171
 *      -> SolveLDLT math.h:232-240).  H_ii <- H_ii * scale (double, Marquardt multiplicative), then
 *      dx = -H^-1 g by pivoted LDL^T with Eigen's acceptance rule (info()==Success && isPositive()).
 *      dx_dev: [P][n] T; ok_dev: [P] int32 (1 = solved, 0 = "not positive definite" => solver failure).
 *      n <= 63: one wavefront per matrix.  64 <= n <= 128: one workgroup per matrix (blocked LDL^T, trailing updates
 *      on the matrix cores).  128 < n <= 1024 (fp64: 512), P <= 65535: one workgroup per matrix, blocked Cholesky through L2.
 *      Beyond, up to 4096 (P <= 65535): rocSOLVER batched Cholesky (potrf + potrs). */
int toa_solve_damped(toa_handle h, int dtype, int n, int64_t P, const void* H_dev, const void* g_dev,
                     double scale, void* dx_dev, int32_t* ok_dev);

/* ---- robust norms alone (replaces `losses::Huber(n2, th2, true)` & co., robust_norms.h:32-316, docs/API.md:396-406):
 *      loss_dev[i], scale_dev[i] = rho(n2_dev[i], th2) for `count` squared norms of type `dtype`; kind = TOA_LOSS_*. */
int toa_robust_norm(toa_handle h, int kind, int dtype, int64_t count, const void* n2_dev, double th2,
                    void* loss_dev, void* scale_dev);

/* ---- the M-estimator of the handle's cost functor (replaces wrapping a residual in `losses::Huber(n2, th2, true)` & co.
 *      inside the user's cost functor, losses/robust_norms.h:20-26, docs/API.md:396-411): from now on every launch on
 *      this handle passes each residual ITEM's squared norm through rho (TOA_MODEL_DENSE_ROW, and
 *      TOA_MODEL_DENSE_ROW_NATURAL for every n it takes — the one-kernel form at 64 <= n <= 128, the launch-per-stage
 *      pipeline beyond it and for fp64 rows above n = 96, the stepping form at n >= 64 (round 5): each row's r_i^2;
 *      TOA_MODEL_CIRCLE_FIT / DENSE_ROW_AD6: each item's ||r||^2): cost += l, the item's J^T J and J^T r scaled by
 *      s = dl/dn2, inliers (n2 <= th2) reported through final_inlier_ratio (cost.h:84-95).  kind = TOA_LOSS_* (TOA_LOSS_L2
 *      = off, the default); th2 = squared threshold.  TOA_MODEL_SE3_REPROJ keeps its loss in its data header.  Both
 *      bundle-adjustment forms (toa_ba_run, toa_ba_lists_run) honour it per OBSERVATION (round 4): n2 = the squared norm of the
 *      observation's two reprojection residuals, both of them inliers when n2 <= th2.  The other families have none and
 *      refuse a handle that carries one. */
int toa_set_loss(toa_handle h, int kind, double th2);

/* ---- the dual numbers alone (replaces ceres::Jet<T, N>, include/tinyopt/3rdparty/ceres/jet.h:216-1400, as the residual
 *      functors of the device-AD families use it): out[i] = (f, df/da, df/db) of function `fn` at (a[i], b[i]), evaluated on
 *      Jet<T, 2> seeded on (a, b).  fn: 0 + 1 - 2 * 3 / 4 abs 5 log 6 exp 7 sqrt 8 cos 9 sin 10 tan 11 atan 12 tanh
 *      13 atan2(a,b) 14 pow(a,2.5) 15 acos 16 asin 17 sinh 18 cosh 19 cbrt 20 exp2 21 log2 22 log10 23 log1p 24 expm1
 *      25 hypot(a,b) 26 fmax 27 fmin 28 erf 29 erfc 30 pow(a,b) 31 pow(scalar a, b) 32 fma(a,b,a) 33 fdim 34 floor 35 ceil
 *      36 norm 37 copysign(a,b) 38 mixed scalar/Jet arithmetic 2/a + a/4 - 3b  39 hypot(a, b, a*b)
 *      40 BesselJ0 41 BesselJ1 42 BesselJn(3, a) 43 cyl_bessel_j(0, a) + cyl_bessel_j(2, b) 44 lerp(a, b, a*b) 45 midpoint(a, b)
 *      46 the classification / comparison functions (isfinite isinf isnan isnormal signbit isless ... fpclassify) of (a, b) as
 *      a bit mask in the value (jet.h:958-1218).  a, b: [count] of T;
 *      out: [count][3] of T. */
int toa_jet_eval(toa_handle h, int fn, int dtype, int64_t count, const void* a_dev, const void* b_dev, void* out_dev);

/* ---- covariance seam (replaces tinyopt::InvCov / DenseInvCov, math.h:41-91, used by Output::Covariance
 *      output.h:80-94 and SolverLM::Covariance lm.h:174): C = H^-1 by LDL^T against the identity, same acceptance
 *      rule as SolveLDLT.  H_dev, C_dev: [P][n*n] T; ok_dev: [P] int32 (0 = "not invertible" => std::nullopt).
 *      n > 63 (up to 4096, P <= 65535): Cholesky against the identity through rocSOLVER. */
int toa_inv_cov(toa_handle h, int dtype, int n, int64_t P, const void* H_dev, void* C_dev, int32_t* ok_dev);

#define TOA_NUM_COUNTERS 8

/* ---- fused batched solve (replaces Optimizer_::OptimizeAcc optimizer.h:242-327 + Step :331-539 +
 *      SolverLM lm.h:46-171 for P independent problems).  x_dev: [P][n] T, updated in place
 *      (reference: `x` by non-const ref).  One launch; no host round trips; each wavefront runs whole
 *      problems to their StopReason.  counters_dev (optional, [TOA_NUM_COUNTERS = 8] uint64): {accumulate passes
 *      that streamed the rows, evaluate-only passes, linear solves, problems, Builds served WITHOUT streaming the rows
 *      (see below), 3 reserved} ADDED to by every path (the caller zeroes them) — [0] + [1] are the units the roofline
 *      accounting in bench.py multiplies by the algorithmic bytes per pass.
 *      Builds without a data pass ([4]): the Accumulate callback is a pure function of x, and the loop asks for the same
 *      linearisation twice in two places — a failed solve re-enters Build at the same x (optimizer.h:358-393), and the
 *      iteration after a rejected step accumulates again at the rolled-back point (optimizer.h:283-287, :266).  The fused
 *      kernel of the DenseRow families parks the Gram registers of every accepted point (one slot per resident wave) and,
 *      when the roll-back restored x BIT FOR BIT, reads them back instead of streaming the rows again; g, H and the cost
 *      are the bits a second pass would have produced (tests/test_gpu_memo.py; toa_tuning::memo_off switches the memo off).
 *      One asynchronous launch on the handle's stream, capturable into a hipGraph once a first un-captured call of the
 *      shape has made the handle's workspaces (they grow on demand, which a capture cannot do: such a call is refused with
 *      TOA_E_UNSUPPORTED, the capture left valid; tests/test_gpu_graph_capture.py) — except TOA_MODEL_DENSE_ROW_NATURAL beyond
 *      n = 128 (a launch per stage): the
 *      host enqueues two passes ahead and waits for the (active, want-Jacobian) pair of pass k only before it enqueues pass
 *      k + 2 where every stage is a kernel of this library (fp32, 16-byte aligned rows, n <= 1024) — the call returns when
 *      the solve is done; under stream capture that form records its whole pass budget instead (every kernel skips finished
 *      problems) and the graph replays the solve with no host in the loop; with a library stage in the pass (no capture) (fp64 / odd shapes: rocBLAS
 *      GEMM, rocSOLVER beyond the LDS) the pair is read back after every pass.  64 <= n <= 128 is one persistent kernel
 *      like the rest.  A capture of that form records at most 1024 passes: options whose pass budget is larger
 *      (max_consec_failures == 0: up to 255 retries per iteration) are refused under capture with the node count.
 *      BATCH INDEPENDENCE: within one execution form a problem's bits depend neither on the batch size nor on its position
 *      in the batch (every sum has a fixed order; tests/test_gpu_coop.py, test_gpu_large_n.py).  The FORM, however, is chosen
 *      from the batch shape — a handful of huge problems is split by rows over the chip (n <= 63: P * 4 <= #CUs and m >= 512;
 *      64 <= n <= 128: m * n >= 393 216 and P * m * n <= 2^25), everything else is one wavefront / one workgroup per problem
 *      — and two forms sum in different orders: the same problem solved in a batch on either side of such a crossover agrees
 *      to rounding, not bit for bit (toa_tuning::wide_no_autosplit pins the per-problem form).
 *      LIFETIME under graphs: a captured launch bakes the device pointers of the handle's workspaces (scratch, memo, work
 *      arrays) into the graph.  Once any launch of a handle has been captured, a workspace that a later, larger eager call
 *      outgrows is no longer freed but kept until toa_destroy (the new block is allocated beside it), so an earlier graph can
 *      be replayed at any time before toa_destroy — never after it (tests/test_gpu_graph_capture.py). */
int toa_lm_run(toa_handle h, int model, int dtype, int n, int m, int64_t P,
               const void* data_dev, void* x_dev, const toa_options* options,
               const toa_results* results, uint64_t* counters_dev);

/* ---- stepping form (replaces `lm::Optimizer<H_t> optimizer(options)` + `optimizer.Step(x, acc, out)`,
 *      include/tinyopt/optimizers/optimizer.h:199,331-539; the class form `optimizer(x, f, max_iters)` is a loop of
 *      steps): the same state machine as toa_lm_run, ONE pass of the loop body (optimizer.h:266-310) per call for every
 *      problem that is still running, with the per-problem state (damping, counters, last step, ...) parked in
 *      state_dev (toa_lm_state_bytes(dtype, n, P) bytes of device memory, opaque: the scalar state plus the H of each
 *      problem's last build, which eval-only iterations keep solving with — lm.h:96-117) between calls.
 *        toa_lm_begin  constructs the state from x and the options (no data pass);
 *        toa_lm_step   runs one iteration: x_dev is updated in place (as the reference does at every Step), finished
 *                      problems get their full results exactly as in toa_lm_run and are skipped by later calls,
 *                      running ones report num_iters / final_cost so far with stop_reason == kNone;
 *                      active_dev (optional, int32, zeroed by the caller) += 1 per problem still running.
 *      Every family steps.  TOA_MODEL_DENSE_ROW_NATURAL (64 <= n <= 1024) steps on the launch-per-stage kernels of
 *      csrc/large_n.hip whatever n is (bit for bit what toa_lm_run gives through that pipeline —
 *      toa_tuning::large_pipeline; the one-kernel form of 64 <= n <= 128 sums in another order): at most 65 535 problems per
 *      call, no handle loss, and toa_lm_step reads three integers back (a solve that failed and is retried with a larger
 *      damping stays inside its iteration, optimizer.h:370-390: the retry passes need a count), so it blocks the host.
 */
size_t toa_lm_state_bytes(int dtype, int n, int64_t P);
int toa_lm_begin(toa_handle h, int model, int dtype, int n, int m, int64_t P, const void* data_dev, void* x_dev,
                 const toa_options* options, const toa_results* results, void* state_dev);
int toa_lm_step(toa_handle h, int model, int dtype, int n, int m, int64_t P, const void* data_dev, void* x_dev,
                const toa_options* options, const toa_results* results, uint64_t* counters_dev, void* state_dev,
                int32_t* active_dev);

/* ---- host-side stop controls through the stepping form (replaces Options::stop_callback / stop_callback2,
 *      include/tinyopt/optimizers/options.h:97-106, consulted at optimizer.h:529-534 when no numeric stop test fired, and
 *      Options::max_duration_ms -> kTimedOut, optimizer.h:302-305).  Host callables cannot run on the device, so the
 *      adaptors run the loop of toa_lm_step calls themselves:
 *        toa_lm_step_info  after a step: err (the iteration's cost), |dx|^2, |g|^2 [P] double and, optionally, the dx and
 *                          g vectors [P][n] of T of every problem's last iteration (any pointer may be NULL);
 *        toa_lm_stop       ends the still-running problems p with stop_request_dev[p] != 0 with that StopReason
 *                          (TOA_STOP_USER_STOPPED / TOA_STOP_TIMED_OUT): results written exactly as toa_lm_step writes
 *                          them for a problem that stops by itself (final undamped Hessian included). */
int toa_lm_step_info(toa_handle h, int dtype, int n, int64_t P, const void* state_dev, double* err_dev, double* dx_norm2_dev,
                     double* grad_norm2_dev, void* dx_dev, void* g_dev);
int toa_lm_stop(toa_handle h, int model, int dtype, int n, int m, int64_t P, const void* data_dev, void* x_dev,
                const toa_options* options, const toa_results* results, uint64_t* counters_dev, void* state_dev,
                const int32_t* stop_request_dev);
/* The rest of what the reference's per-iteration log line prints (optimizer.h:463-516, off-by-default Options::log; the adaptors
 * form the line from this and toa_lm_step_info): after a step, per problem, the solver's damping lambda (SolverLM::stateAsString
 * prints 1 / lambda, lm.h:150-154; 0 for Gauss-Newton), the residual count of the iteration's cost and its inlier residuals
 * (Cost::NumInliers, cost.h:84).  Any pointer may be NULL. */
int toa_lm_step_log(toa_handle h, int dtype, int n, int64_t P, const void* state_dev, double* lambda_dev, int32_t* num_residuals_dev,
                    int32_t* num_inliers_dev);

/* ---- row-split execution of the same solve, for FEW, HUGE problems (BASELINE configs C2 / C5: P = 1,
 *      m = 10^3 .. 5*10^4).  Same contract and results as toa_lm_run; the rows of each problem are split into
 *      `splits` chunks (0 = choose automatically) whose partial (H, g, cost) are folded in a fixed order before
 *      each LM iteration, nothing read back by the host: ONE persistent launch when P * chunks <= #CUs (the chunk waves
 *      hand over through generation counters in HBM), else one (partial, step) kernel pair per iteration.
 *      toa_lm_run selects this path by itself when P*4 <= #CUs and m >= 512, and — team form: the chunk waves of a
 *      problem are the waves of one workgroup, hand-over through LDS and workgroup barriers — for batches of small
 *      problems (n <= 15, 512 <= m <= 4096) up to one or two problems per compute unit.  DenseRow and SE3Reproj only. */
int toa_lm_run_split(toa_handle h, int model, int dtype, int n, int m, int64_t P,
                     const void* data_dev, void* x_dev, const toa_options* options,
                     const toa_results* results, uint64_t* counters_dev, int splits);

/* ---- bundle adjustment with the points eliminated (SURVEY §8f rank 4 "block-sparse / Schur"; replaces running
 *      tinyopt::Optimize on ONE parameter object x = (C SE3 poses, N points) with the full (6C + 3N)^2 Hessian — dense
 *      LDL^T, math.h:232-240, or Eigen's SimplicialLDLT on the sparse matrix, math.h:266-277, README.md:30,165-167).
 *      Same Levenberg-Marquardt state machine, StopReasons and Output fields; the linear step uses the block structure:
 *      per-point 3x3 blocks eliminated, the reduced camera system (6C <= 60 unknowns) accumulated on the matrix cores and
 *      solved by the workgroup's blocked LDL^T (pivoted one-wavefront fallback), points recovered by back-substitution; Marquardt damping on every diagonal
 *      entry.  One workgroup per scene, one launch per solve, P independent scenes per call (P <= 65535).
 *        x_dev:    [P][12 C + 3 N] of T = C poses (rotation matrix row-major, translation), then N points; updated in
 *                  place: pose <- pose * exp(delta) (3rdparty/traits/sophus.h:24-26), point += delta (traits.h:184-190)
 *        data_dev: [P][8 + 3 C N] of T = [f cx cy 0 0 0 0 0 | uv: C x N x 2 | vis: C x N (1 observed, 0 not)]
 *      results->final_hessian is not written (the block Hessian is not exported). */
int toa_ba_run(toa_handle h, int dtype, int num_cameras, int num_points, int64_t P, const void* data_dev, void* x_dev,
               const toa_options* options, const toa_results* results, uint64_t* counters_dev);

/* ---- bundle adjustment with VISIBILITY LISTS: tens to hundreds of cameras, each point observed by a few of them — the shape
 *      Eigen's SimplicialLDLT path exists for (math.h:266-277; README.md:30,165-167 "sparse is slow").  Same unknowns, update
 *      rules, state machine, StopReasons and Output fields as toa_ba_run; the observations arrive as a LIST instead of a dense
 *      C x N mask, and the reduced camera system (6 C unknowns, in HBM) is solved by the workgroup LDL^T up to 128 unknowns, by
 *      the one-workgroup blocked Cholesky up to 512 (85 cameras in fp64; 170 in fp32) and by rocSOLVER's potrf + potrs (opened
 *      with dlopen, one scene per call) beyond — up to 682 cameras.  A pipeline of small
 *      kernels per Build + Solve attempt (csrc/ba_schur.hip, "bl_*"); every sum has a fixed order.  The host enqueues two passes
 *      ahead and reads each pass's stop flag (a pinned ring) two passes late; the call returns when the solve is done.  Under
 *      stream capture (round 5) the whole pass budget of the options is recorded instead — (max_iters + 2) x (max_consec_failures + 1)
 *      passes, at most 256, needs max_consec_failures > 0, use_ldlt, max_duration_ms == 0, workspaces from an earlier eager call of
 *      the shape; scenes still running at the end of the budget end with kMaxIters — and the replay gives the bits of the eager
 *      call (the graph LIFETIME rule above applies).  With max_duration_ms > 0 the flag is read after every pass, which is where it is
 *      honoured: > 0 ends every scene still running with kTimedOut once the launches' device time exceeds it
 *      (Options::max_duration_ms, optimizer.h:302-305).
 *        intr_dev:    [P][4] of T = f cx cy 0
 *        obs_cam_dev, obs_pt_dev: [P][num_obs] int32, SORTED by (point, camera), each pair at most once (a scene whose list is
 *                     malformed ends with kSkipped); obs_uv_dev: [P][num_obs][2] of T (pixels)
 *        x_dev:       [P][12 C + 3 N] as for toa_ba_run, updated in place. */
int toa_ba_lists_run(toa_handle h, int dtype, int num_cameras, int num_points, int num_obs, int64_t P, const void* intr_dev,
                     const int32_t* obs_cam_dev, const int32_t* obs_pt_dev, const void* obs_uv_dev, void* x_dev,
                     const toa_options* options, const toa_results* results, uint64_t* counters_dev, double max_duration_ms);

/* ---- run-time user functors (replaces "pass any callable": `Optimize(x, [](const auto& x) { return r(x); })`,
 *      include/tinyopt/optimize.h:16-33, optimizers/optimizer.h:145-160, docs/API.md:21-35 — the residual is a C++ template
 *      the reference differentiates with ceres::Jet when the USER's program is compiled).  A device path cannot take a host
 *      callable; instead the user hands over the BODY of the residual as C++ source text, written like the reference's
 *      lambda, generic in its scalar type:
 *          const S dx = p[0] - x[0];  const S dy = p[1] - x[1];  r[0] = dx * dx + dy * dy - x[2] * x[2];     (tests/circle.cpp:32-68)
 *      with  S  the scalar type (toa::Jet<T, N> for Accumulate, plain T for the cost-only form: the same text serves both,
 *      optimize_autodiff.h:91-166),  x[j]  parameter j as an S,  p[k]  the item's data scalars and  h[k]  the problem's header
 *      scalars as T,  r[q]  the item's residuals; every function of ceres::Jet (jet.h:557-1400: sin, exp, pow, atan2, ...)
 *      is in scope.  toa_model_compile builds lm_fused_kernel / accumulate_kernel for JetModel<T, that functor> with hiprtc
 *      (opened with dlopen on first use; ~2-3 s, once) and loads the code object: no rebuild of the library.
 *        num_params <= 63 (narrow blocks — up to 10 parameters in fp32, 5 in fp64 —, items of several residuals up to 12 parameters and
 *        TOA_MANIFOLD_SE3: JetModel, an item per lane and the Gram in registers; beyond: RowModel, see toa_model_compile_ex below — the border
 *        is a measured one, profiles/r06_ab_log.md section 11); data_dev: [P][header_scalars + num_items * scalars_per_item]; x_dev: [P][num_params];
 *        m = num_items * residuals_per_item residuals per problem.  log_out (optional): the compiler's diagnostics.
 *        toa_jit_lm_run / toa_jit_accumulate: the contracts of toa_lm_run / toa_accumulate.  The handle's M-estimator
 *        (toa_set_loss) applies to each item's squared residual norm, as for TOA_MODEL_CIRCLE_FIT. */
typedef struct toa_jit_model_s* toa_jit_model;
int toa_model_compile(toa_handle h, int dtype, int num_params, int residuals_per_item, int scalars_per_item, int header_scalars,
                      const char* residual_body, toa_jit_model* out, char* log_out, size_t log_cap);
/*      Round 4 — the general form.
 *        num_params up to 63: beyond the narrow blocks named above the model is RowModel (csrc/row_model.hpp, round 6; the path of
 *          TOA_MODEL_DENSE_ROW_AD; the row-split / stepping kernels of a model with up to 12 parameters stay on JetModel, whose one-launch
 *          persistent form serves few, huge problems):
 *          an item is evaluated by ONE lane — TOA_JIT_RESIDUAL bodies on Jets, twelve parameters at a time; TOA_JIT_ACCUMULATE
 *          bodies on plain T with the Jacobian rows they fill — and its rows [J | r] are staged through LDS into the operand
 *          layout of the matrix-core Gram.  Euclidean parameters or TOA_MANIFOLD_USER (round 6: up to 64 stored scalars; an AD body is then differentiated through
 *          x (+) d, whose Jets are formed once per pass, a TOA_JIT_ACCUMULATE body fills J over the TANGENT); the handle's
 *          M-estimator (toa_set_loss) applies per item; residuals_per_item up to 8 (an item's residuals are consecutive rows),
 *          scalars_per_item up to 512 (as many as the LDS stage holds with num_params: the build is refused with a message
 *          otherwise); the row-split and the stepping forms take it too.
 *        manifold = TOA_MANIFOLD_SE3: x is ONE pose stored as R (row-major 9) + t (3) = 12 scalars, num_params = 6 (its tangent
 *          in Sophus order upsilon, omega); the body reads the pose through x[0..11] — Jets over the right perturbation
 *          x * exp(delta) at delta = 0 (optimize_autodiff.h:48-77, 3rdparty/traits/sophus.h:13-27) — and may call
 *          se3_log<S, T>(R, t, xi); the update is pose <- pose * exp(delta).  tests/sophus.cpp:26-44 `Optimize(pose, lambda)`.
 *        manifold = TOA_MANIFOLD_USER (round 5): the caller's own parameter container — the reference's extension point
 *          traits::params_trait<T> (traits.h:103-359; 3rdparty/traits/lieplusplus.h) as text.  x is stored as x_scalars scalars
 *          per problem, num_params is the dimension of its tangent, and plus_body is the body of
 *              template <class S> void plus(const T* x, const S* d, S* xp)        xp[0 .. x_scalars) = x (+) d
 *          written once over the scalar type S like the residual.  The update is x <- plus(x, +-delta) on plain T (PlusEq,
 *          traits.h:184-190; the roll-back is plus(x, -last_delta), optimizer.h:283-287), the derivative is taken through
 *          plus(x, Jets seeded on d at d = 0) (optimize_autodiff.h:48-77); the residual body reads x[0 .. x_scalars).
 *        kind = TOA_JIT_ACCUMULATE: a manual Accumulate callback (docs/API.md:37-57, tests/optimize_easy.cpp:35-79) — the body
 *          fills r[q] and, `if (want_grad)`, the Jacobian rows J[q][a] itself (plain T, no AD); x[j], h[k], p[k] as before.
 *          Any num_params up to 63 (round 6; benchmarks/dense.cpp:57-66,90-99 is such a callback at n = 50: bench.py --workload c4_text).
 *        A compiled model is cached on disk (code object keyed by the generated source, the library's headers, the hiprtc
 *          version and the device architecture): toa_jit_set_cache_dir(dir), default $XDG_CACHE_HOME/tinyopt_amd or
 *          $HOME/.cache/tinyopt_amd; "" = off, NULL = the default again.  The library's headers are embedded in it: no source tree is needed at run time. */
#define TOA_MANIFOLD_EUCLID 0
#define TOA_MANIFOLD_SE3 1
#define TOA_MANIFOLD_USER 2   /* spec.plus_body + spec.x_scalars: a user parameter container (traits::params_trait<T>, traits.h:103-359) */
#define TOA_JIT_RESIDUAL 0
#define TOA_JIT_ACCUMULATE 1
typedef struct toa_jit_spec {
  int32_t dtype, num_params, residuals_per_item, scalars_per_item, header_scalars;
  int32_t manifold;   /* TOA_MANIFOLD_* */
  int32_t kind;       /* TOA_JIT_* */
  int32_t x_scalars;  /* TOA_MANIFOLD_USER: scalars of x as STORED ([P][x_scalars]; 1 .. 32, beyond 12 parameters 1 .. 64); num_params = the tangent's dimension */
  const char* plus_body;   /* TOA_MANIFOLD_USER: the body of `template <class S> void plus(const T* x, const S* d, S* xp)`: xp = x (+) d */
  int32_t reserved[6];
} toa_jit_spec;
int toa_model_compile_ex(toa_handle h, const toa_jit_spec* spec, const char* body, toa_jit_model* out, char* log_out, size_t log_cap);
int toa_jit_set_cache_dir(const char* dir);
int toa_jit_model_info(toa_jit_model m, int* from_cache, int* xdim);   /* from_cache: 1 = the code object came from the disk cache */
/* What the run-time build of the fused kernel came out as: resident workgroups (of four wavefronts) per compute unit, its dynamic
 * LDS per workgroup, vector registers (VGPR + AGPR) per lane and scratch bytes per lane (0 = nothing spilled).  Any pointer may be NULL.
 * A model has TWO builds (round 6): the one toa_model_compile makes and this call describes is WITHOUT the M-estimator branch of the passes
 * — the estimators' exp / log / atan2, in double precision above all, cost every kernel that merely contains them its occupancy —; the one
 * with it is made the first time the model runs on a handle that has a loss set (toa_set_loss): 2-3 s once, then from the disk cache, and
 * refused (TOA_E_UNSUPPORTED, nothing recorded) if that first time is under stream capture. */
int toa_jit_model_stats(toa_jit_model m, int* wg_per_cu, int* lds_bytes_per_wg, int* num_regs, int* scratch_bytes);
int toa_model_destroy(toa_jit_model m);   /* waits for the model's last launch before the code is unloaded */
int toa_jit_lm_run(toa_handle h, toa_jit_model model, int num_items, int64_t P, const void* data_dev, void* x_dev,
                   const toa_options* options, const toa_results* results, uint64_t* counters_dev);
int toa_jit_accumulate(toa_handle h, toa_jit_model model, int num_items, int64_t P, const void* data_dev, const void* x_dev,
                       int want_grad, void* g_dev, void* H_dev, double* cost_dev, int32_t* nres_dev);
/*      Row-split execution of a run-time model (any num_params; beyond 15 always the launch-per-iteration form) for FEW, HUGE problems — the reference's one `Optimize(x, cost)`
 *      over tens of thousands of residuals (BASELINE C2 / C5 shapes) with the residual supplied as text: the contract of
 *      toa_lm_run_split.  The items of each problem are cut into `splits` chunks (0 = chosen automatically), a wavefront per
 *      chunk, partials folded in fixed order; ONE persistent launch when P * splits <= the device's compute units, one launch
 *      pair per iteration otherwise.  Its kernels are a second code object, compiled (or loaded from the cache) at the first
 *      such call.  toa_jit_lm_run takes this route by itself when P * 4 <= #CUs and m >= 512 (toa_tuning::wide_no_autosplit) —
 *      since round 5 also for models beyond 12 parameters.  Consequences for a caller: the first such call pays that second
 *      build, and it is refused with TOA_E_UNSUPPORTED (nothing recorded) when that first call happens under stream capture:
 *      run the shape once un-captured beforehand, or set wide_no_autosplit to keep the one-wavefront form; the chunked sum
 *      order differs from the one-wavefront form's, so results across the crossover agree to round-off, not bit for bit. */
int toa_jit_lm_run_split(toa_handle h, toa_jit_model model, int num_items, int64_t P, const void* data_dev, void* x_dev,
                         const toa_options* options, const toa_results* results, uint64_t* counters_dev, int splits);
/*      The stepping form of a run-time model (any num_params): the contracts of toa_lm_begin / toa_lm_step / toa_lm_stop with
 *      state_dev = toa_lm_state_bytes(dtype, num_params, P) bytes; toa_lm_step_info reads that block as for the built-in
 *      families.  This is what lets Options::stop_callback / stop_callback2 / max_duration_ms (options.h:96-106) work for a
 *      residual that arrived as text: both host mirrors run their callback loop over it. */
int toa_jit_lm_begin(toa_handle h, toa_jit_model model, int num_items, int64_t P, const void* data_dev, void* x_dev,
                     const toa_options* options, const toa_results* results, void* state_dev);
int toa_jit_lm_step(toa_handle h, toa_jit_model model, int num_items, int64_t P, const void* data_dev, void* x_dev,
                    const toa_options* options, const toa_results* results, uint64_t* counters_dev, void* state_dev,
                    int32_t* active_dev);
int toa_jit_lm_stop(toa_handle h, toa_jit_model model, int num_items, int64_t P, const void* data_dev, void* x_dev,
                    const toa_options* options, const toa_results* results, uint64_t* counters_dev, void* state_dev,
                    const int32_t* stop_request_dev);

/* ---- C1: the result gather of a sharded batch (SURVEY §8(b) export list `gather(handle_group...)`, §8(e)).
 *      Problems are independent (the reference optimises exactly one x per call, docs/API.md:12), so a batch of P_total
 *      problems shards across the GPUs of a node with no data-path communication: one process per GPU, rank g solves the
 *      contiguous id block toa_shard_range(P_total, g, nranks) with toa_lm_run.  Exactly ONE collective ends the job:
 *      toa_gather = pack -> ncclGather (RCCL over xGMI) -> unpack on the root, in problem-id order, native types
 *      (xdim scalars of x, stop_reason, num_iters, final_cost per problem: 216 B at C4 = 2.7 MB per GPU).
 *        toa_comm_unique_id  called by ONE rank; the host program hands the 128 bytes to every rank (MPI_Bcast, a file,
 *                            torch.distributed's store ...) — the library has no side channel of its own;
 *        toa_comm_init_rank  collective over the ranks of that id (ncclCommInitRank on the handle's GPU);
 *        toa_gather          stream-ordered on the handle's stream.  local: the rank's result arrays — stop_reason,
 *                            num_iters and final_cost are required; all / x_all_dev: destination arrays of P_total entries on
 *                            the root (ignored elsewhere; NULL members are skipped).  xdim: stored scalars of x per
 *                            problem, any width (n = 50 at C4; up to 1024 for TOA_MODEL_DENSE_ROW_NATURAL; 12 C + 3 N for
 *                            toa_ba_run) — only P_total x record bytes is bounded (1 TiB over all ranks).
 *      RCCL is opened with dlopen on first use (TOA_E_UNSUPPORTED if absent); single-problem configs (C2, C5) do not shard. */
#define TOA_COMM_ID_BYTES 128
typedef struct toa_comm_s* toa_comm;
int toa_shard_range(int64_t P_total, int rank, int nranks, int64_t* lo, int64_t* hi);
int toa_comm_unique_id(void* id_bytes_out);
int toa_comm_init_rank(toa_handle h, const void* id_bytes, int nranks, int rank, toa_comm* out);
int toa_comm_destroy(toa_comm c);
int toa_gather(toa_handle h, toa_comm c, int dtype, int xdim, int64_t P_total, const void* x_dev, const toa_results* local,
               int root, void* x_all_dev, const toa_results* all);

#ifdef __cplusplus
}
#endif
#endif /* TINYOPT_AMD_H_ */
