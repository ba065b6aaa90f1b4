// K4 — per-problem Levenberg-Marquardt / Gauss-Newton state machine, executed wave-uniformly.
//
// Device restatement of Optimizer_::OptimizeAcc / Step (include/tinyopt/optimizers/optimizer.h:
// 242-327, 331-539) and SolverLM / SolverGN (include/tinyopt/solvers/lm.h:46-171, gn.h:77-171).
// Line references below are to those files.
//
// Register discipline.  The accumulate pass (K1) is the only hot loop and wants the whole VGPR
// budget (Gram accumulators + a prefetch ring).  Everything the state machine must remember between
// passes therefore lives in a per-wave LDS record (`LmState`, plus x / dx / last_dx vectors and a
// private copy of the options/results PODs), and `reg_fence()` barriers around the passes stop the
// compiler from caching any of it in registers across them.  The state machine itself runs once per
// ~10^5 cycles, so its LDS round trips are free.
//
// A `Model` supplies the Accumulate callback contract (docs/API.md:37-57) on the device:
//   accumulate(L, n, lane, cost, nres)  fills L.g[], L.hd[] (UNDAMPED diagonal); keeps whatever it
//                                        needs to later emit H (registers or LDS)
//   evaluate(L, n, lane, cost, nres)    cost only (grad == nullptr)
//   write_sym(M, LD, n, lane)           emit the symmetric undamped H into an LD-strided image
#pragma once
#include <type_traits>

#ifdef __HIPCC_RTC__   // run-time compilation (jit.hip): the headers are handed to hiprtc by NAME, embedded in the library
#include "tinyopt_amd.h"
#else
#include "../../include/tinyopt_amd.h"
#endif
#include "ldlt_blocked.hpp"
#include "ldlt_lds.hpp"
#include "ldlt_regs.hpp"
#include "wave_utils.hpp"

namespace toa {

template <typename T>
__device__ __forceinline__ T float_epsilon();  // math.h:297-301
template <>
__device__ __forceinline__ float float_epsilon<float>() { return 1e-4f; }
template <>
__device__ __forceinline__ double float_epsilon<double>() { return double(1e-7f); }

constexpr double kDblMax = 1.7976931348623157e+308;

// Inlier residuals of the model's last pass: models with a robust loss expose `ninl`, the others count every
// residual as an inlier (Cost's default inlier_ratio = 1, cost.h:23,95).
template <typename M>
__device__ __forceinline__ auto model_inliers(const M& m, int nres, int) -> decltype(m.ninl) { return m.ninl < 0 ? nres : m.ninl; }
template <typename M>
__device__ __forceinline__ int model_inliers(const M&, int nres, long) { return nres; }

__device__ __forceinline__ bool bits_equal(float a, float b) { return __float_as_uint(a) == __float_as_uint(b); }
__device__ __forceinline__ bool bits_equal(double a, double b) { return __double_as_longlong(a) == __double_as_longlong(b); }

// Compiler-only barrier: nothing held in registers may be assumed equal to memory across it.
__device__ __forceinline__ void reg_fence() { asm volatile("" ::: "memory"); }

// Per-problem scalar state (one record per wave, in LDS).
template <typename T>
struct LmState {
  // SolverLM  lm.h:191-194
  T lambda, prev_lambda, bad_factor;
  int rebuild;
  // solver cost_  base.h:64
  double cost_val;
  int cost_nres;
  // Output  output.h:104-117
  int final_nres;
  int cost_ninl, final_ninl;  // inlier residuals (Cost::NumInliers, cost.h:84); == nres for models without a robust loss
  double final_cost, final_rerr;
  int stop, num_iters;
  unsigned num_failures, num_consec;  // uint8_t in the reference: wrapped explicitly
  // OptimizeAcc locals  optimizer.h:262-263
  int has_last_dx, last_was_success;
  int iter, max_iters;
  // Memo of linearisations (fused kernel, models with kMemo; see lm_memo_* below).  acc_at_x: the model's Gram registers hold
  // the linearisation AT the current x; memo_valid: the wave's HBM slot holds the Gram of the point tagged in L.xsv;
  // memo_hit: the roll-back of the last iteration restored exactly that point, the next Build may read the slot back.
  int acc_at_x, memo_valid, memo_hit;
  // per-wave work counters (summed into the batch counters at kernel exit)
  unsigned long long acc_passes, eval_passes, solves, problems;
  unsigned long long reused_passes;   // Builds served without streaming the rows (memo read-back or Gram still in registers)
  // this WAVE's memo slot in HBM (0 = memo off: every path but the fused kernel).  Kept here, in LDS, rather than in the
  // WaveLds carve: the carve lives in registers across the hot pass and the wave index is not provably uniform (2 VGPRs).
  unsigned long long memo_slot;
};

// Models whose Accumulate result can be parked and read back bit for bit declare `static constexpr bool kMemo = true` and
// supply memo_save / memo_restore / memo_reextract.
template <typename M, typename = void>
struct ModelMemo { static constexpr bool value = false; };
template <typename M>
struct ModelMemo<M, std::enable_if_t<M::kMemo>> { static constexpr bool value = true; };

// Per-wave LDS carve.  All vectors are 64 elements so lane-indexed access needs no bounds.
template <typename T>
struct WaveLds {
  T* M;       // n × LD factorisation workspace
  T* xs;      // x — the master copy of the parameters while a problem is being solved
  T* g;       // gradient J^T r
  T* hd;      // CURRENT (damped) Hessian diagonal, lm.h:108-117 acts on this
  T* tmp;     // scratch
  T* vec;     // scratch
  T* dx;      // step of the current iteration
  T* ldx;     // last accepted / tried step (optimizer.h:262 last_dx)
  T* aux;     // model scratch that survives the factorisation (row-split path: folded H for n <= 8)
  T* xsv;     // the x the memo slot's linearisation was taken at (lm_memo_*)
  int* perm;  // pivot permutation
  LmState<T>* st;
  toa_options* opt;
  toa_results* res;
  int LD;
  static __host__ __device__ int ld_for(int n) { return n | 1; }
  // workspace elements, rounded so that the 64-element vectors behind it stay 16-byte aligned (ds_read_b128)
  static __host__ __device__ size_t m_elems(int n) { return (size_t(n) * ld_for(n) + 3) & ~size_t(3); }
  static __host__ __device__ size_t bytes(int n) {
    size_t b = (m_elems(n) + 9 * 64) * sizeof(T) + 64 * sizeof(int);
    b = (b + 15) & ~size_t(15);
    b += (sizeof(LmState<T>) + 15) & ~size_t(15);
    b += (sizeof(toa_options) + 15) & ~size_t(15);
    b += (sizeof(toa_results) + 15) & ~size_t(15);
    return b;
  }
  // The carve STARTS with everything that is dead while a data pass runs — the factorisation workspace, the solve's scratch
  // vectors and permutation, the step (written by the solve, consumed before the next pass), the row-split scratch: a model
  // whose pass stages rows through LDS (DenseRowGram::pass16s) lays its stage over these bytes (lm_fused_kernel: the stage
  // begins in front of the carve and ends pass_dead_bytes into it).  Alive across a pass and therefore behind them: x, the
  // gradient and the damped diagonal (an evaluate-only iteration solves with the last build's), the last step, the memo's x.
  static __host__ __device__ size_t pass_dead_bytes(int n) { return (m_elems(n) + 4 * 64) * sizeof(T) + 64 * sizeof(int); }
  __device__ static WaveLds carve(char* base, int n) {
    WaveLds w;
    w.LD = ld_for(n);
    T* p = reinterpret_cast<T*>(base);
    w.M = p; p += m_elems(n);
    w.tmp = p; p += 64;
    w.vec = p; p += 64;
    w.dx = p; p += 64;
    w.aux = p; p += 64;
    w.perm = reinterpret_cast<int*>(p);
    p = reinterpret_cast<T*>(reinterpret_cast<char*>(p) + 64 * sizeof(int));
    w.xs = p; p += 64;
    w.g = p; p += 64;
    w.hd = p; p += 64;
    w.ldx = p; p += 64;
    w.xsv = p; p += 64;
    size_t off = (m_elems(n) + 9 * 64) * sizeof(T) + 64 * sizeof(int);
    off = (off + 15) & ~size_t(15);
    w.st = reinterpret_cast<LmState<T>*>(base + off);
    off += (sizeof(LmState<T>) + 15) & ~size_t(15);
    w.opt = reinterpret_cast<toa_options*>(base + off);
    off += (sizeof(toa_options) + 15) & ~size_t(15);
    w.res = reinterpret_cast<toa_results*>(base + off);
    return w;
  }
};

// base.h:41-45
__device__ __forceinline__ double normalize_cost(double c, int nres, const toa_options& o) {
  if (!o.use_squared_norm) c = sqrt(c);
  if (o.downscale_by_2) c *= 0.5f;
  if (o.normalize && nres > 0) c /= nres;
  return c;
}

template <typename T>
__device__ __forceinline__ void lm_bad_step(LmState<T>& S, const toa_options& o) {  // lm.h:140-145
  if (o.solver_type != 0) return;
  const T s = S.bad_factor;
  S.prev_lambda = S.lambda;
  S.lambda = fmin(fmax(S.lambda * s, T(o.damping_min)), T(o.damping_max));
  S.bad_factor = S.bad_factor * T(o.bad_factor);
}
template <typename T>
__device__ __forceinline__ void lm_good_step(LmState<T>& S, const toa_options& o, T quality) {  // lm.h:123-137
  if (o.solver_type != 0) return;
  T s = o.good_factor;
  if (quality != T(0)) {
    const T q = T(2.0f) * quality - T(1.0f);
    s = fmax(s, T(1.0f) - q * q * q);
  }
  if (S.bad_factor != T(o.bad_factor)) s /= S.bad_factor;
  S.prev_lambda = S.lambda;
  S.lambda = fmin(fmax(S.lambda * s, T(o.damping_min)), T(o.damping_max));
  S.bad_factor = o.bad_factor;
}

// ---- Build (lm.h:59-120 / gn.h:117-147) + Solve (gn.h:150-171) with the retry loop of
//      Step (optimizer.h:354-399).  Returns 0 = got a step (in L.dx), 1 = solver failed for good
//      (stop may or may not be set), 2 = early return with stop set.
template <typename T, typename Model>
__device__ __forceinline__ int lm_build_and_solve(Model& model, WaveLds<T>& L, const int n, const int lane) {
  LmState<T>& S = *L.st;
  const toa_options& opt = *L.opt;
  const bool in_n = lane < n;
  const bool is_lm = opt.solver_type == 0;
  const unsigned max_tries = opt.max_consec_failures > 0 ? (opt.max_consec_failures > 1 ? opt.max_consec_failures : 1) : 255;
  while (S.num_consec <= max_tries) {  // :358
    bool built;
    const bool do_acc = !is_lm || S.rebuild;
    T c;
    int nres;
    reg_fence();
    bool streamed = true;
    if (do_acc) {  // clear + acc(x, grad, H)  gn.h:77-81,109-113
      if constexpr (ModelMemo<Model>::value) {
        // The callback is a pure function of x: when the linearisation at THIS x (bit for bit) is still in the Gram
        // registers (a failed solve re-entering Build, optimizer.h:358-393) or in the wave's memo slot (the re-accumulation
        // after a roll-back, optimizer.h:283-287 + :266), it is taken from there instead of streaming the rows again.
        if (S.memo_slot && S.acc_at_x) { model.memo_reextract(L, n, lane, c, nres); streamed = false; }
        else if (S.memo_slot && S.memo_hit) { model.memo_restore(L, n, lane, c, nres); streamed = false; }
        else model.accumulate(L, n, lane, c, nres);
        S.acc_at_x = 1;
        S.memo_hit = 0;
      } else {
        model.accumulate(L, n, lane, c, nres);
      }
    } else {
      model.evaluate(L, n, lane, c, nres);           // lm.h:96-105 -> gn.h:97-105 (grad == nullptr)
    }
    reg_fence();
    if (!streamed) S.reused_passes++;
    else if (do_acc) S.acc_passes++;
    else S.eval_passes++;
    S.cost_val = normalize_cost(double(c), nres, opt);
    S.cost_nres = nres;
    S.cost_ninl = model_inliers(model, nres, 0);
    built = nres > 0 && S.cost_val != kDblMax;  // cost.h:83 isValid
    if (built && do_acc) {
      if (opt.grad_clipping != 0) {  // base.h:29-38
        const T mm = opt.grad_clipping;
        if (in_n) L.g[lane] = fmin(fmax(L.g[lane], -mm), mm);
      }
      if (opt.check_min_H_diag > 0) {  // lm.h:82-86
        const bool low = in_n && fabs(L.hd[lane]) < T(opt.check_min_H_diag);
        if (__any(low)) built = false;
      }
    }
    if (built && is_lm && S.lambda > T(0)) {  // lm.h:108-117, s in double
      const double s = S.rebuild ? 1.0 + double(S.lambda) : (1.0 + double(S.lambda)) / (1.0 + double(S.prev_lambda));
      if (in_n) L.hd[lane] = T(double(L.hd[lane]) * s);
    }
    bool solver_failed = true;
    if (built) {  // gn.h:150-171
      wave_sync();
      bool ok;
      if (opt.use_ldlt) {
        model.write_sym(L.M, L.LD, n, lane);
        wave_sync();
        if (in_n) L.M[lane * L.LD + lane] = L.hd[lane];
        wave_sync();
        const T rhs = in_n ? -L.g[lane] : T(0);
        {  // fast path: unpivoted LDL^T (positive-definite case) — one register panel, or blocked on the matrix cores
          LdltFast<T, Model::kNpad> F;
          ok = F.factor(L.M, L.LD, n, lane);
          if (ok) L.dx[lane] = F.solve(L.M, L.LD, n, lane, rhs);
          else if (LdltFast<T, Model::kNpad>::kClobbersM) {  // the blocked form works in place: re-create the image
            wave_sync();
            model.write_sym(L.M, L.LD, n, lane);
            wave_sync();
            if (in_n) L.M[lane * L.LD + lane] = L.hd[lane];
            wave_sync();
          }
        }
        if (!ok) {  // not safely positive definite: Eigen's pivoted algorithm + acceptance rule decides
          ok = ldlt_factor_wave<T>(L.M, L.LD, L.perm, L.tmp, n, lane);
          if (ok) L.dx[lane] = ldlt_solve_wave<T>(L.M, L.LD, L.perm, L.vec, n, lane, rhs);
        }
      } else if (n == 1) {  // gn.h:157-162, Dims == 1 branch
        const T h = L.hd[0];
        const T d = (h > float_epsilon<T>()) ? -(T(1) / h) * L.g[0] : T(0);
        L.dx[lane] = in_n ? d : T(0);
        ok = true;
      } else {  // gn.h:162: dx = -H.inverse() * g, UNCHECKED (no success flag, no positivity test).  Here: the pivoted
                // factorisation with its verdict ignored — the same solution to rounding for every non-singular H, definite
                // or not; a singular H gets the pseudo-inverse where Eigen's inverse() returns inf / nan
        model.write_sym(L.M, L.LD, n, lane);
        wave_sync();
        if (in_n) L.M[lane * L.LD + lane] = L.hd[lane];
        wave_sync();
        (void)ldlt_factor_wave<T>(L.M, L.LD, L.perm, L.tmp, n, lane);
        L.dx[lane] = ldlt_solve_wave<T>(L.M, L.LD, L.perm, L.vec, n, lane, in_n ? -L.g[lane] : T(0));
        ok = true;
      }
      wave_sync();
      S.solves++;
      if (ok) solver_failed = false;
    }
    if (!solver_failed) return 0;
    // :370-390
    S.num_consec = (S.num_consec + 1) & 0xff;
    S.num_failures = (S.num_failures + 1) & 0xff;
    if (S.cost_nres == 0) { S.stop = TOA_STOP_SKIPPED; return 2; }
    if (isnan(S.cost_val) || isinf(S.cost_val)) { S.stop = TOA_STOP_NAN_OR_INF; return 2; }
    if (opt.max_consec_failures > 0 && S.num_consec >= unsigned(opt.max_consec_failures)) {
      if (S.final_cost < double(NumLimits<T>::max())) S.stop = TOA_STOP_MAX_CONSEC_NO_DECR;
      return 1;
    }
    lm_bad_step(S, opt);  // FailedStep == BadStep  lm.h:148
  }
  return 1;
}

// ---- the rest of Step (optimizer.h:401-539) once |dx|^2 and |g|^2 are known: scalar, shared by the one-wavefront
//      state machine below and the large-n path (large_n.hip).  Returns bit0 = good step, bit1 = has dx.
template <typename T>
__device__ __forceinline__ int lm_judge_core(LmState<T>& S, const toa_options& opt, const toa_results& res, const long long p,
                                             const double dx_norm2, const double grad_norm2, const bool writer) {
  const double err = S.cost_val;
  if (isnan(err) || isinf(err)) { S.stop = TOA_STOP_NAN_OR_INF; return 0; }  // :405-409
  if (isnan(dx_norm2) || isinf(dx_norm2)) { S.stop = TOA_STOP_NAN_OR_INF; return 0; }  // :416-425
  const double final_cost = S.final_cost;
  const double derr = err - final_cost;             // :428
  const bool is_good_step = derr < double(T(0.0));  // :429
  const double rel_derr = (final_cost > double(float_epsilon<T>()) && final_cost < double(NumLimits<T>::max()))
                              ? (final_cost - err) / final_cost
                              : 0.0;                // :431-434
  const int hs = res.hist_stride;
  if (writer && S.num_iters < hs) {                 // :436-438
    if (res.errs) res.errs[p * hs + S.num_iters] = err;
    if (res.deltas2) res.deltas2[p * hs + S.num_iters] = dx_norm2;
    if (res.successes) res.successes[p * hs + S.num_iters] = is_good_step ? 1 : 0;
  }
  if (is_good_step || S.iter == 0) {  // :441-446
    if (S.iter > 0) lm_good_step(S, opt, opt.use_step_quality_approx ? T(rel_derr) : T(0.0f));
    S.num_consec = 0;
    S.final_cost = err;
    S.final_nres = S.cost_nres;
    S.final_ninl = S.cost_ninl;
    S.final_rerr = rel_derr;
  } else {  // :447-460
    lm_bad_step(S, opt);
    S.num_failures = (S.num_failures + 1) & 0xff;
    S.num_consec = (S.num_consec + 1) & 0xff;
    if (opt.max_consec_failures > 0 && S.num_consec >= unsigned(opt.max_consec_failures)) {
      S.stop = TOA_STOP_MAX_CONSEC_NO_DECR;
      return 0;
    }
    if (opt.max_total_failures > 0 && S.num_failures >= unsigned(opt.max_total_failures)) {
      S.stop = TOA_STOP_MAX_NO_DECR;
      return 0;
    }
  }
  // :519-534 stop tests, fixed priority
  if (opt.min_error > 0 && err < double(opt.min_error)) S.stop = TOA_STOP_MIN_ERROR;
  else if (opt.min_rerr_dec > 0 && rel_derr > 0.0 && rel_derr < double(opt.min_rerr_dec)) S.stop = TOA_STOP_MIN_REL_ERROR;
  else if (opt.min_step_norm2 > 0 && dx_norm2 < double(opt.min_step_norm2)) S.stop = TOA_STOP_MIN_DELTA_NORM;
  else if (opt.min_grad_norm2 > 0 && grad_norm2 < double(opt.min_grad_norm2)) S.stop = TOA_STOP_MIN_GRAD_NORM;
  return (is_good_step ? 1 : 0) | 2;
}

template <typename T>
__device__ __forceinline__ int lm_judge_step(WaveLds<T>& L, const int n, const int lane, const long long p) {
  LmState<T>& S = *L.st;
  const toa_options& opt = *L.opt;
  const bool in_n = lane < n;
  const double err = S.cost_val;
  if (isnan(err) || isinf(err)) { S.stop = TOA_STOP_NAN_OR_INF; return 0; }  // :405-409 (before the norms are formed)
  const T dxl = in_n ? L.dx[lane] : T(0);
  const double dx_norm2 = double(wave_allreduce_sum(dxl * dxl));  // :412
  double grad_norm2 = 0.0;
  if (opt.min_grad_norm2 > 0.0f) {  // :413-415
    const T gl = in_n ? L.g[lane] : T(0);
    grad_norm2 = double(wave_allreduce_sum(gl * gl));
  }
  return lm_judge_core<T>(S, opt, *L.res, p, dx_norm2, grad_norm2, lane == 0);
}

// ---- the three stages of OptimizeAcc (optimizer.h:242-327), separable so that the same state machine runs either
//      inside one wave for a whole solve (lm_solve_problem) or one iteration per launch with the state parked in
//      global memory between launches (the row-split "wide" path for single huge problems, kernels.hpp).
template <typename T>
__device__ __forceinline__ void lm_init(WaveLds<T>& L, const int lane) {
  LmState<T>& S = *L.st;
  const toa_options& opt = *L.opt;
  // SolverLM::reset  lm.h:46-52
  S.lambda = opt.damping_init; S.prev_lambda = 0; S.bad_factor = opt.bad_factor; S.rebuild = 1;
  // Output  output.h:104-117
  S.final_cost = kDblMax; S.final_nres = 0; S.final_ninl = 0; S.cost_ninl = 0; S.final_rerr = kDblMax;
  S.stop = TOA_STOP_NONE; S.num_iters = 0; S.num_failures = 0; S.num_consec = 0;
  S.cost_val = 0; S.cost_nres = 0;
  // OptimizeAcc locals  optimizer.h:248-263
  S.max_iters = opt.max_iters + 1 + (opt.check_final_cost ? 1 : 0);
  S.has_last_dx = 0; S.last_was_success = 1;
  S.iter = 0;
  S.acc_at_x = 0; S.memo_valid = 0; S.memo_hit = 0;
  L.ldx[lane] = T(0);
  L.dx[lane] = T(0);
  wave_sync();
}

// One pass of the loop body at optimizer.h:266-310 (uses and advances S.iter).  Returns false when the loop ends.
template <typename T, typename Model>
__device__ __forceinline__ bool lm_iteration(Model& model, WaveLds<T>& L, const int n, const int lane, const long long p) {
  LmState<T>& S = *L.st;
  // ================= Step  optimizer.h:331-539 =================
  int status = 0;  // bit0 good, bit1 has_dx
  const int rc = lm_build_and_solve<T>(model, L, n, lane);
  if (rc == 1) S.stop = TOA_STOP_SOLVER_FAILED;  // :396-399 (overwrites kMaxConsecNoDecr set just above)
  if (rc == 0) status = lm_judge_step<T>(L, n, lane, p);
  wave_sync();
  // ================= back in OptimizeAcc  optimizer.h:269-309 =================
  const toa_options& opt = *L.opt;
  bool eval_only = false;
  S.memo_hit = 0;                   // (only the roll-back below may arm it, for the Build that follows directly)
  if (status & 1) {                 // :271-279
    if constexpr (ModelMemo<Model>::value) {
      // x is about to leave an ACCEPTED point.  If the step that follows is rejected, the loop comes back here
      // (x (+)= -last_dx, :283-287) and accumulates again (:266 with rebuild == 1): park the linearisation of this point.
      // (An eval-only iteration that succeeds leaves from a point whose linearisation was never formed: nothing to park.)
      if (S.memo_slot) {
        if (S.acc_at_x) {
          model.memo_save(L, lane);
          L.xsv[lane] = L.xs[lane];
          S.memo_valid = 1;
        } else {
          S.memo_valid = 0;
        }
      }
    }
    model.plus_eq(L, L.dx, T(1), n, lane);   // ptrait::PlusEq(x, dx): traits.h:184-190 / sophus.h:24-26
    S.acc_at_x = 0;
    L.ldx[lane] = L.dx[lane];
    S.has_last_dx = 1;
    S.last_was_success = 1;
    if (opt.check_final_cost && S.iter + 1 == S.max_iters) eval_only = true;
  } else {                          // :281-297
    if (S.has_last_dx) {
      model.plus_eq(L, L.ldx, T(-1), n, lane);  // roll back: PlusEq(x, -last_dx)
      S.has_last_dx = 0;
      S.acc_at_x = 0;
      if constexpr (ModelMemo<Model>::value) {
        // (x + dx) - dx is x again only when both roundings cancel: compare the BIT PATTERNS with the parked point's, and
        // let the next Build read the memo back only on a match in every component (never an approximation).
        if (S.memo_slot) {
          wave_sync();
          S.memo_hit = (S.memo_valid && __all(bits_equal(L.xs[lane], L.xsv[lane]))) ? 1 : 0;
        }
      }
    } else if (status & 2) {
      model.plus_eq(L, L.dx, T(1), n, lane);
      S.acc_at_x = 0;
      L.ldx[lane] = L.dx[lane];
      S.has_last_dx = 1;
    }
    eval_only = (S.last_was_success == 0);
    S.last_was_success = 0;
  }
  if (opt.solver_type == 0) S.rebuild = eval_only ? 0 : 1;  // :299, lm.h:55 (GN: base.h:56 no-op)
  S.num_iters = S.num_iters + 1;                            // :307
  S.iter = S.iter + 1;
  wave_sync();
  return S.stop == TOA_STOP_NONE && S.iter < S.max_iters;   // :309 / loop bound :266
}

template <typename T, typename Model>
__device__ __forceinline__ void lm_finalize(Model& model, WaveLds<T>& L, const int n, const int lane, const long long p) {
  LmState<T>& S = *L.st;
  const bool in_n = lane < n;
  if (S.stop == TOA_STOP_NONE && S.num_iters >= S.max_iters) S.stop = TOA_STOP_MAX_ITERS;  // :320-321
  const toa_options& opt = *L.opt;
  const toa_results& res = *L.res;
  // ---- final Hessian, undamped  optimizer.h:313-316, lm.h:157-171
  if (opt.save_last && res.final_hessian && p >= 0) {
    double* Hout = res.final_hessian + size_t(p) * n * n;
    wave_sync();
    model.write_sym(Hout, n, n, lane);
    if (in_n) {
      T d = L.hd[lane];
      if (opt.solver_type == 0 && S.prev_lambda > T(0)) d = d / (T(1.0f) + S.prev_lambda);
      Hout[lane * n + lane] = double(d);
    }
  }
  if (lane == 0 && p >= 0) {   // (p < 0: the ghost problem of a helper wave, lm_fused_kernel)
    res.stop_reason[p] = S.stop;
    res.num_iters[p] = S.num_iters;
    res.final_cost[p] = S.final_cost;
    if (res.num_failures) res.num_failures[p] = int(S.num_failures);
    if (res.num_consec_failures) res.num_consec_failures[p] = int(S.num_consec);
    if (res.final_num_residuals) res.final_num_residuals[p] = S.final_nres;
    if (res.final_rerr_dec) res.final_rerr_dec[p] = S.final_rerr;
    if (res.final_inlier_ratio) res.final_inlier_ratio[p] = S.final_nres > 0 ? float(S.final_ninl) / float(S.final_nres) : 1.0f;
  }
  if (p >= 0) S.problems++;
  wave_sync();
}

// Runs one problem to its StopReason.  On entry L.xs[] holds x0 (lanes >= n: 0), on exit the result.
template <typename T, typename Model>
__device__ __forceinline__ void lm_solve_problem(Model& model, WaveLds<T>& L, const int n, const int lane,
                                                 const long long p) {
  lm_init<T>(L, lane);
  while (lm_iteration<T>(model, L, n, lane, p)) {}
  lm_finalize<T>(model, L, n, lane, p);
}

}  // namespace toa
