// K4 — per-problem Levenberg-Marquardt / Gauss-Newton state machine, executed wave-uniformly.
//
// Device restatement of Optimizer_::OptimizeAcc / Step (include/tinyopt/optimizers/optimizer.h:
// 242-327, 331-539) and SolverLM / SolverGN (include/tinyopt/solvers/lm.h:46-171, gn.h:77-171).
// Line references below are to those files.  All 64 lanes carry identical copies of the scalar
// state (the compiler keeps it in SGPRs where it can); lane j additionally owns x_j, dx_j.
//
// A `Model` supplies the Accumulate callback contract (docs/API.md:37-57) on the device:
//   accumulate(ctx)  -> fills ctx.g[] , ctx.hd[] (UNDAMPED diagonal), returns ||r||^2 and #residuals;
//                       keeps whatever it needs to later emit H (registers or LDS)
//   evaluate(ctx)    -> cost only (grad == nullptr)
//   write_sym(M, LD) -> emit the symmetric undamped H into an LD-strided image
#pragma once
#include "../../include/tinyopt_amd.h"
#include "ldlt_lds.hpp"
#include "wave_utils.hpp"

namespace toa {

template <typename T>
__device__ __forceinline__ T float_epsilon();  // math.h:297-301
template <>
__device__ __forceinline__ float float_epsilon<float>() { return 1e-4f; }
template <>
__device__ __forceinline__ double float_epsilon<double>() { return double(1e-7f); }

constexpr double kDblMax = 1.7976931348623157e+308;

// Per-wave LDS carve.  All vectors are 64 elements so lane-indexed access needs no bounds.
template <typename T>
struct WaveLds {
  T* M;       // n × LD factorisation workspace
  T* xs;      // x (broadcast source for the model)
  T* g;       // gradient J^T r
  T* hd;      // CURRENT (damped) Hessian diagonal, lm.h:108-117 acts on this
  T* tmp;     // scratch
  T* vec;     // scratch
  int* perm;  // pivot permutation
  int LD;
  static __host__ __device__ int ld_for(int n) { return n | 1; }
  static __host__ __device__ size_t bytes(int n) {
    return (size_t(n) * ld_for(n) + 5 * 64) * sizeof(T) + 64 * sizeof(int);
  }
  __device__ static WaveLds carve(char* base, int n) {
    WaveLds w;
    w.LD = ld_for(n);
    T* p = reinterpret_cast<T*>(base);
    w.M = p; p += size_t(n) * w.LD;
    w.xs = p; p += 64;
    w.g = p; p += 64;
    w.hd = p; p += 64;
    w.tmp = p; p += 64;
    w.vec = p; p += 64;
    w.perm = reinterpret_cast<int*>(p);
    return w;
  }
};

struct LmCounters {
  unsigned long long acc_passes, eval_passes, solves, problems;
};

// base.h:41-45
__device__ __forceinline__ double normalize_cost(double c, int nres, const toa_options& o) {
  if (!o.use_squared_norm) c = sqrt(c);
  if (o.downscale_by_2) c *= 0.5f;
  if (o.normalize && nres > 0) c /= nres;
  return c;
}

// Runs one problem to its StopReason.  x_lane: this lane's x_j (lanes >= n ignored), updated in place.
template <typename T, typename Model>
__device__ __forceinline__ void lm_solve_problem(Model& model, WaveLds<T>& L, const int n, const int lane,
                                                 T& x_lane, const toa_options& opt, const toa_results& res,
                                                 const long long p, LmCounters& cnt) {
  const bool in_n = lane < n;
  const bool is_lm = opt.solver_type == 0;
  // ---- SolverLM::reset  lm.h:46-52
  T lambda = opt.damping_init, prev_lambda = 0, bad_factor = opt.bad_factor;
  bool rebuild = true;
  // ---- Output  output.h:104-117
  double final_cost = kDblMax;
  int final_nres = 0;
  double final_rerr = kDblMax;
  int stop = TOA_STOP_NONE;
  int num_iters = 0;
  unsigned num_failures = 0, num_consec = 0;  // uint8_t in the reference: wrap explicitly
  // ---- solver cost_  base.h:64
  double cost_val = 0;
  int cost_nres = 0;
  // ---- OptimizeAcc locals  optimizer.h:248-263
  int max_iters = opt.max_iters + 1 + (opt.check_final_cost ? 1 : 0);
  T last_dx = 0;
  bool has_last_dx = false, last_was_success = true;
  const int hs = res.hist_stride;

  auto bad_step = [&]() {  // lm.h:140-145
    if (!is_lm) return;
    const T s = bad_factor;
    prev_lambda = lambda;
    lambda = fmin(fmax(lambda * s, T(opt.damping_min)), T(opt.damping_max));
    bad_factor *= T(opt.bad_factor);
  };
  auto good_step = [&](T quality) {  // lm.h:123-137
    if (!is_lm) return;
    T s = opt.good_factor;
    if (quality != T(0)) {
      const T q = T(2.0f) * quality - T(1.0f);
      s = fmax(s, T(1.0f) - q * q * q);
    }
    if (bad_factor != T(opt.bad_factor)) s /= bad_factor;
    prev_lambda = lambda;
    lambda = fmin(fmax(lambda * s, T(opt.damping_min)), T(opt.damping_max));
    bad_factor = opt.bad_factor;
  };

  for (int iter = 0; iter < max_iters; ++iter) {
    // ================= Step  optimizer.h:331-539 =================
    bool good = false, has_dx = false;
    T dx = 0;
    do {  // single-pass block so that `break` == "return status" in the reference
      bool solver_failed = true;
      const unsigned max_tries = opt.max_consec_failures > 0 ? (opt.max_consec_failures > 1 ? opt.max_consec_failures : 1) : 255;
      bool early = false;
      while (num_consec <= max_tries) {  // :358
        // ---- Build  lm.h:59-120 / gn.h:117-147
        bool built;
        if (!is_lm || rebuild) {
          wave_sync();
          L.xs[lane] = in_n ? x_lane : T(0);
          wave_sync();
          T c;
          int nres;
          model.accumulate(L, n, lane, c, nres);  // clear + acc(x, grad, H), gn.h:77-81,109-113
          cnt.acc_passes++;
          cost_val = normalize_cost(double(c), nres, opt);
          cost_nres = nres;
          built = cost_nres > 0 && cost_val != kDblMax;  // cost.h:83 isValid
          if (built) {
            if (opt.grad_clipping != 0) {  // base.h:29-38
              const T mm = opt.grad_clipping;
              if (in_n) L.g[lane] = fmin(fmax(L.g[lane], -mm), mm);
            }
            if (opt.check_min_H_diag > 0) {  // lm.h:82-86
              const bool low = in_n && fabs(L.hd[lane]) < T(opt.check_min_H_diag);
              if (__any(low)) built = false;
            }
          }
        } else {  // lm.h:96-105 -> gn.h:97-105 Evaluate(x, acc(x, nullptr, dummy), save)
          wave_sync();
          L.xs[lane] = in_n ? x_lane : T(0);
          wave_sync();
          T c;
          int nres;
          model.evaluate(L, n, lane, c, nres);
          cnt.eval_passes++;
          cost_val = normalize_cost(double(c), nres, opt);
          cost_nres = nres;
          built = cost_nres > 0 && cost_val != kDblMax;
        }
        if (built && is_lm && lambda > T(0)) {  // lm.h:108-117, s in double
          const double s = rebuild ? 1.0 + double(lambda) : (1.0 + double(lambda)) / (1.0 + double(prev_lambda));
          if (in_n) L.hd[lane] = T(double(L.hd[lane]) * s);
        }
        // ---- Solve  gn.h:150-171
        if (built) {
          wave_sync();
          bool ok;
          if (opt.use_ldlt) {
            model.write_sym(L.M, L.LD, n, lane);
            wave_sync();
            if (in_n) L.M[lane * L.LD + lane] = L.hd[lane];
            ok = ldlt_factor_wave<T>(L.M, L.LD, L.perm, L.tmp, n, lane);
            if (ok) dx = ldlt_solve_wave<T>(L.M, L.LD, L.perm, L.vec, n, lane, in_n ? -L.g[lane] : T(0));
          } else {  // gn.h:157-162, Dims == 1 branch only (host rejects n > 1 without LDLT)
            const T h = L.hd[0];
            dx = (h > float_epsilon<T>()) ? -(T(1) / h) * L.g[0] : T(0);
            dx = in_n ? dx : T(0);
            ok = true;
          }
          cnt.solves++;
          if (ok) solver_failed = false;
        }
        if (solver_failed) {  // :370-390
          num_consec = (num_consec + 1) & 0xff;
          num_failures = (num_failures + 1) & 0xff;
          if (cost_nres == 0) { stop = TOA_STOP_SKIPPED; early = true; break; }
          else if (isnan(cost_val) || isinf(cost_val)) { stop = TOA_STOP_NAN_OR_INF; early = true; break; }
          else if (opt.max_consec_failures > 0 && num_consec >= unsigned(opt.max_consec_failures)) {
            if (final_cost < double(NumLimits<T>::max())) stop = TOA_STOP_MAX_CONSEC_NO_DECR;
            break;
          }
          bad_step();  // FailedStep == BadStep  lm.h:148
        } else {
          break;
        }
      }
      if (early) break;
      if (solver_failed) { stop = TOA_STOP_SOLVER_FAILED; break; }  // :396-399
      const double err = cost_val;
      if (isnan(err) || isinf(err)) { stop = TOA_STOP_NAN_OR_INF; break; }  // :405-409
      const double dx_norm2 = double(wave_allreduce_sum(in_n ? dx * dx : T(0)));  // :412
      const bool has_g2 = opt.min_grad_norm2 > 0.0f;                                // :413-415
      double grad_norm2 = 0.0;
      if (has_g2) {
        const T gl = in_n ? L.g[lane] : T(0);
        grad_norm2 = double(wave_allreduce_sum(gl * gl));
      }
      if (isnan(dx_norm2) || isinf(dx_norm2)) { stop = TOA_STOP_NAN_OR_INF; break; }  // :416-425
      const double derr = err - final_cost;                                           // :428
      const bool is_good_step = derr < double(T(0.0));                                // :429
      const double rel_derr = (final_cost > double(float_epsilon<T>()) && final_cost < double(NumLimits<T>::max()))
                                  ? (final_cost - err) / final_cost
                                  : 0.0;                                              // :431-434
      if (lane == 0 && num_iters < hs) {                                              // :436-438
        if (res.errs) res.errs[p * hs + num_iters] = err;
        if (res.deltas2) res.deltas2[p * hs + num_iters] = dx_norm2;
        if (res.successes) res.successes[p * hs + num_iters] = is_good_step ? 1 : 0;
      }
      if (is_good_step || iter == 0) {  // :441-446
        if (iter > 0) good_step(opt.use_step_quality_approx ? T(rel_derr) : T(0.0f));
        num_consec = 0;
        final_cost = cost_val;
        final_nres = cost_nres;
        final_rerr = rel_derr;
      } else {  // :447-460
        bad_step();
        num_failures = (num_failures + 1) & 0xff;
        num_consec = (num_consec + 1) & 0xff;
        if (opt.max_consec_failures > 0 && num_consec >= unsigned(opt.max_consec_failures)) {
          stop = TOA_STOP_MAX_CONSEC_NO_DECR;
          break;
        }
        if (opt.max_total_failures > 0 && num_failures >= unsigned(opt.max_total_failures)) {
          stop = TOA_STOP_MAX_NO_DECR;
          break;
        }
      }
      // :519-534 stop tests, fixed priority
      if (opt.min_error > 0 && err < double(opt.min_error)) stop = TOA_STOP_MIN_ERROR;
      else if (opt.min_rerr_dec > 0 && rel_derr > 0.0 && rel_derr < double(opt.min_rerr_dec)) stop = TOA_STOP_MIN_REL_ERROR;
      else if (opt.min_step_norm2 > 0 && dx_norm2 < double(opt.min_step_norm2)) stop = TOA_STOP_MIN_DELTA_NORM;
      else if (opt.min_grad_norm2 > 0 && grad_norm2 < double(opt.min_grad_norm2)) stop = TOA_STOP_MIN_GRAD_NORM;
      good = is_good_step;
      has_dx = true;
    } while (false);

    // ================= back in OptimizeAcc  optimizer.h:269-309 =================
    bool eval_only = false;
    if (good) {                     // :271-279
      x_lane += dx;                 // PlusEq, traits.h:184-190
      last_dx = dx;
      has_last_dx = true;
      last_was_success = true;
      if (opt.check_final_cost && iter + 1 == max_iters) eval_only = true;
    } else {                        // :281-297
      if (has_last_dx) {
        x_lane += -last_dx;
        has_last_dx = false;
      } else if (has_dx) {
        x_lane += dx;
        last_dx = dx;
        has_last_dx = true;
      }
      eval_only = (last_was_success == false);
      last_was_success = false;
    }
    if (is_lm) rebuild = !eval_only;  // :299, lm.h:55 (GN: base.h:56 no-op)
    num_iters++;                      // :307
    if (stop != TOA_STOP_NONE) break; // :309
  }
  if (stop == TOA_STOP_NONE && num_iters >= max_iters) stop = TOA_STOP_MAX_ITERS;  // :320-321

  // ---- final Hessian, undamped  optimizer.h:313-316, lm.h:157-171
  if (opt.save_last && res.final_hessian) {
    double* Hout = res.final_hessian + size_t(p) * n * n;
    wave_sync();
    model.write_sym(Hout, n, n, lane);
    if (in_n) {
      T d = L.hd[lane];
      if (is_lm && prev_lambda > T(0)) d = d / (T(1.0f) + prev_lambda);
      Hout[lane * n + lane] = double(d);
    }
  }
  if (lane == 0) {
    res.stop_reason[p] = stop;
    res.num_iters[p] = num_iters;
    res.final_cost[p] = final_cost;
    if (res.num_failures) res.num_failures[p] = int(num_failures);
    if (res.num_consec_failures) res.num_consec_failures[p] = int(num_consec);
    if (res.final_num_residuals) res.final_num_residuals[p] = final_nres;
    if (res.final_rerr_dec) res.final_rerr_dec[p] = final_rerr;
  }
  cnt.problems++;
}

}  // namespace toa
