// ============================================================================
// TEST INFRASTRUCTURE ONLY — NOT PART OF THE PRODUCT PATH.
//
// CPU restatement ("oracle") of tinyopt's Gauss-Newton / Levenberg-Marquardt
// inner loop, in plain C++17 with no Eigen.  Only tests/, bench.py's
// `cpu_baseline` leg and __graft_entry__.smoke() may call into this; the
// product (tinyopt_amd/csrc + include/) never includes or links it.
//
// Every function cites the reference file:line it restates (paths relative to
// /root/reference/include/tinyopt unless stated).
//
// Third-party arithmetic on the path that is NOT under /root/reference:
//   Eigen 3.4.0 (pin: cmake/ThirdParties.cmake:19) — `LDLT` + `isPositive()`
//   used by SolveLDLT (math.h:232-240).  Eigen is absent from this image and
//   there is no network, so its published algorithm (Eigen/src/Cholesky/LDLT.h,
//   `ldlt_inplace<Lower>::unblocked` and `LDLT::_solve_impl`) is restated in
//   `ldlt_factor` / `ldlt_solve` below.  Summation order inside Eigen's
//   vectorised dot products is not reproducible, so parity with real Eigen is
//   "to rounding", never bitwise.
//
// Parity status: pinned against the reference's own known-answer tests
// (oracle/pin_reference_tests.cpp restates tests/sqrt2.cpp, basic.cpp,
// solvers.cpp, optimize_easy.cpp, optimize_hard.cpp, circle.cpp, cov.cpp and
// the README √2 trace).  The reference holds NO golden vectors for this path
// (SURVEY.md §4, §8c); per-iteration trajectories are pinned only by the README
// trace.  The reference itself cannot be built here (needs Eigen ≥3.4, Catch2,
// <format>), so there is no oracle/_ref.
// ============================================================================
#pragma once

#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace oracle {

// stop_reasons.h:14-43 — same integer values.
enum StopReason : int {
  kOutOfMemory = -4,
  kSolverFailed = -3,
  kSystemHasNaNOrInf = -2,
  kSkipped = -1,
  kNone = 0,
  kMinError,
  kMinRelError,
  kMinDeltaNorm,
  kMinGradNorm,
  kMaxIters,
  kMaxNoDecr,
  kMaxConsecNoDecr,
  kTimedOut,
  kUserStopped
};

// optimizers/options.h:18-156 — numeric knobs only (no logging, no callbacks:
// stop_callback* and max_duration_ms are host-side concerns outside the path).
struct Options {
  enum Solver { LevenbergMarquardt = 0, GaussNewton = 1 };
  int solver_type = LevenbergMarquardt;   // options.h:24-30
  bool check_final_cost = false;          // options.h:43
  bool use_step_quality_approx = false;   // options.h:46
  float grad_clipping = 0;                // options.h:49
  bool use_ldlt = true;                   // options.h:59
  bool H_is_full = true;                  // options.h:61
  float check_min_H_diag = 0;             // options.h:63
  bool save_last = true;                  // options.h:66
  bool use_squared_norm = true;           // options.h:76
  bool downscale_by_2 = false;            // options.h:77
  bool normalize = false;                 // options.h:79
  uint16_t max_iters = 50;                // options.h:89
  float min_error = 1e-12f;               // options.h:90
  float min_rerr_dec = 1e-10f;            // options.h:91
  float min_step_norm2 = 1e-14f;          // options.h:92
  float min_grad_norm2 = 1e-18f;          // options.h:93
  uint8_t max_total_failures = 0;         // options.h:94
  uint8_t max_consec_failures = 5;        // options.h:95
  float damping_init = 1e-4f;             // options.h:133
  std::array<float, 2> damping_range{{1e-9f, 1e9f}};  // options.h:136
  float good_factor = 1.0f / 3.0f;        // options.h:138
  float bad_factor = 2.0f;                // options.h:139
};

// benchmarks/options.h:10-27
inline Options BenchmarkOptions() {
  Options o;
  o.max_iters = 10;
  o.min_error = 0;
  o.min_rerr_dec = 1e-12f;
  o.min_step_norm2 = 1e-16f;
  o.max_consec_failures = 3;
  o.save_last = false;
  return o;
}

// cost.h:18-97
struct Cost {
  double cost = 0;
  int num_residuals = 0;
  float inlier_ratio = 1.0f;
  Cost() = default;
  Cost(double c) : cost(c), num_residuals(1) {}                 // cost.h:22 scalar -> (v,1)
  Cost(double c, int n, float ir = 1.0f) : cost(c), num_residuals(n), inlier_ratio(ir) {}
  bool isValid() const {                                        // cost.h:83
    return num_residuals > 0 && cost != std::numeric_limits<double>::max();
  }
};

// output.h:26-145
struct Output {
  Cost final_cost = Cost(std::numeric_limits<double>::max(), 0);   // output.h:104
  double final_rerr_dec = std::numeric_limits<double>::max();       // output.h:105
  int stop_reason = kNone;
  uint16_t num_residuals = 0;   // never written by Step
  uint16_t num_iters = 0;
  uint8_t num_failures = 0;
  uint8_t num_consec_failures = 0;
  std::vector<double> errs, deltas2;
  std::vector<uint8_t> successes;
  std::vector<double> final_hessian;  // n*n col-major, undamped; empty if not saved
  bool Succeeded() const { return stop_reason >= kNone; }                                // output.h:30
  bool Converged() const { return stop_reason >= kMinError && stop_reason < kMaxIters; } // output.h:33-35
};

// math.h:297-301.  Note the double branch really is the *float* literal 1e-7f widened.
template <typename T>
inline T FloatEpsilon() {
  return static_cast<T>(std::is_same<T, float>::value ? 1e-4f : 1e-7f);
}

// ---------------------------------------------------------------------------
// Eigen 3.4.0 LDLT restated (Eigen/src/Cholesky/LDLT.h, ldlt_inplace<Lower>::unblocked).
// `A` is n×n column-major; only the UPPER triangle is read
// (math.h:235 `A.selfadjointView<Upper>().ldlt()`: Eigen copies the self-adjoint view to a
// full matrix and factorises its lower part in place).
// Returns info()==Success; *positive receives isPositive().
// ---------------------------------------------------------------------------
template <typename T>
struct LDLT {
  int n = 0;
  std::vector<T> m;         // n×n col-major; strict lower = L, diagonal = D
  std::vector<int> transp;  // transpositions
  bool ok = false;
  int sign = 0;  // 0 ZeroSign, 1 PositiveSemiDef, -1 NegativeSemiDef, 2 Indefinite

  T& at(int r, int c) { return m[size_t(c) * n + r]; }
  T at(int r, int c) const { return m[size_t(c) * n + r]; }

  bool isPositive() const { return sign == 1 || sign == 0; }

  void compute(int n_, const T* A_upper_colmajor) {
    n = n_;
    m.assign(size_t(n) * n, T(0));
    transp.assign(n, 0);
    // selfadjointView<Upper> -> dense: m(r,c) = A(min,max)
    for (int c = 0; c < n; ++c)
      for (int r = 0; r < n; ++r) {
        const int i = std::min(r, c), j = std::max(r, c);
        at(r, c) = A_upper_colmajor[size_t(j) * n + i];
      }
    ok = true;
    sign = 0;
    if (n == 0) return;
    if (n == 1) {  // LDLT.h: size<=1 fast path
      transp[0] = 0;
      const T d = at(0, 0);
      if (d < T(0)) sign = -1;
      else if (d > T(0)) sign = 1;
      else sign = 0;
      return;
    }
    bool found_zero_pivot = false;
    std::vector<T> temp(n);
    for (int k = 0; k < n; ++k) {
      // biggest |diagonal| in the remaining corner
      int big = k;
      T bigv = std::abs(at(k, k));
      for (int i = k + 1; i < n; ++i) {
        const T v = std::abs(at(i, i));
        if (v > bigv) { bigv = v; big = i; }
      }
      transp[k] = big;
      if (k != big) {
        // symmetric swap of rows/cols k and big touching only the lower triangle
        const int s = n - big - 1;
        for (int c = 0; c < k; ++c) std::swap(at(k, c), at(big, c));
        for (int r = 0; r < s; ++r) std::swap(at(big + 1 + r, k), at(big + 1 + r, big));
        std::swap(at(k, k), at(big, big));
        for (int i = k + 1; i < big; ++i) std::swap(at(i, k), at(big, i));
      }
      const int rs = n - k - 1;
      if (k > 0) {
        for (int j = 0; j < k; ++j) temp[j] = at(j, j) * at(k, j);
        T acc = 0;
        for (int j = 0; j < k; ++j) acc += at(k, j) * temp[j];
        at(k, k) -= acc;
        for (int r = 0; r < rs; ++r) {
          T a = 0;
          for (int j = 0; j < k; ++j) a += at(k + 1 + r, j) * temp[j];
          at(k + 1 + r, k) -= a;
        }
      }
      const T realAkk = at(k, k);
      const bool pivot_is_valid = std::abs(realAkk) > T(0);
      if (k == 0 && !pivot_is_valid) {
        // LDLT.h "The entire diagonal is zero, there is nothing more to do except filling the
        // transpositions, and checking that the other entries are zero."
        sign = 0;
        for (int j = 0; j < n; ++j) {
          transp[j] = j;
          for (int r = j + 1; r < n; ++r) ok = ok && (at(r, j) == T(0));
        }
        return;
      }
      if (rs > 0 && pivot_is_valid) {
        for (int r = 0; r < rs; ++r) at(k + 1 + r, k) /= realAkk;
      } else if (rs > 0) {
        bool all_zero = true;
        for (int r = 0; r < rs; ++r) all_zero = all_zero && (at(k + 1 + r, k) == T(0));
        ok = ok && all_zero;
      }
      if (found_zero_pivot && pivot_is_valid) ok = false;  // factorization failed
      else if (!pivot_is_valid) found_zero_pivot = true;

      if (sign == 1) { if (realAkk < T(0)) sign = 2; }
      else if (sign == -1) { if (realAkk > T(0)) sign = 2; }
      else if (sign == 0) {
        if (realAkk > T(0)) sign = 1;
        else if (realAkk < T(0)) sign = -1;
      }
    }
  }

  // LDLT::_solve_impl: x = P^T L^-T D^+ L^-1 P b, D^+ pseudo-inverse with
  // tolerance = numeric_limits<T>::min().
  void solve(const T* b, T* x) const {
    for (int i = 0; i < n; ++i) x[i] = b[i];
    for (int i = 0; i < n; ++i) std::swap(x[i], x[transp[i]]);  // P b
    for (int i = 0; i < n; ++i)                                  // L^-1 (unit lower)
      for (int j = 0; j < i; ++j) x[i] -= at(i, j) * x[j];
    const T tol = std::numeric_limits<T>::min();
    for (int i = 0; i < n; ++i) {
      if (std::abs(at(i, i)) > tol) x[i] /= at(i, i);
      else x[i] = T(0);
    }
    for (int i = n - 1; i >= 0; --i)                             // L^-T
      for (int j = i + 1; j < n; ++j) x[i] -= at(j, i) * x[j];
    for (int i = n - 1; i >= 0; --i) std::swap(x[i], x[transp[i]]);  // P^T
  }
};

// math.h:232-240 SolveLDLT: success iff info()==Success && isPositive().
template <typename T>
inline bool SolveLDLT(int n, const T* A, const T* b, T* x) {
  LDLT<T> chol;
  chol.compute(n, A);
  if (chol.ok && chol.isPositive()) {
    chol.solve(b, x);
    return true;
  }
  return false;
}

// `-H.inverse()*g` path (gn.h:157-162) restated with partial-pivot LU (Eigen's
// PartialPivLU for dynamic sizes; closed forms for fixed n<=4 differ only by rounding).
template <typename T>
inline void SolveInverse(int n, const T* H, const T* b, T* x) {
  std::vector<T> a(H, H + size_t(n) * n);
  std::vector<int> p(n);
  for (int i = 0; i < n; ++i) { p[i] = i; x[i] = b[i]; }
  auto A = [&](int r, int c) -> T& { return a[size_t(c) * n + r]; };
  for (int k = 0; k < n; ++k) {
    int piv = k;
    for (int i = k + 1; i < n; ++i) if (std::abs(A(i, k)) > std::abs(A(piv, k))) piv = i;
    if (piv != k) { for (int c = 0; c < n; ++c) std::swap(A(k, c), A(piv, c)); std::swap(x[k], x[piv]); }
    for (int i = k + 1; i < n; ++i) {
      const T f = A(i, k) / A(k, k);
      A(i, k) = f;
      for (int c = k + 1; c < n; ++c) A(i, c) -= f * A(k, c);
      x[i] -= f * x[k];
    }
  }
  for (int i = n - 1; i >= 0; --i) {
    for (int c = i + 1; c < n; ++c) x[i] -= A(i, c) * x[c];
    x[i] /= A(i, i);
  }
}

// ---------------------------------------------------------------------------
// SolverGN (solvers/gn.h) + SolverLM (solvers/lm.h) restated as one class with
// an `is_lm` switch: SolverLM derives from SolverGN and overrides
// reset/Rebuild/Build/GoodStep/BadStep/FailedStep/Hessian.
// H is n×n column-major (Eigen default), g is n.
// Acc: Cost acc(const X& x, T* grad_or_null, T* H_or_null)  — the Accumulate
// callback contract (docs/API.md:37-57): grad==nullptr ⇒ cost only.
// ---------------------------------------------------------------------------
template <typename T>
class Solver {
 public:
  explicit Solver(const Options& o, int n) : opt(o), n_(n) {
    is_lm_ = (o.solver_type == Options::LevenbergMarquardt);
    H.assign(size_t(n) * n, T(0));
    g.assign(n, T(0));
    reset();
  }

  void reset() {  // lm.h:46-52 (gn.h:45: clear only)
    clear();
    lambda_ = opt.damping_init;
    prev_lambda_ = 0;
    bad_factor_ = opt.bad_factor;
    rebuild_ = true;
  }
  void clear() {  // gn.h:77-81
    std::fill(H.begin(), H.end(), T(0));
    std::fill(g.begin(), g.end(), T(0));
  }
  void Rebuild(bool b) { if (is_lm_) rebuild_ = b; }  // lm.h:55; base.h:56 no-op for GN

  // base.h:41-45
  void NormalizeCost(Cost& c) const {
    if (!opt.use_squared_norm) c.cost = std::sqrt(c.cost);
    if (opt.downscale_by_2) c.cost *= 0.5f;
    if (opt.normalize && c.num_residuals > 0) c.cost /= c.num_residuals;
  }

  // lm.h:59-120 (LM) / gn.h:117-147 (GN)
  template <typename X, typename Acc>
  bool Build(const X& x, const Acc& acc) {
    if (!is_lm_ || rebuild_) {
      clear();
      cost_ = acc(x, g.data(), H.data());  // gn.h:109-113 Accumulate
      NormalizeCost(cost_);
      if (!cost_.isValid()) return false;
      if (opt.grad_clipping != 0) {  // base.h:29-38
        const T mm = opt.grad_clipping;
        for (auto& v : g) v = std::min(std::max(v, -mm), mm);
      }
      if (opt.check_min_H_diag > 0) {  // lm.h:82-86
        for (int i = 0; i < n_; ++i)
          if (std::abs(H[size_t(i) * n_ + i]) < opt.check_min_H_diag) return false;
      }
      if (!opt.H_is_full && !opt.use_ldlt) {  // lm.h:89-94: lower = upper^T
        for (int c = 0; c < n_; ++c)
          for (int r = c + 1; r < n_; ++r) H[size_t(c) * n_ + r] = H[size_t(r) * n_ + c];
      }
    } else {  // lm.h:96-105 -> gn.h:97-105 Evaluate(x, acc, save=true)
      Cost c = acc(x, (T*)nullptr, (T*)nullptr);
      NormalizeCost(c);
      cost_ = c;
      if (!cost_.isValid()) return false;
    }
    if (is_lm_ && lambda_ > 0.0) {  // lm.h:108-117: multiplicative Marquardt damping, s in double
      const double s = rebuild_ ? 1.0 + lambda_ : (1.0 + lambda_) / (1.0 + prev_lambda_);
      for (int i = 0; i < n_; ++i) {
        T& d = H[size_t(i) * n_ + i];
        d = static_cast<T>(d * s);
      }
    }
    return true;
  }

  // gn.h:150-171
  bool Solve(T* dx) const {
    if (!cost_.isValid()) return false;
    std::vector<T> mg(n_);
    for (int i = 0; i < n_; ++i) mg[i] = -g[i];
    if (opt.use_ldlt) {
      return SolveLDLT<T>(n_, H.data(), mg.data(), dx);
    } else {
      if (n_ == 1) {
        if (H[0] > FloatEpsilon<T>()) dx[0] = -(T(1) / H[0]) * g[0];
        else dx[0] = T(0);
        return true;
      }
      SolveInverse<T>(n_, H.data(), mg.data(), dx);
      return true;
    }
  }

  void GoodStep(T quality) {  // lm.h:123-137
    if (!is_lm_) return;
    T s = opt.good_factor;
    if (quality != T(0.0)) s = std::max<T>(s, T(1.0f - std::pow(2.0f * quality - 1.0f, 3.0f)));
    if (bad_factor_ != opt.bad_factor) s /= bad_factor_;
    prev_lambda_ = lambda_;
    lambda_ = std::clamp<T>(lambda_ * s, opt.damping_range[0], opt.damping_range[1]);
    bad_factor_ = opt.bad_factor;
  }
  void BadStep() {  // lm.h:140-145
    if (!is_lm_) return;
    const T s = bad_factor_;
    prev_lambda_ = lambda_;
    lambda_ = std::clamp<T>(lambda_ * s, opt.damping_range[0], opt.damping_range[1]);
    bad_factor_ *= opt.bad_factor;
  }
  void FailedStep() { BadStep(); }  // lm.h:148

  // lm.h:157-171 (GN: gn.h:188 returns H_ as is)
  std::vector<T> Hessian() const {
    std::vector<T> out = H;
    if (is_lm_ && prev_lambda_ > 0.0) {
      const T s = 1.0f + prev_lambda_;
      for (int i = 0; i < n_; ++i) out[size_t(i) * n_ + i] /= s;
    }
    return out;
  }
  T GradientSquaredNorm() const {  // gn.h:196
    T s = 0;
    for (auto v : g) s += v * v;
    return s;
  }
  const Cost& cost() const { return cost_; }
  T lambda() const { return lambda_; }
  T prev_lambda() const { return prev_lambda_; }
  T bad_factor() const { return bad_factor_; }
  bool rebuild() const { return rebuild_; }
  int dims() const { return n_; }

  std::vector<T> H, g;

 private:
  const Options opt;
  int n_;
  bool is_lm_ = true;
  Cost cost_;
  T lambda_ = 1e-4f, prev_lambda_ = 0, bad_factor_ = 2.0f;  // lm.h:191-193
  bool rebuild_ = true;                                      // lm.h:194
};

// Per-iteration trace recorded by the oracle (not in the reference's Output; used to
// pin trajectories against the README trace and to generate golden fixtures).
struct TraceRow {
  double cost, lambda_used, dx_norm2;
  int good;
};

// ---------------------------------------------------------------------------
// Optimizer_<Solver> (optimizers/optimizer.h) restated.
// X is the parameter object; `plus(x, dx, sign)` applies x ⊞= sign*dx
// (traits.h:184-190 Euclidean; 3rdparty/traits/sophus.h:24-26 SE3 right-exp).
// ---------------------------------------------------------------------------
template <typename T>
class Optimizer {
 public:
  Optimizer(const Options& o, int n) : opt(o), solver(o, n), n_(n) {}

  // optimizer.h:331-539.  Returns (good, has_dx); dx written to `dx`.
  template <typename X, typename Acc>
  std::pair<bool, bool> Step(const X& x, const Acc& acc, Output& out, std::vector<T>& dx) {
    const auto iter = out.num_iters;
    dx.assign(n_, T(0));
    if (n_ == 0) {  // optimizer.h:61-67 (dynamic dims == 0)
      out.stop_reason = kSkipped;
      return {false, false};
    }
    Cost cost(NAN, out.num_residuals);  // optimizer.h:352
    bool solver_failed = true;
    const uint8_t max_tries =
        opt.max_consec_failures > 0 ? std::max<uint8_t>(1, opt.max_consec_failures) : 255;  // :356-357
    const T lambda_before = solver.lambda();
    for (; out.num_consec_failures <= max_tries;) {  // :358
      if (solver.Build(x, acc)) {
        if (solver.Solve(dx.data())) solver_failed = false;
      }
      cost = solver.cost();
      if (solver_failed) {
        out.num_consec_failures++;
        out.num_failures++;
        if (cost.num_residuals == 0) {  // :374-377
          out.stop_reason = kSkipped;
          return {false, false};
        } else if (std::isnan(cost.cost) || std::isinf(cost.cost)) {  // :378-381
          out.stop_reason = kSystemHasNaNOrInf;
          return {false, false};
        } else if (opt.max_consec_failures > 0 &&
                   out.num_consec_failures >= opt.max_consec_failures) {  // :382-386
          if (out.final_cost.cost < std::numeric_limits<T>::max()) out.stop_reason = kMaxConsecNoDecr;
          break;
        }
        solver.FailedStep();  // :389
      } else {
        break;
      }
    }
    (void)lambda_before;
    if (solver_failed) {  // :396-399
      out.stop_reason = kSolverFailed;
      return {false, false};
    }
    const double err = cost.cost;
    if (std::isnan(err) || std::isinf(err)) {  // :405-409
      out.stop_reason = kSystemHasNaNOrInf;
      return {false, false};
    }
    T dx2T = 0;
    for (auto v : dx) dx2T += v * v;
    const double dx_norm2 = dx2T;  // :412
    const bool has_grad_norm2 = opt.min_grad_norm2 > 0.0f;  // :413-414 (no callbacks here)
    const double grad_norm2 = has_grad_norm2 ? double(solver.GradientSquaredNorm()) : 0.0;
    if (std::isnan(dx_norm2) || std::isinf(dx_norm2)) {  // :416-425
      out.stop_reason = kSystemHasNaNOrInf;
      return {false, false};
    }
    const double derr = err - out.final_cost.cost;   // :428
    const bool is_good_step = derr < T(0.0);         // :429
    const double rel_derr =                          // :431-434
        (out.final_cost.cost > FloatEpsilon<T>() && out.final_cost.cost < std::numeric_limits<T>::max())
            ? (out.final_cost.cost - err) / out.final_cost.cost
            : 0.0f;
    out.errs.push_back(err);                         // :436-438
    out.deltas2.push_back(dx_norm2);
    out.successes.push_back(is_good_step);
    trace.push_back({err, double(solver.lambda()), dx_norm2, is_good_step ? 1 : 0});

    if (is_good_step || iter == 0) {                 // :441-446
      if (iter > 0) solver.GoodStep(opt.use_step_quality_approx ? T(rel_derr) : T(0.0f));
      out.num_consec_failures = 0;
      out.final_cost = cost;
      out.final_rerr_dec = rel_derr;
    } else {                                         // :447-460
      solver.BadStep();
      out.num_failures++;
      out.num_consec_failures++;
      if (opt.max_consec_failures > 0 && out.num_consec_failures >= opt.max_consec_failures) {
        out.stop_reason = kMaxConsecNoDecr;
        return {false, false};
      }
      if (opt.max_total_failures > 0 && out.num_failures >= opt.max_total_failures) {
        out.stop_reason = kMaxNoDecr;
        return {false, false};
      }
    }
    // :519-534 stop tests, fixed priority
    if (opt.min_error > 0 && err < opt.min_error) out.stop_reason = kMinError;
    else if (opt.min_rerr_dec > 0 && rel_derr > 0.0 && rel_derr < opt.min_rerr_dec) out.stop_reason = kMinRelError;
    else if (opt.min_step_norm2 > 0 && dx_norm2 < opt.min_step_norm2) out.stop_reason = kMinDeltaNorm;
    else if (opt.min_grad_norm2 > 0 && grad_norm2 < opt.min_grad_norm2) out.stop_reason = kMinGradNorm;
    return {is_good_step, true};
  }

  // optimizer.h:242-327
  template <typename X, typename Acc, typename Plus>
  Output OptimizeAcc(X& x, const Acc& acc, const Plus& plus, int max_iters = -1) {
    Output out;
    trace.clear();
    if (max_iters < 0) max_iters = opt.max_iters;  // :248
    max_iters++;                                   // :249
    if (opt.check_final_cost) max_iters++;         // :250
    std::vector<T> last_dx, dx;
    bool has_last_dx = false;
    bool last_was_success = true;                  // :263
    for (int iter = 0; iter < max_iters; ++iter) { // :266
      const auto st = Step(x, acc, out, dx);
      const bool success = st.first, has_dx = st.second;
      bool eval_only = false;
      if (success) {                               // :271-279
        plus(x, dx, T(1));
        last_dx = dx; has_last_dx = true;
        last_was_success = true;
        if (opt.check_final_cost && iter + 1 == max_iters) eval_only = true;
      } else {                                     // :281-297
        if (has_last_dx) {
          plus(x, last_dx, T(-1));
          has_last_dx = false;
        } else if (has_dx) {
          plus(x, dx, T(1));
          last_dx = dx; has_last_dx = true;
        }
        eval_only = (last_was_success == false);
        last_was_success = false;
      }
      solver.Rebuild(!eval_only);                  // :299
      out.num_iters++;                             // :307
      if (out.stop_reason != kNone) break;         // :309
    }
    if (opt.save_last) {                           // :313-316
      const auto Hh = solver.Hessian();
      out.final_hessian.assign(Hh.begin(), Hh.end());
    }
    if (out.stop_reason == kNone && out.num_iters >= max_iters) out.stop_reason = kMaxIters;  // :320-321
    return out;
  }

  const Options opt;
  Solver<T> solver;
  std::vector<TraceRow> trace;

 private:
  int n_;
};

// Euclidean PlusEq on contiguous scalars (traits.h:184-190).
template <typename T>
struct EuclidPlus {
  void operator()(std::vector<T>& x, const std::vector<T>& dx, T sign) const {
    for (size_t i = 0; i < x.size(); ++i) x[i] += sign * dx[i];
  }
};

// ---------------------------------------------------------------------------
// The AD bridge's meaning of (grad, H, cost) for a residual VECTOR
// (diff/optimize_autodiff.h:123-164): given residuals r (m) and Jacobian J
// (m×n, row-major here), grad = Jᵀ r (:151), H = Jᵀ J FULL matrix (:156),
// cost = (‖r‖², m) (:164).  Scalar residual: (:109-121) cost = (r², 1).
// All arithmetic in T like Eigen's Matrix<T> products.
// ---------------------------------------------------------------------------
template <typename T>
inline Cost AccumulateFromJ(int m, int n, const T* r, const T* J_rowmajor, T* g, T* H) {
  if (g) {
    for (int j = 0; j < n; ++j) {
      T s = 0;
      for (int i = 0; i < m; ++i) s += J_rowmajor[size_t(i) * n + j] * r[i];
      g[j] = s;
    }
    if (H) {
      for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) {
          T s = 0;
          for (int i = 0; i < m; ++i) s += J_rowmajor[size_t(i) * n + a] * J_rowmajor[size_t(i) * n + b];
          H[size_t(b) * n + a] = s;
        }
    }
  }
  T c = 0;
  for (int i = 0; i < m; ++i) c += r[i] * r[i];
  return Cost(double(c), m);
}

}  // namespace oracle
