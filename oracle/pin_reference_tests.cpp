// TEST INFRASTRUCTURE ONLY.  Pins oracle/lm_oracle.hpp against every known-answer test the
// reference holds for the LM hot path.  Each case restates the PROBLEM of a
// reference test (cost function, start, options) and asserts exactly what that test asserts
// (final x, StopReason, iteration window, Succeeded/Converged).  Where the reference test
// uses automatic differentiation, the analytic Jacobian is supplied instead (forward-mode AD
// is exact, so J is identical to rounding) and folded through AccumulateFromJ, which restates
// diff/optimize_autodiff.h:123-164.
//
// Build+run:  make -C oracle pin     (exit code != 0 on any failure)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "lm_oracle.hpp"
#include "robust.hpp"
#include "se3.hpp"
#include "testfns.hpp"

using namespace oracle;

static int g_fail = 0, g_pass = 0;
#define CHECK(cond)                                                         \
  do {                                                                      \
    if (cond) { ++g_pass; }                                                 \
    else { ++g_fail; std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); } \
  } while (0)
#define CHECK_NEAR(a, b, tol) CHECK(std::abs(double(a) - double(b)) <= (tol))

template <typename T>
using Vec = std::vector<T>;

// ---- tests/sqrt2.cpp:30-56  TestSqrt2 (float, manual acc returning r*r => Cost(v,1)) ----
static void sqrt2_manual(float x0) {
  Options o;  // tests/sqrt2.cpp:22-28
  o.max_iters = 20;
  o.max_consec_failures = 0;
  Vec<float> x{x0};
  auto acc = [](const Vec<float>& x, float* g, float* H) {
    float res = x[0] * x[0] - 2;
    float J = 2 * x[0];
    if (g) { g[0] = J * res; H[0] = J * J; }
    return Cost(double(res * res));
  };
  Optimizer<float> opt(o, 1);
  Output out = opt.OptimizeAcc(x, acc, EuclidPlus<float>());
  CHECK(out.Succeeded());
  CHECK(out.Converged());
  CHECK_NEAR(std::abs(x[0]), std::sqrt(2.0), 1e-5);
}

// ---- tests/sqrt2.cpp:58-70  TestSqrt2Jet (double AD scalar residual, downscale_by_2) ----
// AD scalar path: optimize_autodiff.h:109-121: grad = J*r, H = J*J, cost = r*r (1 residual).
static void sqrt2_jet(double x0, bool downscale) {
  Options o;
  o.max_iters = 20;
  o.max_consec_failures = 0;
  o.use_squared_norm = true;
  o.downscale_by_2 = downscale;
  Vec<double> x{x0};
  auto acc = [](const Vec<double>& x, double* g, double* H) {
    double r = x[0] * x[0] - 2.0, J = 2 * x[0];
    if (g) { g[0] = J * r; H[0] = J * J; }
    return Cost(r * r);
  };
  Optimizer<double> opt(o, 1);
  Output out = opt.OptimizeAcc(x, acc, EuclidPlus<double>());
  CHECK(out.Succeeded());
  CHECK(out.Converged());
  CHECK_NEAR(std::abs(x[0]), std::sqrt(2.0), 1e-5);
}

// ---- tests/sqrt2.cpp:72-92  TestSqrt2Jet2 (2 residuals: r, 0.1 r) ----
static void sqrt2_jet2(double x0) {
  Options o;
  o.max_iters = 20;
  o.max_consec_failures = 0;
  Vec<double> x{x0};
  auto acc = [](const Vec<double>& x, double* g, double* H) {
    double r[2] = {x[0] * x[0] - 2.0, 0.1 * (x[0] * x[0] - 2.0)};
    double J[2] = {2 * x[0], 0.1 * 2 * x[0]};
    return AccumulateFromJ<double>(2, 1, r, J, g, H);
  };
  Optimizer<double> opt(o, 1);
  Output out = opt.OptimizeAcc(x, acc, EuclidPlus<double>());
  CHECK(out.Succeeded());
  CHECK(out.Converged());
  CHECK_NEAR(std::abs(x[0]), std::sqrt(2.0), 1e-5);
}

// ---- README.md:91-96 trace: x = 1 -> 1.49995 -> 1.41667 -> 1.41422 -> 1.41421,
//      |dx| = 5.00e-01, 8.33e-02, 2.45e-03 (the only per-iteration numbers in the repo) ----
static void readme_trace() {
  Options o;  // defaults
  Vec<double> x{1.0};
  Vec<double> xs;
  auto acc = [&](const Vec<double>& x, double* g, double* H) {
    double r = x[0] * x[0] - 2.0, J = 2 * x[0];
    if (g) { g[0] = J * r; H[0] = J * J; xs.push_back(x[0]); }
    return Cost(r * r);
  };
  Optimizer<double> opt(o, 1);
  Output out = opt.OptimizeAcc(x, acc, EuclidPlus<double>());
  CHECK(out.Succeeded());
  CHECK(xs.size() >= 5);
  if (xs.size() >= 5) {
    CHECK_NEAR(xs[0], 1.0, 0);
    CHECK_NEAR(xs[1], 1.49995, 5e-6);
    CHECK_NEAR(xs[2], 1.41667, 5e-6);
    CHECK_NEAR(xs[3], 1.41422, 5e-6);
    CHECK_NEAR(xs[4], 1.41421, 5e-6);
  }
  CHECK_NEAR(std::sqrt(out.deltas2[0]), 5.00e-1, 5e-4);
  CHECK_NEAR(std::sqrt(out.deltas2[1]), 8.33e-2, 5e-5);
  // README prints 2.45e-03 (older revision, log format differs); exact Newton algebra from the
  // printed x's gives 1.41667-1.41422 = 2.45e-3..2.46e-3, so pin to 1e-5 absolute.
  CHECK_NEAR(std::sqrt(out.deltas2[2]), 2.45e-3, 1e-5);
  // lambda sequence: 1e-4, then /3 per good step (README λ column 1.00e-04, 3.33e-05, 1.11e-05 ...)
  CHECK_NEAR(opt.trace[0].lambda_used, 1e-4, 1e-9);
}

// ---- tests/basic.cpp:22-54  "Normal Test Case LM": r = x-2, returns |r| ----
static Cost xm2_acc(const Vec<double>& x, double* g, double* H) {
  double res = x[0] - 2;
  if (g) { H[0] = 1; g[0] = res; }
  return Cost(std::abs(res));
}
static void basic_success() {
  {  // LM default options -> kMinDeltaNorm, 2..5 iters, final_cost<1e-5, H(0,0)>0 (basic.cpp:22-54)
    Options o;
    Vec<double> x{1.0};
    Optimizer<double> opt(o, 1);
    Output out = opt.OptimizeAcc(x, xm2_acc, EuclidPlus<double>());
    CHECK(out.Succeeded());
    CHECK(out.num_iters >= 2 && out.num_iters <= 5);
    CHECK(out.final_cost.cost < 1e-5);
    CHECK(out.Converged());
    CHECK(out.errs.size() == size_t(out.num_iters));
    CHECK(out.successes.size() == out.errs.size());
    CHECK(out.deltas2.size() == out.errs.size());
    CHECK(!out.final_hessian.empty() && out.final_hessian[0] > 0);
    CHECK(out.stop_reason == kMinDeltaNorm);
  }
  {  // GN -> kMinError (basic.cpp:72-87)
    Options o;
    o.solver_type = Options::GaussNewton;
    Vec<double> x{1.0};
    Optimizer<double> opt(o, 1);
    Output out = opt.OptimizeAcc(x, xm2_acc, EuclidPlus<double>());
    CHECK(out.Succeeded());
    CHECK(out.num_iters >= 2 && out.num_iters <= 5);
    CHECK(out.final_cost.cost < 1e-5);
    CHECK(out.Converged());
    CHECK(out.stop_reason == kMinError);
  }
  {  // GN min_error=1e-2 -> kMinError (basic.cpp:107-124)
    Options o;
    o.min_error = 1e-2f;
    o.solver_type = Options::GaussNewton;
    Vec<double> x{1.0};
    Optimizer<double> opt(o, 1);
    Output out = opt.OptimizeAcc(x, xm2_acc, EuclidPlus<double>());
    CHECK(out.Succeeded() && out.Converged());
    CHECK(out.num_iters >= 2 && out.num_iters <= 5);
    CHECK(out.stop_reason == kMinError);
  }
}

// ---- tests/basic.cpp:147-258 failure paths: empty history + exact stop reasons ----
static void failure_checks(const Output& out, int expected, int max_iters = 1) {
  CHECK(!out.Succeeded());
  CHECK(!out.Converged());
  CHECK(out.num_iters <= max_iters);
  CHECK(out.errs.empty());
  CHECK(out.successes.empty());
  CHECK(out.deltas2.empty());
  CHECK(out.stop_reason == expected);
}
static void basic_failures() {
  const double inf = std::numeric_limits<double>::infinity();
  {  // NaN in grad (basic.cpp:160-172)
    auto acc = [](const Vec<double>& x, double* g, double* H) {
      double res = x[0] - 2;
      if (g) { H[0] = 1; g[0] = NAN; }
      return Cost(std::abs(res));
    };
    Vec<double> x{1};
    Optimizer<double> opt(Options(), 1);
    failure_checks(opt.OptimizeAcc(x, acc, EuclidPlus<double>()), kSystemHasNaNOrInf);
  }
  {  // Inf in grad (basic.cpp:174-186)
    auto acc = [&](const Vec<double>& x, double* g, double* H) {
      double res = x[0] - 2;
      if (g) { H[0] = 1; g[0] = inf; }
      return Cost(std::abs(res));
    };
    Vec<double> x{1};
    Optimizer<double> opt(Options(), 1);
    failure_checks(opt.OptimizeAcc(x, acc, EuclidPlus<double>()), kSystemHasNaNOrInf);
  }
  {  // Inf in res (basic.cpp:188-200)
    auto acc = [&](const Vec<double>& x, double* g, double* H) {
      double res = x[0] + inf;
      if (g) { H[0] = 1; g[0] = inf; }
      return Cost(std::abs(res));
    };
    Vec<double> x{1};
    Optimizer<double> opt(Options(), 1);
    failure_checks(opt.OptimizeAcc(x, acc, EuclidPlus<double>()), kSystemHasNaNOrInf);
  }
  {  // Inf cost (basic.cpp:202-214)
    auto acc = [&](const Vec<double>& x, double* g, double* H) {
      double res = x[0] + 1;
      if (g) { H[0] = 1; g[0] = res; }
      return Cost(inf);
    };
    Vec<double> x{1};
    Optimizer<double> opt(Options(), 1);
    failure_checks(opt.OptimizeAcc(x, acc, EuclidPlus<double>()), kSystemHasNaNOrInf);
  }
  {  // forgot H, GN + check_min_H_diag -> kSolverFailed within <=3 iters (basic.cpp:219-233)
    auto acc = [](const Vec<double>& x, double*, double*) { return Cost(std::abs(x[0] - 2)); };
    Options o;
    o.solver_type = Options::GaussNewton;
    o.check_min_H_diag = 1e-7f;
    Vec<double> x{1};
    Optimizer<double> opt(o, 1);
    failure_checks(opt.OptimizeAcc(x, acc, EuclidPlus<double>()), kSolverFailed, 3);
  }
  {  // no residuals -> kSkipped (basic.cpp:234-243): VecX() => Cost(0, 0)
    auto acc = [](const Vec<double>&, double*, double*) { return Cost(0.0, 0); };
    Vec<double> x{1};
    Optimizer<double> opt(Options(), 1);
    failure_checks(opt.OptimizeAcc(x, acc, EuclidPlus<double>()), kSkipped);
  }
  {  // empty x -> kSkipped (basic.cpp:244-258)
    auto acc = [](const Vec<float>&, float*, float*) { return Cost(1.0); };
    Vec<float> x;
    Optimizer<float> opt(Options(), 0);
    failure_checks(opt.OptimizeAcc(x, acc, EuclidPlus<float>()), kSkipped);
  }
}

// ---- tests/solvers.cpp:20-45 one-shot Build+Solve on r = x - y: dx ≈ y ±1e-2;
//      tests/solvers.cpp:74-110 skip-rebuild call counts ----
static void solvers() {
  for (int lm = 0; lm < 2; ++lm) {
    Options o;
    o.solver_type = lm ? Options::LevenbergMarquardt : Options::GaussNewton;
    Solver<double> s(o, 2);
    Vec<double> x{0, 0};
    const double y[2] = {4, 5};
    auto acc = [&](const Vec<double>& x, double* g, double* H) {
      double r[2] = {x[0] - y[0], x[1] - y[1]};
      double J[4] = {1, 0, 0, 1};
      return AccumulateFromJ<double>(2, 2, r, J, g, H);
    };
    CHECK(s.Build(x, acc));
    double dx[2];
    CHECK(s.Solve(dx));
    CHECK_NEAR(dx[0], 4, 1e-2);
    CHECK_NEAR(dx[1], 5, 1e-2);
  }
  {
    Solver<double> s(Options(), 2);
    Vec<double> x{0, 0};
    const double y[2] = {4, 5};
    int num_grad_updates = 0;
    auto acc = [&](const Vec<double>& x, double* g, double* H) {
      double r[2] = {x[0] - y[0], x[1] - y[1]};
      if (g) { g[0] = r[0]; g[1] = r[1]; H[0] = 1; H[1] = 0; H[2] = 0; H[3] = 1; num_grad_updates++; }
      return Cost(r[0] * r[0] + r[1] * r[1], 2);
    };
    CHECK(s.Build(x, acc));
    CHECK(num_grad_updates == 1);
    s.Rebuild(false);
    CHECK(s.Build(x, acc));
    CHECK(num_grad_updates == 1);
    double dx[2];
    CHECK(s.Solve(dx));
    CHECK_NEAR(dx[0], 4, 1e-2);
    CHECK_NEAR(dx[1], 5, 1e-2);
  }
}

// ---- tests/optimize_easy.cpp:35-79 Rosenbrock with user Hessian (bad-step branches) ----
static void rosenbrock() {  // tests/optimize_easy.cpp:35-79
  Vec<double> x{-1.2, 1.0};
  Optimizer<double> opt(testfn::reference_options(testfn::kRosenbrock), 2);
  Output out = opt.OptimizeAcc(x, testfn::Acc<double>{testfn::kRosenbrock}, EuclidPlus<double>());
  CHECK(out.Succeeded());
  CHECK(out.Converged());
  CHECK_NEAR(x[0], 1.0, 1e-5);
  CHECK_NEAR(x[1], 1.0, 1e-5);
  // must have exercised the bad-step branch
  int bad = 0;
  for (auto s : out.successes) bad += !s;
  CHECK(bad > 0);
}

// ---- tests/optimize_easy.cpp:88-144 Easom-like plateau ----
static void plateau() {
  const double PI = std::acos(-1.0);
  Vec<double> x{3.0, 3.0};
  Optimizer<double> opt(testfn::reference_options(testfn::kPlateau), 2);
  Output out = opt.OptimizeAcc(x, testfn::Acc<double>{testfn::kPlateau}, EuclidPlus<double>());
  CHECK(out.Succeeded());
  CHECK_NEAR(x[0], PI, 1e-4);
  CHECK_NEAR(x[1], PI, 1e-4);
}

// ---- tests/optimize_easy.cpp:153-221 Powell singular ----
static void powell() {
  Vec<double> x{3.0, -1.0, 0.0, 1.0};
  Optimizer<double> opt(testfn::reference_options(testfn::kPowell), 4);
  Output out = opt.OptimizeAcc(x, testfn::Acc<double>{testfn::kPowell}, EuclidPlus<double>());
  CHECK(out.Succeeded());
  for (int i = 0; i < 4; ++i) CHECK(std::abs(x[i]) < 1e-3);
}

// ---- tests/optimize_hard.cpp:34-63 Beale (AD residual vector) ----
static void beale() {
  Vec<double> x{1.0, 1.0};
  Optimizer<double> opt(testfn::reference_options(testfn::kBeale), 2);
  Output out = opt.OptimizeAcc(x, testfn::Acc<double>{testfn::kBeale}, EuclidPlus<double>());
  CHECK(out.Succeeded());
  CHECK_NEAR(x[0], 3.0, 1e-4);
  CHECK_NEAR(x[1], 0.5, 1e-4);
}

// ---- tests/optimize_hard.cpp:72-102 Himmelblau ----
static void himmelblau() {
  Vec<double> x{3.5, 2.5};
  Optimizer<double> opt(testfn::reference_options(testfn::kHimmelblau), 2);
  (void)opt.OptimizeAcc(x, testfn::Acc<double>{testfn::kHimmelblau}, EuclidPlus<double>());
  CHECK_NEAR(x[0], 3.0, 1e-4);
  CHECK_NEAR(x[1], 2.0, 1e-4);
}

// ---- tests/circle.cpp:32-68 circle fit, lambda0 = 10 -> (2,7,2) ±1e-5.
//      Observations: 10 points on the circle + 1e-5 noise (tests/circle.cpp:20-30; seeded here). ----
static void circle() {
  const int n = 10;
  const float radius = 2, cx = 2, cy = 7;
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  std::vector<float> ox(n), oy(n);
  float angle = 0;
  const float pi = 3.14159265358979f;
  for (int i = 0; i < n; ++i) {
    ox[i] = cx + radius * cosf(angle) + 1e-5f * U(rng);
    oy[i] = cy + radius * sinf(angle) + 1e-5f * U(rng);
    angle += 2 * pi / (n - 1);
  }
  Vec<double> x{0, 0, 1};
  auto acc = [&](const Vec<double>& v, double* g, double* H) {
    std::vector<double> r(n), J(n * 3);
    for (int i = 0; i < n; ++i) {
      double dx = ox[i] - v[0], dy = oy[i] - v[1];
      r[i] = dx * dx + dy * dy - v[2] * v[2];
      J[i * 3 + 0] = -2 * dx;
      J[i * 3 + 1] = -2 * dy;
      J[i * 3 + 2] = -2 * v[2];
    }
    return AccumulateFromJ<double>(n, 3, r.data(), J.data(), g, H);
  };
  Options o;
  o.damping_init = 1e1;
  Optimizer<double> opt(o, 3);
  Output out = opt.OptimizeAcc(x, acc, EuclidPlus<double>());
  CHECK(out.Succeeded());
  CHECK_NEAR(x[0], cx, 1e-5);
  CHECK_NEAR(x[1], cy, 1e-5);
  CHECK_NEAR(std::abs(x[2]), radius, 1e-5);
}

// ---- tests/cov.cpp:20-47: Gaussian prior, manual acc returning res.norm(); covariance from the
//      final UNDAMPED Hessian recovers the prior stdevs to 1e-7 ----
static void cov_prior() {
  const double y[2] = {3.7, -8.1}, sd[2] = {4.2, 4.2};
  auto acc = [&](const Vec<double>& x, double* g, double* H) {
    double r[2] = {(x[0] - y[0]) / sd[0], (x[1] - y[1]) / sd[1]};  // mahalanobis.h:124-136
    if (g) {
      g[0] = r[0] / sd[0]; g[1] = r[1] / sd[1];                    // J = diag(1/sd); grad = J*res
      H[0] = 1 / (sd[0] * sd[0]); H[3] = 1 / (sd[1] * sd[1]);      // H.diagonal() = sd^-2
    }
    return Cost(std::sqrt(r[0] * r[0] + r[1] * r[1]));
  };
  Vec<double> x{0, 0};
  Optimizer<double> opt(Options(), 2);
  Output out = opt.OptimizeAcc(x, acc, EuclidPlus<double>());
  CHECK(out.Succeeded());
  CHECK(out.Converged());
  CHECK(out.final_hessian.size() == 4);
  if (out.final_hessian.size() == 4) {
    // InvCov of a diagonal H: C_ii = 1/H_ii  (math.h:41-57)
    CHECK_NEAR(std::sqrt(1.0 / out.final_hessian[0]), sd[0], 1e-7);
    CHECK_NEAR(std::sqrt(1.0 / out.final_hessian[3]), sd[1], 1e-7);
  }
}

// ---- tests/cov.cpp:91-146: prior with the general covariance Cy = [[10,2],[2,4]], whitened by Lt = chol(Cy^-1).U;
//      the covariance from the final Hessian equals Cy +-1e-5 ----
static void cov_prior_general() {
  const double Cy[4] = {10, 2, 2, 4};
  const double det = Cy[0] * Cy[3] - Cy[1] * Cy[2];
  const double I[4] = {Cy[3] / det, -Cy[1] / det, -Cy[2] / det, Cy[0] / det};   // information matrix
  // upper Cholesky factor U of I (I = U^T U)
  const double u00 = std::sqrt(I[0]), u01 = I[1] / u00, u11 = std::sqrt(I[3] - u01 * u01);
  const double y[2] = {1.3, -0.7}, U[4] = {u00, u01, 0, u11};
  auto acc = [&](const Vec<double>& x, double* g, double* H) {
    const double d0 = x[0] - y[0], d1 = x[1] - y[1];
    const double r[2] = {U[0] * d0 + U[1] * d1, U[3] * d1};     // mahalanobis.h:160-171 (UU * res)
    return AccumulateFromJ<double>(2, 2, r, U, g, H);
  };
  Vec<double> x{0, 0};
  Optimizer<double> opt(Options(), 2);
  Output out = opt.OptimizeAcc(x, acc, EuclidPlus<double>());
  CHECK(out.Succeeded());
  CHECK(out.Converged());
  CHECK_NEAR(x[0], y[0], 1e-8);
  CHECK_NEAR(x[1], y[1], 1e-8);
  CHECK(out.final_hessian.size() == 4);
  if (out.final_hessian.size() == 4) {
    const double* H = out.final_hessian.data();
    const double dh = H[0] * H[3] - H[1] * H[2];
    const double Cv[4] = {H[3] / dh, -H[1] / dh, -H[2] / dh, H[0] / dh};
    for (int i = 0; i < 4; ++i) CHECK_NEAR(Cv[i], Cy[i], 1e-5);
  }
}

// ---- tests/sophus.cpp:26-44: SE3 pose prior, residual log(prior_inv * x) differentiated by Jets over the right
//      perturbation; Succeeded && Converged && ||log(pose * prior_inv)|| < 1e-5.  Plus the pieces it stands on:
//      log(exp(xi)) == xi, and the Jet Jacobian against central differences of the right perturbation. ----
static void se3_pose_prior() {
  std::mt19937 rng(7);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  for (int trial = 0; trial < 8; ++trial) {
    se3::Pose<double> ident{};
    ident[0] = ident[4] = ident[8] = 1;
    // exp / log round trip (also through the small-angle branch)
    const double scale = trial == 0 ? 1e-3 : (trial == 1 ? 0.03 : 0.9);
    std::vector<double> xi(6);
    for (auto& v : xi) v = scale * U(rng);
    se3::Pose<double> T = ident;
    se3::plus_eq(T, xi, 1.0);
    double back[6];
    se3::se3_log<double, double>(T.data(), T.data() + 9, back);
    for (int i = 0; i < 6; ++i) CHECK_NEAR(back[i], xi[i], 1e-11);
    // the reference test
    std::vector<double> a(6), b(6);
    for (auto& v : a) v = 0.8 * U(rng);
    for (auto& v : b) v = 0.8 * U(rng);
    se3::Pose<double> prior_inv = ident, pose = ident;
    se3::plus_eq(prior_inv, a, 1.0);
    se3::plus_eq(pose, b, 1.0);
    se3::PosePriorAcc<double> acc{prior_inv.data()};
    {  // Jacobian check at the start pose
      double g[6], H[36];
      (void)acc(pose, g, H);
      const double eps = 1e-6;
      for (int k = 0; k < 6; ++k) {
        std::vector<double> d(6, 0.0);
        d[k] = eps;
        se3::Pose<double> pp = pose, pm = pose;
        se3::plus_eq(pp, d, 1.0);
        se3::plus_eq(pm, d, -1.0);
        const double cp = acc(pp, nullptr, nullptr).cost, cm = acc(pm, nullptr, nullptr).cost;
        CHECK_NEAR(0.5 * (cp - cm) / (2 * eps), g[k], 1e-6 * (1 + std::abs(g[k])));   // grad of 1/2 ||r||^2 = J^T r
      }
    }
    Optimizer<double> opt(Options(), 6);
    Output out = opt.OptimizeAcc(pose, acc, se3::Plus<double>());
    CHECK(out.Succeeded());
    CHECK(out.Converged());
    // (pose * prior_inv).log().norm() < 1e-5
    se3::Pose<double> prod;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j)
        prod[3 * i + j] = pose[3 * i] * prior_inv[j] + pose[3 * i + 1] * prior_inv[3 + j] + pose[3 * i + 2] * prior_inv[6 + j];
      prod[9 + i] = pose[3 * i] * prior_inv[9] + pose[3 * i + 1] * prior_inv[10] + pose[3 * i + 2] * prior_inv[11] + pose[9 + i];
    }
    double lg[6], nrm = 0;
    se3::se3_log<double, double>(prod.data(), prod.data() + 9, lg);
    for (double v : lg) nrm += v * v;
    CHECK(std::sqrt(nrm) < 1e-5);
  }
}

// ---- LDLT restatement: SPD solve accuracy, pivoting, and the reference's failure policy
//      (math.h:236: fail iff info()!=Success || !isPositive()) ----
static void ldlt_policy() {
  {  // SPD 3x3, known solution
    double A[9] = {4, 0, 0, 12, 37, 0, -16, -43, 98};  // upper stored col-major: A(0,1)=12, A(0,2)=-16, A(1,2)=-43
    double b[3] = {1, 2, 3}, x[3];
    CHECK(SolveLDLT<double>(3, A, b, x));
    // verify A x = b with the symmetric matrix
    double S[3][3] = {{4, 12, -16}, {12, 37, -43}, {-16, -43, 98}};
    for (int i = 0; i < 3; ++i) {
      double s = 0;
      for (int j = 0; j < 3; ++j) s += S[i][j] * x[j];
      CHECK_NEAR(s, b[i], 1e-9);
    }
  }
  {  // negative definite -> isPositive() false -> failure
    double A[4] = {-1, 0, 0, -2}, b[2] = {1, 1}, x[2];
    CHECK(!SolveLDLT<double>(2, A, b, x));
  }
  {  // indefinite -> failure
    double A[4] = {1, 0, 0, -2}, b[2] = {1, 1}, x[2];
    CHECK(!SolveLDLT<double>(2, A, b, x));
  }
  {  // semi-definite (zero pivot) passes isPositive(); pseudo-inverse zeroes that component
    double A[4] = {2, 0, 0, 0}, b[2] = {4, 0}, x[2];
    CHECK(SolveLDLT<double>(2, A, b, x));
    CHECK_NEAR(x[0], 2, 1e-15);
    CHECK_NEAR(x[1], 0, 0);
  }
  {  // all-zero H ("forgot to fill H"): ZeroSign, passes, dx = 0
    double A[4] = {0, 0, 0, 0}, b[2] = {1, 1}, x[2];
    CHECK(SolveLDLT<double>(2, A, b, x));
    CHECK_NEAR(x[0], 0, 0);
  }
}

// tests/robust_norms.cpp:53-115 — closed forms (LOSS_WRAPPER expected_code) and the derivative check the
// reference does with CalculateJac (here: central differences of the loss), margin 1e-5, th = 1.3;
// scalar n2 = 0.5, inlier 0.3, outlier 2.3^2; vector x = (.1,-.2,-.3,.4) with th = 1.3 (inlier) and 0.03 (outlier).
static double robust_expected(int kind, double n2, double th2) {
  const double th = std::sqrt(th2), n = std::sqrt(n2);
  switch (kind) {
    case oracle::robust::kTruncated: return n > th ? th2 : n2;
    case oracle::robust::kHuber: return n > th ? (2.0 * th * n - th2) : n2;
    case oracle::robust::kTukey: return n > th ? th2 : (th2 * (1.0 - std::pow(1.0 - n2 / th2, 3.0)));
    case oracle::robust::kArctan: return th * std::atan2(n2, th);
    case oracle::robust::kCauchy: return th2 * std::log(1.0 + n2 / th2);
    case oracle::robust::kGemanMcClure: return n2 / (n2 + th2);
    default: return -std::log(std::exp(-n2) + std::exp(-th2));
  }
}
static void robust_norms() {
  using namespace oracle::robust;
  for (int kind = kTruncated; kind <= kBlakeZisserman; ++kind) {
    const double th = 1.3, th2 = th * th;
    CHECK_NEAR(Apply(kind, 0.5, th2).l, robust_expected(kind, 0.5, th2), 1e-5);  // "Scalar"
    for (double n2 : {0.3, 2.3 * 2.3}) {                                          // "Scalar Inlier" / "Scalar Outlier"
      const auto ls = Apply(kind, n2, th2);
      CHECK_NEAR(ls.l, robust_expected(kind, n2, th2), 1e-5);
      const double h = 1e-6;
      const double fd = (Apply(kind, n2 + h, th2).l - Apply(kind, n2 - h, th2).l) / (2 * h);
      CHECK_NEAR(ls.s, fd, 1e-5);
    }
    const double x[4] = {.1, -0.2, -0.3, 0.4};
    for (double thv : {1.3, 0.03}) {  // "Vec Inlier" / "Vec Outlier": J = s * d(||x||^2)/dx = s * 2x
      const double t2 = thv * thv;
      double n2 = 0;
      for (double v : x) n2 += v * v;
      const auto ls = Apply(kind, n2, t2);
      CHECK_NEAR(ls.l, robust_expected(kind, n2, t2), 1e-5);
      for (int i = 0; i < 4; ++i) {
        const double h = 1e-6;
        double xp[4] = {x[0], x[1], x[2], x[3]}, xm[4] = {x[0], x[1], x[2], x[3]};
        xp[i] += h; xm[i] -= h;
        double np = 0, nm = 0;
        for (int k = 0; k < 4; ++k) { np += xp[k] * xp[k]; nm += xm[k] * xm[k]; }
        const double fd = (Apply(kind, np, t2).l - Apply(kind, nm, t2).l) / (2 * h);
        CHECK_NEAR(ls.s * 2 * x[i], fd, 1e-5);
      }
    }
  }
  // docs/API.md:399 — `Huber(y.squaredNorm(), 0.8)`: inlier below the squared threshold returns n2 itself
  CHECK(Huber(0.5, 0.8).l == 0.5 && Huber(0.5, 0.8).s == 1.0);
}

int main() {
  for (float x0 : {1.0f, -0.3f, 3.2f}) {  // tests/sqrt2.cpp:106-112
    sqrt2_manual(x0);
    sqrt2_jet(x0, true);
    sqrt2_jet2(x0);
    if (x0 > 0) sqrt2_jet(x0, false);
  }
  readme_trace();
  basic_success();
  basic_failures();
  solvers();
  rosenbrock();
  plateau();
  powell();
  beale();
  himmelblau();
  circle();
  cov_prior();
  cov_prior_general();
  se3_pose_prior();
  ldlt_policy();
  robust_norms();
  std::printf("pin_reference_tests: %d passed, %d failed\n", g_pass, g_fail);
  return g_fail ? 1 : 0;
}
