// TEST INFRASTRUCTURE ONLY.  The analytic test functions of the reference's optimizer tests, as Accumulate
// callbacks `Cost acc(x, g, H)` (g == nullptr: cost only; H col-major n x n, ASSIGNED):
//   kRosenbrock  tests/optimize_easy.cpp:35-79    manual callback, exact (Newton) Hessian, scalar cost
//   kPlateau     tests/optimize_easy.cpp:88-144   Easom-like dip, exact Hessian (indefinite on the plateau)
//   kPowell      tests/optimize_easy.cpp:153-221  Powell singular, exact Hessian (singular at the solution)
//   kBeale       tests/optimize_hard.cpp:34-63    residual vector (3), J^T J / J^T r as the AD bridge folds it
//   kHimmelblau  tests/optimize_hard.cpp:72-102   residual vector (2)
//   kXMinus2     tests/basic.cpp:41-54,72-87      scalar x - 2: grad = res, H = 1, cost = |res| (LM -> kMinDeltaNorm, GN -> kMinError)
// One definition shared by oracle/pin_reference_tests.cpp (the reference's own starts, options and known answers)
// and oracle_testfn_lm (batches of starts, for the GPU parity tests of the LM state machine's bad-step,
// failed-solve and rollback branches).
#pragma once
#include <cmath>
#include <vector>

#include "lm_oracle.hpp"

namespace oracle {
namespace testfn {

enum Id { kRosenbrock = 0, kPlateau = 1, kPowell = 2, kBeale = 3, kHimmelblau = 4, kXMinus2 = 5 };
inline int dims(int fn) { return fn == kPowell ? 4 : (fn == kXMinus2 ? 1 : 2); }

template <typename T>
inline Cost accumulate(int fn, const std::vector<T>& v, T* g, T* Hc) {
  switch (fn) {
    case kRosenbrock: {
      const T xv = v[0], yv = v[1];
      const T t1 = T(1.0) - xv, t2 = yv - xv * xv;
      if (g) {
        g[0] = T(-2.0) * t1 - T(400.0) * xv * t2;
        g[1] = T(200.0) * t2;
        Hc[0] = T(2.0) - T(400.0) * yv + T(1200.0) * xv * xv;
        Hc[2] = T(-400.0) * xv;  // (0,1) col-major
        Hc[1] = T(-400.0) * xv;  // (1,0)
        Hc[3] = T(200.0);
      }
      return Cost(double(t1 * t1 + T(100.0) * t2 * t2));
    }
    case kPlateau: {
      const T PI = T(std::acos(-1.0));
      const T dx = v[0] - PI, dy = v[1] - PI;
      const T ex = std::exp(-(dx * dx + dy * dy));
      const T cx = std::cos(v[0]), cy = std::cos(v[1]), sx = std::sin(v[0]), sy = std::sin(v[1]);
      const T cost = T(1.0) - (cx * cy * ex);
      if (g) {
        g[0] = cy * ex * (sx + T(2.0) * dx * cx);
        g[1] = cx * ex * (sy + T(2.0) * dy * cy);
        Hc[0] = cy * ex * (cx - T(4.0) * dx * sx + (T(2.0) - T(4.0) * dx * dx) * cx);
        Hc[3] = cx * ex * (cy - T(4.0) * dy * sy + (T(2.0) - T(4.0) * dy * dy) * cy);
        Hc[2] = ex * (sx + T(2.0) * dx * cx) * (sy + T(2.0) * dy * cy);
        Hc[1] = Hc[2];
      }
      return Cost(double(cost));
    }
    case kPowell: {
      const T x1 = v[0], x2 = v[1], x3 = v[2], x4 = v[3];
      const T t1 = x1 + T(10.0) * x2, t2 = x3 - x4, t3 = x2 - T(2.0) * x3, t4 = x1 - x4;
      if (g) {
        auto H = [&](int r, int c) -> T& { return Hc[c * 4 + r]; };
        g[0] = T(2.0) * t1 + T(40.0) * T(std::pow(t4, 3));
        g[1] = T(20.0) * t1 + T(4.0) * T(std::pow(t3, 3));
        g[2] = T(10.0) * t2 - T(8.0) * T(std::pow(t3, 3));
        g[3] = T(-10.0) * t2 - T(40.0) * T(std::pow(t4, 3));
        for (int i = 0; i < 16; ++i) Hc[i] = 0;
        H(0, 0) = T(2.0); H(0, 1) = T(20.0); H(1, 0) = T(20.0); H(1, 1) = T(200.0);
        H(2, 2) += T(10.0); H(2, 3) += T(-10.0); H(3, 2) += T(-10.0); H(3, 3) += T(10.0);
        const T d3 = T(12.0) * t3 * t3;
        H(1, 1) += d3; H(1, 2) += T(-2.0) * d3; H(2, 1) += T(-2.0) * d3; H(2, 2) += T(4.0) * d3;
        const T d4 = T(120.0) * t4 * t4;
        H(0, 0) += d4; H(0, 3) += -d4; H(3, 0) += -d4; H(3, 3) += d4;
      }
      return Cost(double(t1 * t1 + T(5.0) * t2 * t2 + T(std::pow(t3, 4)) + T(std::pow(t4, 4)) * T(10.0)));
    }
    case kXMinus2: {
      const T res = v[0] - T(2);
      if (g) { Hc[0] = T(1); g[0] = res; }
      return Cost(double(std::abs(res)));
    }
    case kBeale: {
      const T xv = v[0], yv = v[1];
      const T r[3] = {T(1.5) - xv + xv * yv, T(2.25) - xv + xv * yv * yv, T(2.625) - xv + xv * yv * yv * yv};
      const T J[6] = {T(-1) + yv, xv, T(-1) + yv * yv, T(2) * xv * yv, T(-1) + yv * yv * yv, T(3) * xv * yv * yv};
      return AccumulateFromJ<T>(3, 2, r, J, g, Hc);
    }
    default: {  // kHimmelblau
      const T r[2] = {v[0] * v[0] + v[1] - T(11.0), v[0] + v[1] * v[1] - T(7.0)};
      const T J[4] = {T(2) * v[0], T(1), T(1), T(2) * v[1]};
      return AccumulateFromJ<T>(2, 2, r, J, g, Hc);
    }
  }
}

template <typename T>
struct Acc {
  int fn;
  Cost operator()(const std::vector<T>& v, T* g, T* H) const { return accumulate<T>(fn, v, g, H); }
};

// the reference test's own options for each function
inline Options reference_options(int fn) {
  Options o;
  switch (fn) {
    case kRosenbrock: o.max_iters = 200; o.min_rerr_dec = 0; o.max_consec_failures = 20; break;
    case kPlateau: o.damping_init = 1e-6f; break;
    case kPowell: o.max_iters = 200; o.max_consec_failures = 0; o.min_error = 1e-30f; o.min_rerr_dec = 1e-30f; o.damping_init = 1e-1f; break;
    case kBeale: o.max_iters = 200; o.max_consec_failures = 0; o.min_error = 1e-30f; o.damping_init = 1e-3f; break;
    default: o.max_iters = 200; o.max_consec_failures = 0; o.min_error = 1e-30f; o.damping_init = 1e-4f; break;
  }
  return o;
}

}  // namespace testfn
}  // namespace oracle
