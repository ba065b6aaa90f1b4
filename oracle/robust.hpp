// TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's M-estimators
// (include/tinyopt/losses/robust_norms.h:32-316): each takes a squared norm n2 and a squared threshold th2
// and returns the robust loss `l` plus the scale `s` = d l / d n2 that multiplies the Jacobian / gradient
// ("export_jac = true" form, docs/API.md:402-406).  Pinned by oracle/pin_reference_tests.cpp against the
// closed forms and derivative checks of tests/robust_norms.cpp:53-115.
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>

namespace oracle {
namespace robust {

enum Kind { kL2 = 0, kTruncated = 1, kHuber = 2, kTukey = 3, kArctan = 4, kCauchy = 5, kGemanMcClure = 6, kBlakeZisserman = 7 };

template <typename T>
struct LossScale { T l, s; };

template <typename T>
inline LossScale<T> Truncated(T n2, T th2) {  // robust_norms.h:36-57
  return n2 <= th2 ? LossScale<T>{n2, T(1)} : LossScale<T>{th2, T(0)};
}
template <typename T>
inline LossScale<T> Huber(T n2, T th2) {  // robust_norms.h:73-105
  if (n2 <= th2) return {n2, T(1)};
  const T th = std::sqrt(th2), n = std::sqrt(n2);
  return {T(2.0) * th * n - th2, std::max<T>(std::numeric_limits<T>::min(), th / n)};
}
template <typename T>
inline LossScale<T> Tukey(T n2, T th2) {  // robust_norms.h:122-152
  if (n2 <= th2) {
    const T s = T(1.0) - n2 / th2, s2 = s * s;
    return {th2 * (T(1.0) - s2 * s), T(3.0) * (th2 - n2) * (th2 - n2) / (th2 * th2)};
  }
  return {th2, T(0)};
}
template <typename T>
inline LossScale<T> Arctan(T n2, T th2) {  // robust_norms.h:168-190
  const T th = std::sqrt(th2);
  const T tmp = n2 * n2 / th2;
  return {th * std::atan2(n2, th), std::max<T>(std::numeric_limits<T>::min(), T(1.0) / (tmp + T(1.0)))};
}
template <typename T>
inline LossScale<T> Cauchy(T n2, T th2) {  // robust_norms.h:207-228
  const T s = T(1.0) + n2 / th2;
  return {th2 * std::log(s), std::max<T>(std::numeric_limits<T>::min(), T(1.0) / s)};
}
template <typename T>
inline LossScale<T> GemanMcClure(T n2, T th2) {  // robust_norms.h:245-265
  const T e = n2 + th2;
  return {n2 / e, th2 / (e * e)};
}
template <typename T>
inline LossScale<T> BlakeZisserman(T n2, T th2) {  // robust_norms.h:282-303
  const T eps = std::exp(-th2);
  return {-std::log(std::exp(-n2) + eps), T(1.0) / (eps * std::exp(n2) + T(1.0))};
}

template <typename T>
inline LossScale<T> Apply(int kind, T n2, T th2) {
  switch (kind) {
    case kTruncated: return Truncated(n2, th2);
    case kHuber: return Huber(n2, th2);
    case kTukey: return Tukey(n2, th2);
    case kArctan: return Arctan(n2, th2);
    case kCauchy: return Cauchy(n2, th2);
    case kGemanMcClure: return GemanMcClure(n2, th2);
    case kBlakeZisserman: return BlakeZisserman(n2, th2);
    default: return {n2, T(1)};
  }
}

}  // namespace robust
}  // namespace oracle
