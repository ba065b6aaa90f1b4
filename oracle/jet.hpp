// TEST INFRASTRUCTURE ONLY.  Host restatement of the reference's vendored dual numbers, ceres::Jet<T, N>
// (include/tinyopt/3rdparty/ceres/jet.h:216-301; arithmetic :304-430; sqrt :643, cos :651, sin :664, atan2 :1223),
// as far as the oracle needs them: it differentiates residual functions of a pose exactly the way
// OptimizeWithAutoDiff does (include/tinyopt/diff/optimize_autodiff.h:48-77: Jets seeded on the tangent at 0).
#pragma once
#include <array>
#include <cmath>

namespace oracle {

template <typename T, int N>
struct Jet {
  T a{};                  // jet.h:293
  std::array<T, N> v{};   // jet.h:296
  Jet() = default;
  Jet(T value) : a(value) {}                              // jet.h:232
  Jet(T value, int k) : a(value) { v[k] = T(1); }         // jet.h:238
};

#define ORACLE_JET template <typename T, int N> inline
ORACLE_JET Jet<T, N> operator-(const Jet<T, N>& f) { Jet<T, N> r; r.a = -f.a; for (int i = 0; i < N; ++i) r.v[i] = -f.v[i]; return r; }  // :314
ORACLE_JET Jet<T, N> operator+(const Jet<T, N>& f, const Jet<T, N>& g) { Jet<T, N> r; r.a = f.a + g.a; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] + g.v[i]; return r; }  // :320
ORACLE_JET Jet<T, N> operator+(const Jet<T, N>& f, T s) { Jet<T, N> r = f; r.a += s; return r; }      // :332
ORACLE_JET Jet<T, N> operator+(T s, const Jet<T, N>& f) { Jet<T, N> r = f; r.a += s; return r; }      // :338
ORACLE_JET Jet<T, N> operator-(const Jet<T, N>& f, const Jet<T, N>& g) { Jet<T, N> r; r.a = f.a - g.a; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] - g.v[i]; return r; }  // :344
ORACLE_JET Jet<T, N> operator-(const Jet<T, N>& f, T s) { Jet<T, N> r = f; r.a -= s; return r; }      // :356
ORACLE_JET Jet<T, N> operator-(T s, const Jet<T, N>& f) { Jet<T, N> r = -f; r.a += s; return r; }     // :362
ORACLE_JET Jet<T, N> operator*(const Jet<T, N>& f, const Jet<T, N>& g) {                              // :368
  Jet<T, N> r; r.a = f.a * g.a; for (int i = 0; i < N; ++i) r.v[i] = f.a * g.v[i] + f.v[i] * g.a; return r;
}
ORACLE_JET Jet<T, N> operator*(const Jet<T, N>& f, T s) { Jet<T, N> r; r.a = f.a * s; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] * s; return r; }  // :380
ORACLE_JET Jet<T, N> operator*(T s, const Jet<T, N>& f) { return f * s; }                             // :386
ORACLE_JET Jet<T, N> operator/(const Jet<T, N>& f, const Jet<T, N>& g) {                              // :392-410
  const T gi = T(1.0) / g.a, q = f.a * gi;
  Jet<T, N> r; r.a = q; for (int i = 0; i < N; ++i) r.v[i] = (f.v[i] - q * g.v[i]) * gi; return r;
}
ORACLE_JET Jet<T, N> operator/(T s, const Jet<T, N>& g) { const T k = -s / (g.a * g.a); Jet<T, N> r; r.a = s / g.a; for (int i = 0; i < N; ++i) r.v[i] = g.v[i] * k; return r; }  // :413
ORACLE_JET Jet<T, N> operator/(const Jet<T, N>& f, T s) { const T si = T(1.0) / s; return f * si; }   // :420
ORACLE_JET bool operator<(const Jet<T, N>& f, T s) { return f.a < s; }                                // :426-460 (scalar part)
ORACLE_JET bool operator>(const Jet<T, N>& f, T s) { return f.a > s; }
ORACLE_JET Jet<T, N> chain(const Jet<T, N>& f, T fa, T dfa) { Jet<T, N> r; r.a = fa; for (int i = 0; i < N; ++i) r.v[i] = dfa * f.v[i]; return r; }
ORACLE_JET Jet<T, N> sqrt(const Jet<T, N>& f) { const T s = std::sqrt(f.a); return chain(f, s, T(1.0) / (T(2.0) * s)); }   // :643
ORACLE_JET Jet<T, N> cos(const Jet<T, N>& f) { return chain(f, std::cos(f.a), -std::sin(f.a)); }                           // :651
ORACLE_JET Jet<T, N> sin(const Jet<T, N>& f) { return chain(f, std::sin(f.a), std::cos(f.a)); }                            // :664
ORACLE_JET Jet<T, N> atan2(const Jet<T, N>& g, const Jet<T, N>& f) {                                                       // :1223
  const T tmp = T(1.0) / (f.a * f.a + g.a * g.a);
  Jet<T, N> r; r.a = std::atan2(g.a, f.a);
  for (int i = 0; i < N; ++i) r.v[i] = tmp * (-g.a * f.v[i] + f.a * g.v[i]);
  return r;
}
#undef ORACLE_JET
// scalar part of a plain number / of a Jet (branch decisions are taken on it, jet.h:426-460)
template <typename T> inline T scalar_part(const T& x) { return x; }
template <typename T, int N> inline T scalar_part(const Jet<T, N>& x) { return x.a; }

}  // namespace oracle
