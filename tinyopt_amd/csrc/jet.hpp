// Device forward-mode dual numbers — the GPU counterpart of the reference's vendored ceres::Jet<T, N>
// (include/tinyopt/3rdparty/ceres/jet.h:216-301; arithmetic :304-430; functions :557-1340) that
// OptimizeWithAutoDiff (include/tinyopt/diff/optimize_autodiff.h:21-169) evaluates user residuals on.
//
// Same algebra, same formulas (each operator cites the reference line it restates); what differs is the
// storage: the infinitesimal part is a plain T[N] that lives in VGPRs (N is a compile-time constant; the
// reference's N = Dynamic heap vectors have no place in a kernel — wide parameter blocks are differentiated
// in chunks instead, JetRowModel in kernels.hpp).  With it a user writes only
// `r(x)` as a template over the scalar type, exactly like a tinyopt residual functor, and JetModel
// (kernels.hpp) turns it into the Accumulate contract on the device.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

namespace toa {

// keep the scalar overloads visible next to the Jet ones (functors are templates over the scalar type and
// call sin(t), sqrt(t), ... unqualified; the Jet overloads below would otherwise hide ::sin inside toa::)
using ::sqrt; using ::sin; using ::cos; using ::tan; using ::atan; using ::atan2; using ::tanh;
using ::exp; using ::log; using ::pow; using ::fabs; using std::abs;
using ::isnan; using ::isinf; using ::isfinite; using ::signbit;   // (the Jet overloads below must not hide the scalar ones)
using ::acos; using ::asin; using ::sinh; using ::cosh; using ::cbrt; using ::exp2; using ::log2; using ::log10;
using ::log1p; using ::expm1; using ::hypot; using ::erf; using ::erfc; using ::floor; using ::ceil; using ::fmax;
using ::fmin; using ::fdim; using ::fma; using ::copysign;

template <typename T, int N>
struct Jet {
  T a;     // scalar part          (jet.h:293)
  T v[N];  // infinitesimal part   (jet.h:296)

  __host__ __device__ Jet() : a(T(0)) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = T(0);
  }
  __host__ __device__ Jet(T value) : a(value) {  // jet.h:232 "constant"
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = T(0);
  }
  __host__ __device__ Jet(T value, int k) : a(value) {  // jet.h:238 "k-th variable"
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = (i == k) ? T(1) : T(0);
  }
  __host__ __device__ Jet& operator+=(const Jet& y) { *this = *this + y; return *this; }
  __host__ __device__ Jet& operator-=(const Jet& y) { *this = *this - y; return *this; }
  __host__ __device__ Jet& operator*=(const Jet& y) { *this = *this * y; return *this; }
  __host__ __device__ Jet& operator/=(const Jet& y) { *this = *this / y; return *this; }
  __host__ __device__ Jet& operator+=(T s) { a += s; return *this; }
  __host__ __device__ Jet& operator-=(T s) { a -= s; return *this; }
  __host__ __device__ Jet& operator*=(T s) { *this = *this * s; return *this; }
  __host__ __device__ Jet& operator/=(T s) { *this = *this / s; return *this; }
};

#define TOA_JET_FN template <typename T, int N> __host__ __device__ inline
#define TOA_JET_LOOP _Pragma("unroll") for (int i = 0; i < N; ++i)

TOA_JET_FN Jet<T, N> operator+(const Jet<T, N>& f) { return f; }                                   // jet.h:305
TOA_JET_FN Jet<T, N> operator-(const Jet<T, N>& f) { Jet<T, N> r; r.a = -f.a; TOA_JET_LOOP r.v[i] = -f.v[i]; return r; }  // :314
TOA_JET_FN Jet<T, N> operator+(const Jet<T, N>& f, const Jet<T, N>& g) {                            // :320
  Jet<T, N> r; r.a = f.a + g.a; TOA_JET_LOOP r.v[i] = f.v[i] + g.v[i]; return r;
}
TOA_JET_FN Jet<T, N> operator+(const Jet<T, N>& f, T s) { Jet<T, N> r = f; r.a += s; return r; }    // :332
TOA_JET_FN Jet<T, N> operator+(T s, const Jet<T, N>& f) { Jet<T, N> r = f; r.a += s; return r; }    // :338
TOA_JET_FN Jet<T, N> operator-(const Jet<T, N>& f, const Jet<T, N>& g) {                            // :344
  Jet<T, N> r; r.a = f.a - g.a; TOA_JET_LOOP r.v[i] = f.v[i] - g.v[i]; return r;
}
TOA_JET_FN Jet<T, N> operator-(const Jet<T, N>& f, T s) { Jet<T, N> r = f; r.a -= s; return r; }    // :356
TOA_JET_FN Jet<T, N> operator-(T s, const Jet<T, N>& f) { Jet<T, N> r = -f; r.a += s; return r; }   // :362
TOA_JET_FN Jet<T, N> operator*(const Jet<T, N>& f, const Jet<T, N>& g) {                            // :368
  Jet<T, N> r; r.a = f.a * g.a; TOA_JET_LOOP r.v[i] = f.a * g.v[i] + f.v[i] * g.a; return r;
}
TOA_JET_FN Jet<T, N> operator*(const Jet<T, N>& f, T s) { Jet<T, N> r; r.a = f.a * s; TOA_JET_LOOP r.v[i] = f.v[i] * s; return r; }  // :380
TOA_JET_FN Jet<T, N> operator*(T s, const Jet<T, N>& f) { return f * s; }                           // :386
TOA_JET_FN Jet<T, N> operator/(const Jet<T, N>& f, const Jet<T, N>& g) {                            // :392-410
  const T g_a_inverse = T(1.0) / g.a;
  const T f_a_by_g_a = f.a * g_a_inverse;
  Jet<T, N> r; r.a = f_a_by_g_a;
  TOA_JET_LOOP r.v[i] = (f.v[i] - f_a_by_g_a * g.v[i]) * g_a_inverse;
  return r;
}
TOA_JET_FN Jet<T, N> operator/(T s, const Jet<T, N>& g) {                                           // :413
  const T k = -s / (g.a * g.a);
  Jet<T, N> r; r.a = s / g.a; TOA_JET_LOOP r.v[i] = g.v[i] * k; return r;
}
TOA_JET_FN Jet<T, N> operator/(const Jet<T, N>& f, T s) {                                           // :420
  const T si = T(1.0) / s;
  Jet<T, N> r; r.a = f.a * si; TOA_JET_LOOP r.v[i] = f.v[i] * si; return r;
}
// comparisons act on the scalar part (jet.h:426-460)
TOA_JET_FN bool operator<(const Jet<T, N>& f, const Jet<T, N>& g) { return f.a < g.a; }
TOA_JET_FN bool operator>(const Jet<T, N>& f, const Jet<T, N>& g) { return f.a > g.a; }
TOA_JET_FN bool operator<(const Jet<T, N>& f, T s) { return f.a < s; }
TOA_JET_FN bool operator>(const Jet<T, N>& f, T s) { return f.a > s; }

// chain rule helper: value fa, derivative dfa
TOA_JET_FN Jet<T, N> jet_chain(const Jet<T, N>& f, T fa, T dfa) {
  Jet<T, N> r; r.a = fa; TOA_JET_LOOP r.v[i] = dfa * f.v[i]; return r;
}
// scalar overloads so that functors templated on the scalar type compile for S = T as well
__host__ __device__ inline float jsqrt(float x) { return sqrtf(x); }
__host__ __device__ inline double jsqrt(double x) { return sqrt(x); }

TOA_JET_FN Jet<T, N> abs(const Jet<T, N>& f) { return f.a < T(0) ? -f : f; }                         // :558 (copysign form)
TOA_JET_FN Jet<T, N> log(const Jet<T, N>& f) { return jet_chain(f, T(::log(f.a)), T(1) / f.a); }     // :606
TOA_JET_FN Jet<T, N> exp(const Jet<T, N>& f) { const T e = T(::exp(f.a)); return jet_chain(f, e, e); }  // :628
TOA_JET_FN Jet<T, N> sqrt(const Jet<T, N>& f) {                                                      // :643
  const T s = jsqrt(f.a);
  return jet_chain(f, s, T(1.0) / (T(2.0) * s));
}
TOA_JET_FN Jet<T, N> cos(const Jet<T, N>& f) { return jet_chain(f, T(::cos(f.a)), -T(::sin(f.a))); }  // :651
TOA_JET_FN Jet<T, N> sin(const Jet<T, N>& f) { return jet_chain(f, T(::sin(f.a)), T(::cos(f.a))); }   // :664
TOA_JET_FN Jet<T, N> tan(const Jet<T, N>& f) {                                                       // :677
  const T t = T(::tan(f.a));
  return jet_chain(f, t, T(1.0) + t * t);
}
TOA_JET_FN Jet<T, N> atan(const Jet<T, N>& f) { return jet_chain(f, T(::atan(f.a)), T(1.0) / (T(1.0) + f.a * f.a)); }  // :685
TOA_JET_FN Jet<T, N> tanh(const Jet<T, N>& f) {                                                      // :704
  const T t = T(::tanh(f.a));
  return jet_chain(f, t, T(1.0) - t * t);
}
TOA_JET_FN Jet<T, N> atan2(const Jet<T, N>& g, const Jet<T, N>& f) {                                 // :1223
  const T tmp = T(1.0) / (f.a * f.a + g.a * g.a);
  Jet<T, N> r; r.a = T(::atan2(g.a, f.a));
  TOA_JET_LOOP r.v[i] = tmp * (-g.a * f.v[i] + f.a * g.v[i]);
  return r;
}
TOA_JET_FN Jet<T, N> pow(const Jet<T, N>& f, double g) {                                             // :1258
  const T tmp = T(g) * T(::pow(f.a, T(g) - T(1.0)));
  return jet_chain(f, T(::pow(f.a, T(g))), tmp);
}

// ---- the rest of jet.h:557-1340 (same formulas; the line cited is the reference's) -------------------------------------
TOA_JET_FN bool operator<=(const Jet<T, N>& f, const Jet<T, N>& g) { return f.a <= g.a; }           // :426-460 scalar part
TOA_JET_FN bool operator>=(const Jet<T, N>& f, const Jet<T, N>& g) { return f.a >= g.a; }
TOA_JET_FN bool operator==(const Jet<T, N>& f, const Jet<T, N>& g) { return f.a == g.a; }
TOA_JET_FN bool operator!=(const Jet<T, N>& f, const Jet<T, N>& g) { return f.a != g.a; }
TOA_JET_FN bool operator<=(const Jet<T, N>& f, T s) { return f.a <= s; }
TOA_JET_FN bool operator>=(const Jet<T, N>& f, T s) { return f.a >= s; }
TOA_JET_FN bool operator==(const Jet<T, N>& f, T s) { return f.a == s; }
TOA_JET_FN Jet<T, N> copysign(const Jet<T, N>& f, const Jet<T, N>& g) {                             // :584-604
  const T d = g.a == T(0) ? T(INFINITY) : T(0);   // Dirac delta of the sign flip
  const T sa = T(::copysign(T(1), f.a)), sb = T(::copysign(T(1), g.a));
  Jet<T, N> r; r.a = T(::copysign(f.a, g.a));
  TOA_JET_LOOP r.v[i] = sa * sb * f.v[i] + T(::fabs(f.a)) * d * g.v[i];
  return r;
}
TOA_JET_FN Jet<T, N> log10(const Jet<T, N>& f) { return jet_chain(f, T(::log10(f.a)), T(1.0) / (f.a * T(::log(T(10.0))))); }   // :613
TOA_JET_FN Jet<T, N> log1p(const Jet<T, N>& f) { return jet_chain(f, T(::log1p(f.a)), T(1.0) / (T(1.0) + f.a)); }             // :621
TOA_JET_FN Jet<T, N> expm1(const Jet<T, N>& f) { const T e = T(::expm1(f.a)); return jet_chain(f, e, e + T(1.0)); }           // :635
TOA_JET_FN Jet<T, N> acos(const Jet<T, N>& f) { return jet_chain(f, T(::acos(f.a)), -T(1.0) / jsqrt(T(1.0) - f.a * f.a)); }   // :657
TOA_JET_FN Jet<T, N> asin(const Jet<T, N>& f) { return jet_chain(f, T(::asin(f.a)), T(1.0) / jsqrt(T(1.0) - f.a * f.a)); }    // :670
TOA_JET_FN Jet<T, N> sinh(const Jet<T, N>& f) { return jet_chain(f, T(::sinh(f.a)), T(::cosh(f.a))); }                        // :692
TOA_JET_FN Jet<T, N> cosh(const Jet<T, N>& f) { return jet_chain(f, T(::cosh(f.a)), T(::sinh(f.a))); }                        // :698
TOA_JET_FN Jet<T, N> floor(const Jet<T, N>& f) { return Jet<T, N>(T(::floor(f.a))); }                                         // :715 zero derivative
TOA_JET_FN Jet<T, N> ceil(const Jet<T, N>& f) { return Jet<T, N>(T(::ceil(f.a))); }                                           // :724
TOA_JET_FN Jet<T, N> cbrt(const Jet<T, N>& f) { return jet_chain(f, T(::cbrt(f.a)), T(1.0) / (T(3.0) * T(::cbrt(f.a * f.a)))); }   // :732
TOA_JET_FN Jet<T, N> exp2(const Jet<T, N>& f) { const T e = T(::exp2(f.a)); return jet_chain(f, e, e * T(::log(T(2)))); }     // :739
TOA_JET_FN Jet<T, N> log2(const Jet<T, N>& f) { return jet_chain(f, T(::log2(f.a)), T(1.0) / (f.a * T(::log(T(2))))); }       // :747
TOA_JET_FN Jet<T, N> hypot(const Jet<T, N>& x, const Jet<T, N>& y) {                                                          // :757
  const T h = T(::hypot(x.a, y.a));
  Jet<T, N> r; r.a = h; TOA_JET_LOOP r.v[i] = x.a / h * x.v[i] + y.a / h * y.v[i]; return r;
}
TOA_JET_FN Jet<T, N> hypot(const Jet<T, N>& x, const Jet<T, N>& y, const Jet<T, N>& z) {                                      // :772
  const T h = jsqrt(x.a * x.a + y.a * y.a + z.a * z.a);
  Jet<T, N> r; r.a = h; TOA_JET_LOOP r.v[i] = x.a / h * x.v[i] + y.a / h * y.v[i] + z.a / h * z.v[i]; return r;
}
TOA_JET_FN Jet<T, N> fma(const Jet<T, N>& x, const Jet<T, N>& y, const Jet<T, N>& z) {                                        // :789
  Jet<T, N> r; r.a = T(::fma(x.a, y.a, z.a)); TOA_JET_LOOP r.v[i] = y.a * x.v[i] + x.a * y.v[i] + z.v[i]; return r;
}
// fmax / fmin: the larger / smaller scalar part wins; on equality the two Jets are averaged; NaN = missing data (:800-880)
TOA_JET_FN Jet<T, N> fmax(const Jet<T, N>& x, const Jet<T, N>& y) {
  if (x.a != x.a || y.a != y.a || x.a < y.a || x.a > y.a) return (x.a != x.a || x.a < y.a) ? y : x;
  return (x + y) * T(0.5);
}
TOA_JET_FN Jet<T, N> fmin(const Jet<T, N>& x, const Jet<T, N>& y) {
  if (x.a != x.a || y.a != y.a || x.a < y.a || x.a > y.a) return (x.a != x.a || x.a > y.a) ? y : x;
  return (x + y) * T(0.5);
}
TOA_JET_FN Jet<T, N> fmax(const Jet<T, N>& x, T s) { return fmax(x, Jet<T, N>(s)); }
TOA_JET_FN Jet<T, N> fmax(T s, const Jet<T, N>& x) { return fmax(Jet<T, N>(s), x); }
TOA_JET_FN Jet<T, N> fmin(const Jet<T, N>& x, T s) { return fmin(x, Jet<T, N>(s)); }
TOA_JET_FN Jet<T, N> fmin(T s, const Jet<T, N>& x) { return fmin(Jet<T, N>(s), x); }
TOA_JET_FN Jet<T, N> fdim(const Jet<T, N>& f, const Jet<T, N>& g) {                                                           // :878-888
  if (f.a != f.a || g.a != g.a) return Jet<T, N>(T(NAN));
  return f.a > g.a ? f - g : Jet<T, N>();
}
TOA_JET_FN Jet<T, N> erf(const Jet<T, N>& x) {                                                                                // :890  2/sqrt(pi) exp(-x^2)
  return jet_chain(x, T(::erf(x.a)), T(1.1283791670955125739) * T(::exp(-x.a * x.a)));
}
TOA_JET_FN Jet<T, N> erfc(const Jet<T, N>& x) {                                                                               // :904
  return jet_chain(x, T(::erfc(x.a)), -T(1.1283791670955125739) * T(::exp(-x.a * x.a)));
}
TOA_JET_FN Jet<T, N> norm(const Jet<T, N>& f) { return jet_chain(f, f.a * f.a, T(2) * f.a); }                                 // :1251
TOA_JET_FN Jet<T, N> pow(T f, const Jet<T, N>& g) {                                                                           // :1275-1300
  if (f == T(0) && g.a > T(0)) return Jet<T, N>(T(0));
  if (f < T(0) && g.a == T(::floor(g.a))) {
    Jet<T, N> r(T(::pow(f, g.a)));
    TOA_JET_LOOP if (g.v[i] != T(0)) r.v[i] = T(NAN);
    return r;
  }
  const T p = T(::pow(f, g.a));
  return jet_chain(g, p, T(::log(f)) * p);
}
TOA_JET_FN Jet<T, N> pow(const Jet<T, N>& f, const Jet<T, N>& g) {                                                            // :1338-1400
  if (f.a == T(0) && g.a >= T(1)) return g.a > T(1) ? Jet<T, N>(T(0)) : f;
  if (f.a < T(0) && g.a == T(::floor(g.a))) {
    const T t = g.a * T(::pow(f.a, g.a - T(1.0)));
    Jet<T, N> r = jet_chain(f, T(::pow(f.a, g.a)), t);
    TOA_JET_LOOP if (g.v[i] != T(0)) r.v[i] = T(NAN);
    return r;
  }
  const T t1 = T(::pow(f.a, g.a)), t2 = g.a * T(::pow(f.a, g.a - T(1.0))), t3 = t1 * T(::log(f.a));
  Jet<T, N> r; r.a = t1; TOA_JET_LOOP r.v[i] = t2 * f.v[i] + t3 * g.v[i]; return r;
}

// ---- Bessel functions of the first kind (jet.h:919-1009): J0' = -J1, Jn' = (J(n-1) - J(n+1)) / 2  (dlmf.nist.gov/10.6) ----
__device__ __forceinline__ double BesselJ0(double x) { return ::j0(x); }
__device__ __forceinline__ double BesselJ1(double x) { return ::j1(x); }
__device__ __forceinline__ double BesselJn(int n, double x) { return ::jn(n, x); }
// (fp32 goes through the fp64 routines: the device library's jnf runs its recurrence forward and loses 3 digits near 0)
__device__ __forceinline__ float BesselJ0(float x) { return float(::j0(double(x))); }
__device__ __forceinline__ float BesselJ1(float x) { return float(::j1(double(x))); }
__device__ __forceinline__ float BesselJn(int n, float x) { return float(::jn(n, double(x))); }
TOA_JET_FN Jet<T, N> BesselJ0(const Jet<T, N>& f) { return jet_chain(f, BesselJ0(f.a), -BesselJ1(f.a)); }                    // :958
TOA_JET_FN Jet<T, N> BesselJ1(const Jet<T, N>& f) { return jet_chain(f, BesselJ1(f.a), T(0.5) * (BesselJ0(f.a) - BesselJn(2, f.a))); }   // :969
TOA_JET_FN Jet<T, N> BesselJn(int n, const Jet<T, N>& f) {                                                                    // :981
  return jet_chain(f, BesselJn(n, f.a), T(0.5) * (BesselJn(n - 1, f.a) - BesselJn(n + 1, f.a)));
}
// std::cyl_bessel_j(v, x) for the INTEGER orders the device math library has (jet.h:999-1009; order 0: -J1 f')
TOA_JET_FN Jet<T, N> cyl_bessel_j(int v, const Jet<T, N>& f) { return v == 0 ? BesselJ0(f) : BesselJn(v, f); }
// ---- lerp / midpoint (jet.h:1171-1218, C++20): d lerp = (1 - t) da + t db + (b - a) dt; d midpoint = (da + db) / 2 ----
TOA_JET_FN Jet<T, N> lerp(const Jet<T, N>& a, const Jet<T, N>& b, const Jet<T, N>& t) {
  Jet<T, N> r;
  r.a = a.a + t.a * (b.a - a.a);
  TOA_JET_LOOP r.v[i] = (T(1) - t.a) * a.v[i] + t.a * b.v[i] + (b.a - a.a) * t.v[i];
  return r;
}
TOA_JET_FN Jet<T, N> midpoint(const Jet<T, N>& a, const Jet<T, N>& b) {
  Jet<T, N> r;
  r.a = a.a * T(0.5) + b.a * T(0.5);                       // (overflow-safe, like std::midpoint)
  TOA_JET_LOOP r.v[i] = a.v[i] * T(0.5) + b.v[i] * T(0.5);
  return r;
}
// ---- classification and comparison on the SCALAR part only (jet.h:1011-1168) ----
TOA_JET_FN bool isfinite(const Jet<T, N>& f) { return ::isfinite(f.a); }
TOA_JET_FN bool isinf(const Jet<T, N>& f) { return ::isinf(f.a); }
TOA_JET_FN bool isnan(const Jet<T, N>& f) { return f.a != f.a; }
TOA_JET_FN bool isnormal(const Jet<T, N>& f) {
  const T m = f.a < T(0) ? -f.a : f.a;
  return ::isfinite(f.a) && m >= (sizeof(T) == 4 ? T(1.17549435e-38f) : T(2.2250738585072014e-308));
}
TOA_JET_FN bool signbit(const Jet<T, N>& f) { return ::signbit(f.a); }
TOA_JET_FN int fpclassify(const Jet<T, N>& f) {                 // FP_NAN 0, FP_INFINITE 1, FP_ZERO 2, FP_SUBNORMAL 3, FP_NORMAL 4
  if (f.a != f.a) return 0;
  if (::isinf(f.a)) return 1;
  if (f.a == T(0)) return 2;
  return isnormal(f) ? 4 : 3;
}
TOA_JET_FN bool isless(const Jet<T, N>& f, const Jet<T, N>& g) { return f.a < g.a; }
TOA_JET_FN bool isgreater(const Jet<T, N>& f, const Jet<T, N>& g) { return f.a > g.a; }
TOA_JET_FN bool islessequal(const Jet<T, N>& f, const Jet<T, N>& g) { return f.a <= g.a; }
TOA_JET_FN bool isgreaterequal(const Jet<T, N>& f, const Jet<T, N>& g) { return f.a >= g.a; }
TOA_JET_FN bool islessgreater(const Jet<T, N>& f, const Jet<T, N>& g) { return f.a < g.a || f.a > g.a; }
TOA_JET_FN bool isunordered(const Jet<T, N>& f, const Jet<T, N>& g) { return f.a != f.a || g.a != g.a; }
// the deprecated spellings (jet.h:1130-1168)
TOA_JET_FN bool IsFinite(const Jet<T, N>& f) { return isfinite(f); }
TOA_JET_FN bool IsNaN(const Jet<T, N>& f) { return isnan(f); }
TOA_JET_FN bool IsNormal(const Jet<T, N>& f) { return isnormal(f); }
TOA_JET_FN bool IsInfinite(const Jet<T, N>& f) { return isinf(f); }

#undef TOA_JET_FN
#undef TOA_JET_LOOP

}  // namespace toa
