// Device forward-mode dual numbers — the GPU counterpart of the reference's vendored ceres::Jet<T, N>
// (include/tinyopt/3rdparty/ceres/jet.h:216-301; arithmetic :304-430; functions :557-1340) that
// OptimizeWithAutoDiff (include/tinyopt/diff/optimize_autodiff.h:21-169) evaluates user residuals on.
//
// Same algebra, same formulas (each operator cites the reference line it restates); what differs is the
// storage: the infinitesimal part is a plain T[N] that lives in VGPRs (N is a compile-time constant <= 12
// here; the reference's N = Dynamic heap vectors have no place in a kernel).  With it a user writes only
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

#undef TOA_JET_FN
#undef TOA_JET_LOOP

}  // namespace toa
