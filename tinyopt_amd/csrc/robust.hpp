// Device M-estimators — the reference's robust norms (include/tinyopt/losses/robust_norms.h:32-316) in the
// "squared norm in, (loss, scale) out" form a residual functor uses them (docs/API.md:402-406):
//   l = rho(n2, th2),  s = d l / d n2  = the factor the reference applies to the Jacobian / gradient
//   ("JtJ * dx = Jt*res*s", robust_norms.h:20-26).
// Used inside K1 by the models that carry a loss tag (Se3ReprojModel) and exposed alone as toa_robust_norm.
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>

#ifdef __HIPCC_RTC__   // run-time compilation (jit.hip): the headers are handed to hiprtc by NAME, embedded in the library
#include "tinyopt_amd.h"
#else
#include "../../include/tinyopt_amd.h"
#endif

namespace toa {

template <typename T> struct RobustLim;
template <> struct RobustLim<float> { static __host__ __device__ constexpr float tiny() { return FLT_MIN; } };    // numeric_limits<T>::min()
template <> struct RobustLim<double> { static __host__ __device__ constexpr double tiny() { return DBL_MIN; } };

__device__ __forceinline__ float r_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double r_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float r_log(float x) { return logf(x); }
__device__ __forceinline__ double r_log(double x) { return log(x); }
__device__ __forceinline__ float r_exp(float x) { return expf(x); }
__device__ __forceinline__ double r_exp(double x) { return exp(x); }
__device__ __forceinline__ float r_atan2(float y, float x) { return atan2f(y, x); }
__device__ __forceinline__ double r_atan2(double y, double x) { return atan2(y, x); }

// kind: TOA_LOSS_* (include/tinyopt_amd.h).  Uniform across the wave in every caller, so the switch does not diverge.
template <typename T>
__device__ __forceinline__ void robust_norm(int kind, T n2, T th2, T& l, T& s) {
  switch (kind) {
    case TOA_LOSS_TRUNCATED:  // robust_norms.h:36-57: clip the loss, scale in {0, 1}
      l = n2 <= th2 ? n2 : th2;
      s = n2 <= th2 ? T(1) : T(0);
      break;
    case TOA_LOSS_HUBER: {  // :73-105: l = 2 th n - th^2 beyond the threshold, scale th / n
      if (n2 <= th2) { l = n2; s = T(1); }
      else {
        const T th = r_sqrt(th2), n = r_sqrt(n2);
        l = T(2.0) * th * n - th2;
        s = fmax(RobustLim<T>::tiny(), th / n);
      }
      break;
    }
    case TOA_LOSS_TUKEY: {  // :122-152
      if (n2 <= th2) {
        const T q = T(1.0) - n2 / th2, q2 = q * q;
        l = th2 * (T(1.0) - q2 * q);
        s = T(3.0) * (th2 - n2) * (th2 - n2) / (th2 * th2);
      } else { l = th2; s = T(0); }
      break;
    }
    case TOA_LOSS_ARCTAN: {  // :168-190
      const T th = r_sqrt(th2);
      l = th * r_atan2(n2, th);
      s = fmax(RobustLim<T>::tiny(), T(1.0) / (n2 * n2 / th2 + T(1.0)));
      break;
    }
    case TOA_LOSS_CAUCHY: {  // :207-228
      const T q = T(1.0) + n2 / th2;
      l = th2 * r_log(q);
      s = fmax(RobustLim<T>::tiny(), T(1.0) / q);
      break;
    }
    case TOA_LOSS_GEMAN_MCCLURE: {  // :245-265
      const T e = n2 + th2;
      l = n2 / e;
      s = th2 / (e * e);
      break;
    }
    case TOA_LOSS_BLAKE_ZISSERMAN: {  // :282-303
      const T eps = r_exp(-th2);
      l = -r_log(r_exp(-n2) + eps);
      s = T(1.0) / (eps * r_exp(n2) + T(1.0));
      break;
    }
    default: l = n2; s = T(1); break;  // plain squared L2
  }
}

}  // namespace toa
