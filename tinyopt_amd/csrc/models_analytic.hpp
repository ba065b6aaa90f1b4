// Device residual models, part 2: the families with closed-form Accumulate callbacks and m = n or tiny systems — GaussianPrior
// (the reference's published dense benchmark), MahaPrior, TestFn (its analytic optimizer tests), Sqrt2 (README example).
#pragma once
#include "models_dense.hpp"

namespace toa {

// Gaussian prior  r = (x - y) / sigma,  m = n — the residual of the reference's published dense
// benchmark, with the semantics of its manual Accumulate callback (benchmarks/dense.cpp:57-66, 90-99;
// losses/mahalanobis.h:124-136): grad = J * res with J = diag(1/sigma), H.diagonal() = sigma^-2 (H was
// cleared: off-diagonals are 0), returns res.squaredNorm() as a SCALAR => Cost(v, 1) (cost.h:22).
// Data per problem: [y (n) | sigma (n)].  Same operation order as the oracle => g and H bit-identical.
template <typename T, int NPAD>
struct GaussianPriorModel {
  using Scalar = T;
  __device__ __forceinline__ void set_loss(int, double) {}  // no M-estimator on this family
  static constexpr int kXdim = 0;
  __device__ __forceinline__ void plus_eq(WaveLds<T>& L, const T* d, T sign, int, int lane) const { euclid_plus_eq(L, d, sign, lane); }
  static constexpr int kNpad = NPAD;
  const T* data;
  const T* y;
  const T* sigma;
  int n_;
  __device__ __forceinline__ void init(int n, int, const void* d) { n_ = n; data = static_cast<const T*>(d); }
  __device__ __forceinline__ void bind(long long p) { y = data + size_t(p) * 2 * n_; sigma = y + n_; }
  __device__ __forceinline__ void bind_chunk(long long p, int, int, int) { bind(p); }  // not row-splittable: one chunk
  __device__ __forceinline__ T residual(const WaveLds<T>& L, int n, int lane, T& inv_sigma) const {
    if (lane >= n) { inv_sigma = T(0); return T(0); }
    const T s = sigma[lane];
    inv_sigma = T(1) / s;
    return (L.xs[lane] - y[lane]) / s;
  }
  __device__ __forceinline__ void accumulate(WaveLds<T>& L, int n, int lane, T& cost, int& nres) {
    T is;
    const T r = residual(L, n, lane, is);
    if (lane < n) { L.g[lane] = is * r; L.hd[lane] = is * is; }
    cost = wave_allreduce_sum(r * r);
    nres = 1;
    wave_sync();
  }
  __device__ __forceinline__ void evaluate(WaveLds<T>& L, int n, int lane, T& cost, int& nres) {
    T is;
    const T r = residual(L, n, lane, is);  // MahaSquaredNorm(x - y, stdevs), dense.cpp:63-65
    cost = wave_allreduce_sum(r * r);
    nres = 1;
  }
  template <typename O>
  __device__ __forceinline__ void write_sym(O* M, int LD, int n, int lane) const {
    if (lane < n)
      for (int j = 0; j < n; ++j) M[lane * LD + j] = O(0);
  }
};

// Gaussian prior with a GENERAL covariance, whitened by the upper Cholesky factor U of the information matrix:
// res = U (x - y), J = U  (losses/mahalanobis.h:160-171 MahaWhitenedInfoU; tests/cov.cpp:91-146), folded as the AD
// bridge folds a residual vector: grad = J^T res, H = J^T J (= cov^-1), cost = ||res||^2 over n residuals.
// data: [P][n + n*n] = y, then U row-major (upper triangular).  Lane a owns residual a / gradient entry a / row a of H.
// A parity model (tests/cov.cpp: the covariance of the solve must equal the prior's), not a throughput model: H is
// recomputed from U at every build.
template <typename T, int NPAD>
struct MahaPriorModel {
  using Scalar = T;
  __device__ __forceinline__ void set_loss(int, double) {}  // no M-estimator on this family
  static constexpr int kXdim = 0;
  static constexpr int kNpad = NPAD;
  __device__ __forceinline__ void plus_eq(WaveLds<T>& L, const T* d, T sign, int, int lane) const { euclid_plus_eq(L, d, sign, lane); }
  const T* data;
  const T* y;
  const T* U;
  int n_;
  __device__ __forceinline__ void init(int n, int, const void* d) { n_ = n; data = static_cast<const T*>(d); }
  __device__ __forceinline__ void bind(long long p) { y = data + size_t(p) * (n_ + size_t(n_) * n_); U = y + n_; }
  __device__ __forceinline__ void bind_chunk(long long p, int, int, int) { bind(p); }
  __device__ __forceinline__ T residual(WaveLds<T>& L, int n, int lane) const {
    L.tmp[lane] = lane < n ? L.xs[lane] - y[lane] : T(0);
    wave_sync();
    T r = 0;
    if (lane < n)
      for (int j = lane; j < n; ++j) r += U[size_t(lane) * n + j] * L.tmp[j];  // triangularView<Upper>
    return r;
  }
  __device__ __forceinline__ void accumulate(WaveLds<T>& L, int n, int lane, T& cost, int& nres) {
    const T r = residual(L, n, lane);
    L.vec[lane] = r;
    wave_sync();
    if (lane < n) {
      T g = 0, hd = 0;
      for (int i = 0; i <= lane; ++i) {  // column `lane` of U has its non-zeros in rows 0..lane
        const T u = U[size_t(i) * n + lane];
        g += u * L.vec[i];
        hd += u * u;
      }
      L.g[lane] = g;
      L.hd[lane] = hd;
    }
    cost = wave_allreduce_sum(r * r);
    nres = n;
    wave_sync();
  }
  __device__ __forceinline__ void evaluate(WaveLds<T>& L, int n, int lane, T& cost, int& nres) {
    const T r = residual(L, n, lane);
    cost = wave_allreduce_sum(r * r);
    nres = n;
    wave_sync();
  }
  template <typename O>
  __device__ __forceinline__ void write_sym(O* M, int LD, int n, int lane) const {
    if (lane < n)
      for (int b = 0; b < n; ++b) {  // H[a][b] = sum_i U[i][a] U[i][b], i <= min(a, b)
        const int top = lane < b ? lane : b;
        T h = 0;
        for (int i = 0; i <= top; ++i) h += U[size_t(i) * n + lane] * U[size_t(i) * n + b];
        M[lane * LD + b] = O(h);
      }
  }
};

// The analytic test functions of the reference's optimizer tests as MANUAL Accumulate callbacks
// (`auto loss = [&](const auto& v, auto& grad, auto& H)`), exact Hessians included — they drive the LM state
// machine through its bad-step, failed-solve (indefinite H) and rollback branches:
//   0 Rosenbrock  tests/optimize_easy.cpp:35-79     1 plateau (Easom-like)  :88-144     2 Powell singular  :153-221
//   3 Beale       tests/optimize_hard.cpp:34-63     4 Himmelblau            :72-102   (residual vectors, J^T J / J^T r)
//   5 x - 2       tests/basic.cpp:41-54,72-87 (n = 1): grad = res, H = 1, cost = |res|
// data: [1] = function id (as T).  Every lane evaluates the same scalars (n <= 4): no divergence, no reductions.
template <typename T>
struct TestFnModel {
  using Scalar = T;
  __device__ __forceinline__ void set_loss(int, double) {}  // no M-estimator on this family
  static constexpr int kXdim = 0;
  static constexpr int kNpad = 16;
  __device__ __forceinline__ void plus_eq(WaveLds<T>& L, const T* d, T sign, int, int lane) const { euclid_plus_eq(L, d, sign, lane); }
  int fn;
  T H[16];
  __device__ __forceinline__ void init(int, int, const void* d) { fn = int(static_cast<const T*>(d)[0]); }
  __device__ __forceinline__ void bind(long long) {}
  __device__ __forceinline__ void bind_chunk(long long, int, int, int) {}
  static __device__ __forceinline__ T pw(T t, int e) { return T(::pow(double(t), double(e))); }  // std::pow(t, 3): double
  template <bool WANT>
  __device__ __forceinline__ T eval(const WaveLds<T>& L, T* g, int& nres) {
    const T v0 = L.xs[0], v1 = L.xs[1], v2 = L.xs[2], v3 = L.xs[3];
    nres = 1;
    if (fn == 5) {
      const T res = v0 - T(2);
      if (WANT) { g[0] = res; H[0] = T(1); }
      return res < T(0) ? -res : res;
    }
    if (fn == 0) {
      const T t1 = T(1.0) - v0, t2 = v1 - v0 * v0;
      if (WANT) {
        g[0] = T(-2.0) * t1 - T(400.0) * v0 * t2;
        g[1] = T(200.0) * t2;
        H[0] = T(2.0) - T(400.0) * v1 + T(1200.0) * v0 * v0;
        H[1] = H[4] = T(-400.0) * v0;
        H[5] = T(200.0);
      }
      return t1 * t1 + T(100.0) * t2 * t2;
    }
    if (fn == 1) {
      const T PI = T(3.14159265358979323846);
      const T dx = v0 - PI, dy = v1 - PI;
      const T ex = T(::exp(-(dx * dx + dy * dy)));
      const T cx = T(::cos(v0)), cy = T(::cos(v1)), sx = T(::sin(v0)), sy = T(::sin(v1));
      if (WANT) {
        g[0] = cy * ex * (sx + T(2.0) * dx * cx);
        g[1] = cx * ex * (sy + T(2.0) * dy * cy);
        H[0] = cy * ex * (cx - T(4.0) * dx * sx + (T(2.0) - T(4.0) * dx * dx) * cx);
        H[5] = cx * ex * (cy - T(4.0) * dy * sy + (T(2.0) - T(4.0) * dy * dy) * cy);
        H[1] = H[4] = ex * (sx + T(2.0) * dx * cx) * (sy + T(2.0) * dy * cy);
      }
      return T(1.0) - (cx * cy * ex);
    }
    if (fn == 2) {
      const T t1 = v0 + T(10.0) * v1, t2 = v2 - v3, t3 = v1 - T(2.0) * v2, t4 = v0 - v3;
      if (WANT) {
        g[0] = T(2.0) * t1 + T(40.0) * pw(t4, 3);
        g[1] = T(20.0) * t1 + T(4.0) * pw(t3, 3);
        g[2] = T(10.0) * t2 - T(8.0) * pw(t3, 3);
        g[3] = T(-10.0) * t2 - T(40.0) * pw(t4, 3);
#pragma unroll
        for (int i = 0; i < 16; ++i) H[i] = T(0);
        const T d3 = T(12.0) * t3 * t3, d4 = T(120.0) * t4 * t4;
        H[0 * 4 + 0] = T(2.0) + d4;  H[0 * 4 + 1] = T(20.0);           H[0 * 4 + 3] = -d4;
        H[1 * 4 + 0] = T(20.0);      H[1 * 4 + 1] = T(200.0) + d3;     H[1 * 4 + 2] = T(-2.0) * d3;
        H[2 * 4 + 1] = T(-2.0) * d3; H[2 * 4 + 2] = T(10.0) + T(4.0) * d3; H[2 * 4 + 3] = T(-10.0);
        H[3 * 4 + 0] = -d4;          H[3 * 4 + 2] = T(-10.0);          H[3 * 4 + 3] = T(10.0) + d4;
      }
      return t1 * t1 + T(5.0) * t2 * t2 + pw(t3, 4) + pw(t4, 4) * T(10.0);
    }
    // residual-vector functions: grad = J^T r, H = J^T J, cost = ||r||^2 (optimize_autodiff.h:151-164)
    T r[3], J[3][2];
    int mr;
    if (fn == 3) {
      mr = 3;
      r[0] = T(1.5) - v0 + v0 * v1; r[1] = T(2.25) - v0 + v0 * v1 * v1; r[2] = T(2.625) - v0 + v0 * v1 * v1 * v1;
      J[0][0] = T(-1) + v1;           J[0][1] = v0;
      J[1][0] = T(-1) + v1 * v1;      J[1][1] = T(2) * v0 * v1;
      J[2][0] = T(-1) + v1 * v1 * v1; J[2][1] = T(3) * v0 * v1 * v1;
    } else {
      mr = 2;
      r[0] = v0 * v0 + v1 - T(11.0); r[1] = v0 + v1 * v1 - T(7.0); r[2] = T(0);
      J[0][0] = T(2) * v0; J[0][1] = T(1);
      J[1][0] = T(1);      J[1][1] = T(2) * v1;
      J[2][0] = J[2][1] = T(0);
    }
    nres = mr;
    if (WANT) {
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        T s = 0;
        for (int i = 0; i < mr; ++i) s += J[i][a] * r[i];
        g[a] = s;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          T q = 0;
          for (int i = 0; i < mr; ++i) q += J[i][a] * J[i][b];
          H[a * 4 + b] = q;
        }
      }
    }
    T c = 0;
    for (int i = 0; i < mr; ++i) c += r[i] * r[i];
    return c;
  }
  __device__ __forceinline__ void accumulate(WaveLds<T>& L, int n, int lane, T& cost, int& nres) {
    T g[4] = {0, 0, 0, 0};
    cost = eval<true>(L, g, nres);
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
        if (a < n) { L.g[a] = g[a]; L.hd[a] = H[a * 4 + a]; }
    }
    wave_sync();
  }
  __device__ __forceinline__ void evaluate(WaveLds<T>& L, int, int, T& cost, int& nres) {
    T g[4];
    cost = eval<false>(L, g, nres);
  }
  template <typename O>
  __device__ __forceinline__ void write_sym(O* M, int LD, int n, int lane) const {
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if (a < n && b < n) M[a * LD + b] = O(H[a * 4 + b]);
    }
  }
};

// sqrt(2):  r = x*x - 2, n = m = 1 (tests/sqrt2.cpp:30-70): grad = J r, H = J^2, cost = r^2 (1 residual).
template <typename T>
struct Sqrt2Model {
  using Scalar = T;
  __device__ __forceinline__ void set_loss(int, double) {}  // no M-estimator on this family
  static constexpr int kXdim = 0;
  __device__ __forceinline__ void plus_eq(WaveLds<T>& L, const T* d, T sign, int, int lane) const { euclid_plus_eq(L, d, sign, lane); }
  static constexpr int kNpad = 16;
  __device__ __forceinline__ void init(int, int, const void*) {}
  __device__ __forceinline__ void bind(long long) {}
  __device__ __forceinline__ void bind_chunk(long long, int, int, int) {}
  __device__ __forceinline__ void accumulate(WaveLds<T>& L, int, int lane, T& cost, int& nres) {
    const T x = L.xs[0];
    const T r = x * x - T(2), J = T(2) * x;
    if (lane == 0) { L.g[0] = J * r; L.hd[0] = J * J; }
    cost = r * r;
    nres = 1;
    wave_sync();
  }
  __device__ __forceinline__ void evaluate(WaveLds<T>& L, int, int, T& cost, int& nres) {
    const T x = L.xs[0];
    const T r = x * x - T(2);
    cost = r * r;
    nres = 1;
  }
  template <typename O>
  __device__ __forceinline__ void write_sym(O* M, int, int, int lane) const {
    if (lane == 0) M[0] = O(0);
  }
};

}  // namespace toa
