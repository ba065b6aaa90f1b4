// Device residual models, part 3: manifold policies (how a step is applied to the stored parameters), SE3 / SO3 maps over the
// scalar type, the SE3 prior and reprojection models.
#pragma once
#include "models_analytic.hpp"

namespace toa {

// ---- manifold policies: how a step is applied to the stored parameters ---------------------------------
template <typename T>
struct EuclidManifold {
  static constexpr int kXdim = 0;
  static __device__ __forceinline__ void plus_eq(WaveLds<T>& L, const T* d, T sign, int, int lane) { euclid_plus_eq(L, d, sign, lane); }
};
template <typename T>
struct Se3Manifold {
  static constexpr int kXdim = 12;
  // pose <- pose * exp(sign * delta): SO3 Rodrigues with small-angle series, SE3 V matrix (Sophus' formulas)
  static __device__ __forceinline__ void plus_eq(WaveLds<T>& L, const T* dv, T sign, int, int lane) {
    T dl[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) dl[i] = sign * dv[i];
    const T wx = dl[3], wy = dl[4], wz = dl[5];
    const T t2 = wx * wx + wy * wy + wz * wz;
    const T th = sqrt(t2);
    T A, B, Cc;
    if (t2 < T(1e-10)) { A = T(1) - t2 / T(6); B = T(0.5) - t2 / T(24); Cc = T(1) / T(6) - t2 / T(120); }
    else { T sn, cs; sincos_t(th, &sn, &cs); A = sn / th; B = (T(1) - cs) / t2; Cc = (th - sn) / (t2 * th); }
    T Rd[9];
    Rd[0] = T(1) - B * (wy * wy + wz * wz); Rd[1] = -A * wz + B * wx * wy;          Rd[2] = A * wy + B * wx * wz;
    Rd[3] = A * wz + B * wx * wy;          Rd[4] = T(1) - B * (wx * wx + wz * wz); Rd[5] = -A * wx + B * wy * wz;
    Rd[6] = -A * wy + B * wx * wz;         Rd[7] = A * wx + B * wy * wz;          Rd[8] = T(1) - B * (wx * wx + wy * wy);
    const T c1[3] = {wy * dl[2] - wz * dl[1], wz * dl[0] - wx * dl[2], wx * dl[1] - wy * dl[0]};
    const T c2[3] = {wy * c1[2] - wz * c1[1], wz * c1[0] - wx * c1[2], wx * c1[1] - wy * c1[0]};
    T td[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) td[i] = dl[i] + B * c1[i] + Cc * c2[i];
    T x[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) x[i] = L.xs[i];
    wave_sync();
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) L.xs[3 * i + j] = x[3 * i] * Rd[j] + x[3 * i + 1] * Rd[3 + j] + x[3 * i + 2] * Rd[6 + j];
        L.xs[9 + i] = x[3 * i] * td[0] + x[3 * i + 1] * td[1] + x[3 * i + 2] * td[2] + x[9 + i];
      }
    }
    wave_sync();
  }
};

// ---- SE3 / SO3 log maps over the scalar type S (plain T for cost-only passes, Jet<T, 6> for differentiated ones), from a
//      rotation MATRIX (the published Sophus formulas; Sophus itself is an un-vendored dependency of the reference):
//        SO3: omega = (theta / sin theta) vee(R - R^T)/2, cos theta = (tr R - 1)/2
//        SE3: upsilon = V^-1 t, V^-1 = I - 1/2 [w]x + (1 - theta cos(theta/2) / (2 sin(theta/2))) / theta^2 [w]x^2
//      Near the identity (cos theta > 0.999) the coefficients come from their power series, smooth there: a square root
//      of a vanishing quantity would make every Jet derivative infinite exactly at the solution of a pose prior.
template <typename T> __device__ __forceinline__ T jet_scalar(const T& x) { return x; }
template <typename T, int N> __device__ __forceinline__ T jet_scalar(const Jet<T, N>& x) { return x.a; }

template <typename S, typename T>
__device__ __forceinline__ void se3_log(const S* R, const S* t, S* xi) {
  const S c = (R[0] + R[4] + R[8] - T(1.0)) * T(0.5);
  const S v[3] = {(R[7] - R[5]) * T(0.5), (R[2] - R[6]) * T(0.5), (R[3] - R[1]) * T(0.5)};  // sin(theta) * axis
  const S s2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  const bool small = jet_scalar(c) > T(0.999);
  S k, coef;
  if (small) {  // asin(x)/x, x = sin(theta)
    k = T(1.0) + s2 * (T(1.0 / 6.0) + s2 * (T(3.0 / 40.0) + s2 * (T(5.0 / 112.0) + s2 * T(35.0 / 1152.0))));
  } else {
    const S sn = sqrt(s2);
    k = atan2(sn, c) / sn;
  }
  const S th2 = s2 * k * k;
  if (small) {  // (1 - (theta/2) cot(theta/2)) / theta^2
    coef = T(1.0 / 12.0) + th2 * (T(1.0 / 720.0) + th2 * (T(1.0 / 30240.0) + th2 * T(1.0 / 1209600.0)));
  } else {
    const S th = sqrt(th2), h = th * T(0.5);
    coef = (T(1.0) - th * cos(h) / (T(2.0) * sin(h))) / th2;
  }
  const S w[3] = {v[0] * k, v[1] * k, v[2] * k};
  const S c1[3] = {w[1] * t[2] - w[2] * t[1], w[2] * t[0] - w[0] * t[2], w[0] * t[1] - w[1] * t[0]};
  const S c2[3] = {w[1] * c1[2] - w[2] * c1[1], w[2] * c1[0] - w[0] * c1[2], w[0] * c1[1] - w[1] * c1[0]};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    xi[i] = t[i] - c1[i] * T(0.5) + coef * c2[i];
    xi[3 + i] = w[i];
  }
}

// SE3 pose prior — the reference's own manifold test (tests/sophus.cpp:26-44): residual(x) = log(prior_inv * x) in R^6,
// differentiated on the device by Jet<T, 6> over the RIGHT perturbation x * exp(delta) at delta = 0, exactly what
// OptimizeWithAutoDiff does for a user type (optimize_autodiff.h:48-77 with sophus.h:24-26): exp(delta) enters the Jets
// through its first-order part I + [omega]x, upsilon (exact for first derivatives at 0).  data: [P][12] = prior_inv
// (R row-major, t); x: [P][12].  One wave per problem; the 6 x 6 system is evaluated redundantly by every lane.
template <typename T>
struct Se3PriorModel {
  using Scalar = T;
  __device__ __forceinline__ void set_loss(int, double) {}  // no M-estimator on this family
  static constexpr int kNpad = 16;
  static constexpr int kXdim = 12;
  const T* data;
  const T* P;
  T G[28];  // upper Gram of [J | r] (7 x 7)
  static __device__ __forceinline__ constexpr int tt(int a, int b) { return a * 7 - a * (a - 1) / 2 + (b - a); }
  __device__ __forceinline__ void init(int, int, const void* dp) { data = static_cast<const T*>(dp); }
  __device__ __forceinline__ void bind(long long p) { P = data + size_t(p) * 12; }
  __device__ __forceinline__ void bind_chunk(long long p, int, int, int) { bind(p); }
  template <typename S>
  __device__ __forceinline__ void residual(const S* Rx, const S* tx, S* xi) const {
    S RA[9], tA[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) RA[3 * i + j] = Rx[j] * P[3 * i] + Rx[3 + j] * P[3 * i + 1] + Rx[6 + j] * P[3 * i + 2];
      tA[i] = tx[0] * P[3 * i] + tx[1] * P[3 * i + 1] + tx[2] * P[3 * i + 2] + P[9 + i];
    }
    se3_log<S, T>(RA, tA, xi);
  }
  __device__ __forceinline__ void accumulate(WaveLds<T>& L, int, int lane, T& cost, int& nres) {
    using J6 = Jet<T, 6>;
    T R[9], t[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = L.xs[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = L.xs[9 + i];
    J6 d[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) d[k] = J6(T(0), k);  // delta = (upsilon, omega) seeded at 0
    J6 Rj[9], tj[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {  // R (I + [omega]x)
      Rj[3 * i + 0] = R[3 * i + 0] + (d[5] * R[3 * i + 1] - d[4] * R[3 * i + 2]);
      Rj[3 * i + 1] = R[3 * i + 1] + (d[3] * R[3 * i + 2] - d[5] * R[3 * i + 0]);
      Rj[3 * i + 2] = R[3 * i + 2] + (d[4] * R[3 * i + 0] - d[3] * R[3 * i + 1]);
      tj[i] = t[i] + (d[0] * R[3 * i] + d[1] * R[3 * i + 1] + d[2] * R[3 * i + 2]);
    }
    J6 xi[6];
    residual<J6>(Rj, tj, xi);
#pragma unroll
    for (int i = 0; i < 28; ++i) G[i] = T(0);
#pragma unroll
    for (int i = 0; i < 6; ++i) {  // fold residual i: w = [J_i | r_i]
      T w[7];
#pragma unroll
      for (int k = 0; k < 6; ++k) w[k] = xi[i].v[k];
      w[6] = xi[i].a;
#pragma unroll
      for (int a = 0; a < 7; ++a)
#pragma unroll
        for (int b = a; b < 7; ++b) G[tt(a, b)] += w[a] * w[b];
    }
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < 6; ++a) { L.g[a] = G[tt(a, 6)]; L.hd[a] = G[tt(a, a)]; }
    }
    cost = G[tt(6, 6)];
    nres = 6;
    wave_sync();
  }
  __device__ __forceinline__ void evaluate(WaveLds<T>& L, int, int, T& cost, int& nres) {
    T R[9], t[3], xi[6];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = L.xs[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = L.xs[9 + i];
    residual<T>(R, t, xi);
    T c = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) c += xi[i] * xi[i];
    cost = c;
    nres = 6;
  }
  template <typename O>
  __device__ __forceinline__ void write_sym(O* M, int LD, int, int lane) const {
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = a; b < 6; ++b) { M[a * LD + b] = O(G[tt(a, b)]); M[b * LD + a] = O(G[tt(a, b)]); }
    }
  }
  __device__ __forceinline__ void plus_eq(WaveLds<T>& L, const T* dv, T sign, int n, int lane) const {
    Se3Manifold<T>::plus_eq(L, dv, sign, n, lane);
  }
};

// SE3 pinhole reprojection: parameters = a pose stored as R (row-major 9) + t (3) = 12 scalars,
// tangent n = 6 in Sophus order (upsilon, omega); residual pair per point r = (f X/Z + cx - u, f Y/Z + cy - v),
// p_c = R p + t; Jacobian w.r.t. the RIGHT perturbation at delta = 0 (what OptimizeWithAutoDiff's user-type
// branch differentiates, optimize_autodiff.h:48-55,73-77): d p_c/d upsilon = R, d p_c/d omega = -R [p]x; update
// pose <- pose * exp(delta) (3rdparty/traits/sophus.h:24-26).  Thread-per-residual evaluation: lane l handles
// points l, l+64, ...; the 7x7 upper Gram of [J | r] (28 values) is accumulated in registers and folded across
// the wave once per pass.  Data per problem: [f cx cy 0 0 0 0 0 | x y z u v ...] (coalesced 5-scalar records).
// HEADER_LOSS = false: the variant for callers who guarantee that no problem's header names a loss (toa_tuning::se3_reproj_header_l2; both host
// mirrors set it for models constructed without one): without the estimators' exp / log / atan2 the fp64 fused kernel is 216 registers
// instead of 308 — two waves per SIMD instead of one (profiles/r06_ab_log.md section 12).
template <typename T, bool HEADER_LOSS = true>
struct Se3ReprojModel {
  using Scalar = T;
  __device__ __forceinline__ void set_loss(int, double) {}  // this family carries its loss in the data header
  static constexpr int kNpad = 16;
  static constexpr int kXdim = 12;
  // address_space(1): the data pointer reaches the kernels through a parameter block in memory, so hipcc cannot prove it
  // global and would emit flat loads, which count on the LDS counter too and serialise against the LDS-resident state machine
  using GP = const __attribute__((address_space(1))) T*;
  GP data;
  GP d;
  int npts, pt0, pt1;
  int ninl;  // inlier residuals of the last pass (cost.h:84 NumInliers)
  T G[28];
  static __device__ __forceinline__ constexpr int tt(int a, int b) { return a * 7 - a * (a - 1) / 2 + (b - a); }
  __device__ __forceinline__ void init(int, int m, const void* dp) { npts = m / 2; data = (GP)static_cast<const T*>(dp); }
  __device__ __forceinline__ void bind(long long p) { d = data + size_t(p) * (8 + 5 * size_t(npts)); pt0 = 0; pt1 = npts; }
  __device__ __forceinline__ void bind_chunk(long long p, int row0, int rows, int) {
    d = data + size_t(p) * (8 + 5 * size_t(npts));
    pt0 = row0 / 2;
    pt1 = min(npts, (row0 + rows) / 2);
  }

  template <bool WANT_H>
  __device__ __forceinline__ T pass(const WaveLds<T>& L, int lane) {
    T R[9], t[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = L.xs[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = L.xs[9 + i];
    const T f = d[0], cx = d[1], cy = d[2];
    const int loss = HEADER_LOSS ? int(d[3]) : 0;  // TOA_LOSS_*; 0 = plain squared L2 (wave-uniform)
    const T th2 = d[4];
    if (WANT_H) {
#pragma unroll
      for (int i = 0; i < 28; ++i) G[i] = T(0);
    }
    T csum = 0;
    T inl = 0;  // exact in T: <= 2 * points per lane
    GP pts = d + 8;
    // one point ahead: the next point's five scalars are in flight while this one is folded (a single resident wave
    // per chunk on the row-split path would otherwise pay one HBM round trip per point)
    T nq[5];
    int i = pt0 + lane;
    if (i < pt1) {
#pragma unroll
      for (int k = 0; k < 5; ++k) nq[k] = pts[size_t(i) * 5 + k];
    }
    for (; i < pt1; i += 64) {
      T q[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) q[k] = nq[k];
      if (i + 64 < pt1) {
#pragma unroll
        for (int k = 0; k < 5; ++k) nq[k] = pts[size_t(i + 64) * 5 + k];
      }
      const T px = q[0], py = q[1], pz = q[2];
      const T X = R[0] * px + R[1] * py + R[2] * pz + t[0];
      const T Y = R[3] * px + R[4] * py + R[5] * pz + t[1];
      const T Z = R[6] * px + R[7] * py + R[8] * pz + t[2];
      const T iz = T(1) / Z;
      T w[2][7];
      w[0][6] = f * X * iz + cx - q[3];
      w[1][6] = f * Y * iz + cy - q[4];
      if (WANT_H) {
        const T du0 = f * iz, du2 = -f * X * iz * iz;
        const T dv1 = f * iz, dv2 = -f * Y * iz * iz;
        T D[3][6];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          D[a][0] = R[3 * a]; D[a][1] = R[3 * a + 1]; D[a][2] = R[3 * a + 2];
          D[a][3] = -(R[3 * a + 1] * pz - R[3 * a + 2] * py);
          D[a][4] = -(-R[3 * a] * pz + R[3 * a + 2] * px);
          D[a][5] = -(R[3 * a] * py - R[3 * a + 1] * px);
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          w[0][c] = du0 * D[0][c] + du2 * D[2][c];
          w[1][c] = dv1 * D[1][c] + dv2 * D[2][c];
        }
        if (loss == TOA_LOSS_L2) {
#pragma unroll
          for (int row = 0; row < 2; ++row)
#pragma unroll
            for (int a = 0; a < 7; ++a)
#pragma unroll
              for (int b = a; b < 7; ++b) G[tt(a, b)] += w[row][a] * w[row][b];
        } else {  // M-estimator: cost += l, the point's J^T J and J^T r are scaled by s (robust_norms.h:20-26)
          const T n2 = w[0][6] * w[0][6] + w[1][6] * w[1][6];
          T l, s;
          robust_norm(loss, n2, th2, l, s);
          csum += l;
          inl += n2 <= th2 ? T(2) : T(0);
#pragma unroll
          for (int row = 0; row < 2; ++row)
#pragma unroll
            for (int a = 0; a < 6; ++a) {
              const T sw = s * w[row][a];
#pragma unroll
              for (int b = a; b < 7; ++b) G[tt(a, b)] += sw * w[row][b];
            }
        }
      } else {
        const T n2 = w[0][6] * w[0][6] + w[1][6] * w[1][6];
        if (loss == TOA_LOSS_L2) csum += n2;
        else {
          T l, s;
          robust_norm(loss, n2, th2, l, s);
          csum += l;
          inl += n2 <= th2 ? T(2) : T(0);
        }
      }
    }
    if (loss == TOA_LOSS_L2) ninl = 2 * (pt1 - pt0);
    else ninl = int(wave_allreduce_sum(inl));
    if (WANT_H) {
      wave_allreduce_many(G, lane);
      if (loss == TOA_LOSS_L2) return G[tt(6, 6)];
    }
    return wave_allreduce_sum(csum);
  }
  __device__ __forceinline__ void accumulate(WaveLds<T>& L, int, int lane, T& cost, int& nres) {
    cost = pass<true>(L, lane);
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < 6; ++a) { L.g[a] = G[tt(a, 6)]; L.hd[a] = G[tt(a, a)]; }
    }
    nres = 2 * npts;
    wave_sync();
  }
  __device__ __forceinline__ void evaluate(WaveLds<T>& L, int, int lane, T& cost, int& nres) {
    cost = pass<false>(L, lane);
    nres = 2 * npts;
  }
  template <typename O>
  __device__ __forceinline__ void write_sym(O* M, int LD, int, int lane) const {
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = a; b < 6; ++b) { M[a * LD + b] = O(G[tt(a, b)]); M[b * LD + a] = O(G[tt(a, b)]); }
    }
  }
  __device__ __forceinline__ void plus_eq(WaveLds<T>& L, const T* dv, T sign, int n, int lane) const {
    Se3Manifold<T>::plus_eq(L, dv, sign, n, lane);
  }
};

}  // namespace toa
