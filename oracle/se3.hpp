// TEST INFRASTRUCTURE ONLY.  SE3 pieces of the oracle: the manifold update of
// include/tinyopt/3rdparty/traits/sophus.h:24-26 (`pose *= SE3::exp(delta)`, right update, tangent
// order (upsilon, omega) = translation first) and the pinhole reprojection residual of SURVEY §8(d) C5.
//
// Sophus is an un-vendored, unpinned dependency (cmake/ThirdParties.cmake:74-77, GIT_TAG main) absent from
// the image; its published algorithm is restated: SO3::exp via Rodrigues with the small-angle series,
// SE3::exp = (exp(omega), V(omega) upsilon), V = I + (1-cos t)/t^2 [w]x + (t-sin t)/t^3 [w]x^2.
// Poses are stored as a rotation matrix (row-major 9) + translation (3) = 12 scalars.
// The reference holds no reprojection residual (only a 6-residual pose prior, tests/sophus.cpp:26-44), so this
// model is pinned by (i) group properties of exp, (ii) a finite-difference check of the right-perturbation
// Jacobian in the spirit of diff/gradient_check.h, (iii) recovery of the planted pose.
#pragma once
#include <array>
#include <cmath>
#include <vector>

#include "jet.hpp"
#include "lm_oracle.hpp"
#include "robust.hpp"

namespace oracle {
namespace se3 {

template <typename T>
using Pose = std::array<T, 12>;  // R (row-major 3x3), t

template <typename T>
inline void so3_exp(const T* w, T* R) {
  const T t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const T t = std::sqrt(t2);
  T A, B;  // A = sin t / t, B = (1 - cos t) / t^2
  if (t2 < T(1e-10)) { A = T(1) - t2 / T(6); B = T(0.5) - t2 / T(24); }
  else { A = std::sin(t) / t; B = (T(1) - std::cos(t)) / t2; }
  const T wx = w[0], wy = w[1], wz = w[2];
  R[0] = T(1) - B * (wy * wy + wz * wz); R[1] = -A * wz + B * wx * wy;          R[2] = A * wy + B * wx * wz;
  R[3] = A * wz + B * wx * wy;          R[4] = T(1) - B * (wx * wx + wz * wz); R[5] = -A * wx + B * wy * wz;
  R[6] = -A * wy + B * wx * wz;         R[7] = A * wx + B * wy * wz;          R[8] = T(1) - B * (wx * wx + wy * wy);
}

// exp of a twist delta = (upsilon, omega): returns (Rd, td)
template <typename T>
inline void exp(const T* d, T* Rd, T* td) {
  const T* u = d;
  const T* w = d + 3;
  so3_exp(w, Rd);
  const T t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const T t = std::sqrt(t2);
  T B, C;  // V = I + B [w]x + C [w]x^2
  if (t2 < T(1e-10)) { B = T(0.5) - t2 / T(24); C = T(1) / T(6) - t2 / T(120); }
  else { B = (T(1) - std::cos(t)) / t2; C = (t - std::sin(t)) / (t2 * t); }
  const T wx = w[0], wy = w[1], wz = w[2];
  // [w]x u and [w]x [w]x u
  const T c1[3] = {wy * u[2] - wz * u[1], wz * u[0] - wx * u[2], wx * u[1] - wy * u[0]};
  const T c2[3] = {wy * c1[2] - wz * c1[1], wz * c1[0] - wx * c1[2], wx * c1[1] - wy * c1[0]};
  for (int i = 0; i < 3; ++i) td[i] = u[i] + B * c1[i] + C * c2[i];
}

// pose <- pose * exp(sign * delta)   (sophus.h:24-26)
template <typename T>
inline void plus_eq(Pose<T>& x, const std::vector<T>& delta, T sign) {
  T d[6], Rd[9], td[3];
  for (int i = 0; i < 6; ++i) d[i] = sign * delta[i];
  exp(d, Rd, td);
  T R[9], t[3];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) R[3 * i + j] = x[3 * i] * Rd[j] + x[3 * i + 1] * Rd[3 + j] + x[3 * i + 2] * Rd[6 + j];
    t[i] = x[3 * i] * td[0] + x[3 * i + 1] * td[1] + x[3 * i + 2] * td[2] + x[9 + i];
  }
  for (int i = 0; i < 9; ++i) x[i] = R[i];
  for (int i = 0; i < 3; ++i) x[9 + i] = t[i];
}
template <typename T>
struct Plus {
  void operator()(Pose<T>& x, const std::vector<T>& dx, T sign) const { plus_eq(x, dx, sign); }
};

// ---- log maps, written once over the scalar type S (plain T for the cost-only callback, Jet<T, 6> for the
//      differentiated one), from a rotation MATRIX.  Sophus (un-vendored, unpinned: cmake/ThirdParties.cmake:74-77)
//      computes the same maps through a unit quaternion; the published formulas are restated:
//        SO3:  omega = (theta / sin theta) * vee(R - R^T)/2,   cos theta = (tr R - 1)/2
//        SE3:  upsilon = V^-1 t,  V^-1 = I - 1/2 [w]x + (1 - theta cos(theta/2) / (2 sin(theta/2))) / theta^2 [w]x^2
//      Near the identity (cos theta > 0.999) both coefficients are evaluated by their power series in sin^2 theta /
//      theta^2, which are smooth there — a square root of a vanishing quantity would make every Jet derivative
//      infinite exactly at the solution of a pose prior.  Domain: theta < ~3 rad (the branch at pi is not needed).
using std::sqrt; using std::sin; using std::cos; using std::atan2;

template <typename S, typename T>
inline void so3_log(const S* R, S* w, S& theta2, bool& small) {
  const S c = (R[0] + R[4] + R[8] - T(1.0)) * T(0.5);
  const S v[3] = {(R[7] - R[5]) * T(0.5), (R[2] - R[6]) * T(0.5), (R[3] - R[1]) * T(0.5)};  // sin(theta) * axis
  const S s2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  small = scalar_part(c) > T(0.999);
  S k;  // theta / sin(theta)
  if (small) {  // asin(x)/x = 1 + x^2/6 + 3x^4/40 + 5x^6/112 + 35x^8/1152, x = sin(theta)
    k = T(1.0) + s2 * (T(1.0 / 6.0) + s2 * (T(3.0 / 40.0) + s2 * (T(5.0 / 112.0) + s2 * T(35.0 / 1152.0))));
  } else {
    const S sn = sqrt(s2);
    k = atan2(sn, c) / sn;
  }
  for (int i = 0; i < 3; ++i) w[i] = v[i] * k;
  theta2 = s2 * k * k;
}

// xi = (upsilon, omega) = log of the pose (R, t)
template <typename S, typename T>
inline void se3_log(const S* R, const S* t, S* xi) {
  S w[3], th2;
  bool small;
  so3_log<S, T>(R, w, th2, small);
  S coef;
  if (small) {  // (1 - (theta/2) cot(theta/2)) / theta^2 = 1/12 + theta^2/720 + theta^4/30240 + theta^6/1209600
    coef = T(1.0 / 12.0) + th2 * (T(1.0 / 720.0) + th2 * (T(1.0 / 30240.0) + th2 * T(1.0 / 1209600.0)));
  } else {
    const S th = sqrt(th2), h = th * T(0.5);
    coef = (T(1.0) - th * cos(h) / (T(2.0) * sin(h))) / th2;
  }
  const S c1[3] = {w[1] * t[2] - w[2] * t[1], w[2] * t[0] - w[0] * t[2], w[0] * t[1] - w[1] * t[0]};        // w x t
  const S c2[3] = {w[1] * c1[2] - w[2] * c1[1], w[2] * c1[0] - w[0] * c1[2], w[0] * c1[1] - w[1] * c1[0]};  // w x (w x t)
  for (int i = 0; i < 3; ++i) {
    xi[i] = t[i] - c1[i] * T(0.5) + coef * c2[i];
    xi[3 + i] = w[i];
  }
}

// Pose prior (tests/sophus.cpp:26-44): residual(x) = log(prior_inv * x) in R^6, differentiated by Jets over the
// right perturbation x * exp(delta) at delta = 0 (optimize_autodiff.h:48-77; sophus.h:24-26).  exp(delta) enters the
// Jets through its first-order part I + [omega]x, upsilon — exact for first derivatives at 0.
template <typename T>
struct PosePriorAcc {
  const T* Pinv;  // R (row-major 9), t (3)
  template <typename S>
  void residual(const S* Rx, const S* tx, S* xi) const {
    S RA[9], tA[3];
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) RA[3 * i + j] = Rx[j] * Pinv[3 * i] + Rx[3 + j] * Pinv[3 * i + 1] + Rx[6 + j] * Pinv[3 * i + 2];
      tA[i] = tx[0] * Pinv[3 * i] + tx[1] * Pinv[3 * i + 1] + tx[2] * Pinv[3 * i + 2] + Pinv[9 + i];
    }
    se3_log<S, T>(RA, tA, xi);
  }
  Cost operator()(const Pose<T>& x, T* g, T* H) const {
    if (!g) {
      T xi[6];
      residual<T>(x.data(), x.data() + 9, xi);
      T c = 0;
      for (int i = 0; i < 6; ++i) c += xi[i] * xi[i];
      return Cost(double(c), 6);
    }
    using J6 = Jet<T, 6>;
    J6 d[6];
    for (int k = 0; k < 6; ++k) d[k] = J6(T(0), k);  // delta = (upsilon, omega) seeded at 0
    const T* R = x.data();
    J6 Rj[9], tj[3];
    for (int i = 0; i < 3; ++i) {  // R (I + [omega]x): column j of [omega]x is omega x e_j
      Rj[3 * i + 0] = R[3 * i + 0] + (d[5] * R[3 * i + 1] - d[4] * R[3 * i + 2]);
      Rj[3 * i + 1] = R[3 * i + 1] + (d[3] * R[3 * i + 2] - d[5] * R[3 * i + 0]);
      Rj[3 * i + 2] = R[3 * i + 2] + (d[4] * R[3 * i + 0] - d[3] * R[3 * i + 1]);
      tj[i] = x[9 + i] + (d[0] * R[3 * i] + d[1] * R[3 * i + 1] + d[2] * R[3 * i + 2]);
    }
    J6 xi[6];
    residual<J6>(Rj, tj, xi);
    T r[6], J[36];
    for (int i = 0; i < 6; ++i) {
      r[i] = xi[i].a;
      for (int k = 0; k < 6; ++k) J[6 * i + k] = xi[i].v[k];
    }
    return AccumulateFromJ<T>(6, 6, r, J, g, H);
  }
};

// Reprojection residuals of npts points: r = (f X/Z + cx - u, f Y/Z + cy - v), p_c = R p + t;
// Jacobian w.r.t. the right perturbation at delta = 0: d p_c/d upsilon = R, d p_c/d omega = -R [p]x.
// data per point: [x y z u v]; folded as the AD bridge folds a residual vector (optimize_autodiff.h:123-164).
// Robust variant: header slots intr[3] = loss kind (robust::Kind), intr[4] = th2.  Per point the
// squared norm n2 = ||r||^2 goes through the M-estimator: cost += l, and the scale s multiplies the point's
// contribution to the normal equations "JtJ * dx = Jt*res*s" (robust_norms.h:20-26); a point is an inlier when
// n2 <= th2 (the inlier branch of Truncated/Huber/Tukey), reported through Cost::inlier_ratio (cost.h:84-95).
template <typename T>
struct ReprojAcc {
  int npts;
  const T* intr;  // f, cx, cy
  const T* pts;   // [npts][5]
  void row_pair(const Pose<T>& x, int i, T* r, T* J /*2x6 row-major*/) const {
    const T* q = pts + size_t(i) * 5;
    const T* R = x.data();
    const T X = R[0] * q[0] + R[1] * q[1] + R[2] * q[2] + x[9];
    const T Y = R[3] * q[0] + R[4] * q[1] + R[5] * q[2] + x[10];
    const T Z = R[6] * q[0] + R[7] * q[1] + R[8] * q[2] + x[11];
    const T f = intr[0], iz = T(1) / Z;
    r[0] = f * X * iz + intr[1] - q[3];
    r[1] = f * Y * iz + intr[2] - q[4];
    if (!J) return;
    // d(u,v)/d p_c
    const T du[3] = {f * iz, T(0), -f * X * iz * iz};
    const T dv[3] = {T(0), f * iz, -f * Y * iz * iz};
    // d p_c / d delta = [ R | -R [p]x ],  [p]x = [[0,-z,y],[z,0,-x],[-y,x,0]]
    T D[3][6];
    for (int a = 0; a < 3; ++a) {
      D[a][0] = R[3 * a]; D[a][1] = R[3 * a + 1]; D[a][2] = R[3 * a + 2];
      D[a][3] = -(R[3 * a + 1] * q[2] - R[3 * a + 2] * q[1]);
      D[a][4] = -(-R[3 * a] * q[2] + R[3 * a + 2] * q[0]);
      D[a][5] = -(R[3 * a] * q[1] - R[3 * a + 1] * q[0]);
    }
    for (int c = 0; c < 6; ++c) {
      J[c] = du[0] * D[0][c] + du[1] * D[1][c] + du[2] * D[2][c];
      J[6 + c] = dv[0] * D[0][c] + dv[1] * D[1][c] + dv[2] * D[2][c];
    }
  }
  Cost operator()(const Pose<T>& x, T* g, T* H) const {
    T c = 0;
    const int kind = int(intr[3]);
    const T th2 = intr[4];
    int inliers = 0;
    for (int i = 0; i < npts; ++i) {
      T r[2], J[12];
      row_pair(x, i, r, g ? J : nullptr);
      const T n2 = r[0] * r[0] + r[1] * r[1];
      const auto ls = robust::Apply(kind, n2, th2);  // kind 0: {n2, 1}
      c += ls.l;
      inliers += (kind == 0 || n2 <= th2) ? 2 : 0;
      if (g) {
        for (int row = 0; row < 2; ++row) {
          const T* Jr = J + 6 * row;
          for (int a = 0; a < 6; ++a) {
            const T sJ = ls.s * Jr[a];
            g[a] += sJ * r[row];
            if (H) for (int b = 0; b < 6; ++b) H[size_t(b) * 6 + a] += sJ * Jr[b];
          }
        }
      }
    }
    return Cost(double(c), 2 * npts, npts ? float(inliers) / float(2 * npts) : 1.0f);
  }
};

}  // namespace se3
}  // namespace oracle
