// TEST INFRASTRUCTURE ONLY.  Bundle adjustment as the REFERENCE would solve it: `Optimize(x, acc)` over ONE parameter
// object x = (C SE3 poses, N 3-D points) with the FULL dense (6C + 3N)^2 Hessian and the dense LDL^T of SolveLDLT
// (include/tinyopt/math.h:232-240) — tinyopt has no Schur path; its sparse alternative is Eigen's SimplicialLDLT on the
// same system (math.h:266-277, README.md:30,165-167 "slow").  The device path eliminates the points (Schur complement on
// the reduced camera system); mathematically the step is the same, so this dense restatement is its oracle.
//
//   residual of observation (c, j):  r = pi(R_c p_j + t_c) - uv_cj,  pi(X, Y, Z) = (f X/Z + cx, f Y/Z + cy)
//   pose Jacobian  (right perturbation at delta = 0, sophus.h:24-26):  d p_c / d(upsilon, omega) = [ R | -R [p]x ]
//   point Jacobian:  d p_c / d p = R
//   tangent order: cameras first (6 each, Sophus order upsilon, omega), then points (3 each)
//   update: pose_c <- pose_c * exp(delta_c)  (SE3),  p_j <- p_j + delta_j  (Euclidean, traits.h:184-190)
// data: [f cx cy 0 0 0 0 0 | uv: C x N x 2 | vis: C x N (1 = observed, 0 = not)].
#pragma once
#include <vector>

#include "lm_oracle.hpp"
#include "robust.hpp"
#include "se3.hpp"

namespace oracle {
namespace ba {

template <typename T>
struct Params {
  int C = 0, N = 0;
  std::vector<T> v;  // [12 C poses (R row-major, t) | 3 N points]
};

template <typename T>
struct Plus {
  void operator()(Params<T>& x, const std::vector<T>& dx, T sign) const {
    for (int c = 0; c < x.C; ++c) {
      se3::Pose<T> p;
      for (int i = 0; i < 12; ++i) p[i] = x.v[12 * c + i];
      std::vector<T> d(dx.begin() + 6 * c, dx.begin() + 6 * c + 6);
      se3::plus_eq(p, d, sign);
      for (int i = 0; i < 12; ++i) x.v[12 * c + i] = p[i];
    }
    for (int j = 0; j < 3 * x.N; ++j) x.v[12 * x.C + j] += sign * dx[6 * x.C + j];
  }
};

// loss / th2: an M-estimator on each observation's squared norm n2 = |r|^2 (what a tinyopt user writes inside the cost functor,
// losses/robust_norms.h:20-26 "JtJ * dx = Jt*res*s", docs/API.md:396-411): cost += l(n2), the observation's J^T J and J^T r
// scaled by s = dl/dn2, both of its residuals inliers when n2 <= th2 (cost.h:84-95).  loss 0 = plain squared L2.
template <typename T>
struct Acc {
  int C, N;
  const T* data;
  int loss = 0;
  T th2 = T(0);
  // r (2), Jc (2x6 row-major), Jp (2x3 row-major) of one observation
  void obs(const Params<T>& x, int c, int j, T* r, T* Jc, T* Jp) const {
    const T* P = x.v.data() + 12 * c;
    const T* q = x.v.data() + 12 * C + 3 * j;
    const T* uv = data + 8 + (size_t(c) * N + j) * 2;
    const T f = data[0], cx = data[1], cy = data[2];
    const T X = P[0] * q[0] + P[1] * q[1] + P[2] * q[2] + P[9];
    const T Y = P[3] * q[0] + P[4] * q[1] + P[5] * q[2] + P[10];
    const T Z = P[6] * q[0] + P[7] * q[1] + P[8] * q[2] + P[11];
    const T iz = T(1) / Z;
    r[0] = f * X * iz + cx - uv[0];
    r[1] = f * Y * iz + cy - uv[1];
    if (!Jc) return;
    const T du[3] = {f * iz, T(0), -f * X * iz * iz};
    const T dv[3] = {T(0), f * iz, -f * Y * iz * iz};
    T D[3][6];
    for (int a = 0; a < 3; ++a) {
      D[a][0] = P[3 * a]; D[a][1] = P[3 * a + 1]; D[a][2] = P[3 * a + 2];
      D[a][3] = -(P[3 * a + 1] * q[2] - P[3 * a + 2] * q[1]);
      D[a][4] = -(-P[3 * a] * q[2] + P[3 * a + 2] * q[0]);
      D[a][5] = -(P[3 * a] * q[1] - P[3 * a + 1] * q[0]);
    }
    for (int k = 0; k < 6; ++k) {
      Jc[k] = du[0] * D[0][k] + du[2] * D[2][k];
      Jc[6 + k] = dv[1] * D[1][k] + dv[2] * D[2][k];
    }
    for (int k = 0; k < 3; ++k) {
      Jp[k] = du[0] * P[k] + du[2] * P[6 + k];
      Jp[3 + k] = dv[1] * P[3 + k] + dv[2] * P[6 + k];
    }
  }
  Cost operator()(const Params<T>& x, T* g, T* H) const {
    const int n = 6 * C + 3 * N;
    const T* vis = data + 8 + size_t(C) * N * 2;
    T cost = 0;
    int nres = 0, ninl = 0;
    for (int c = 0; c < C; ++c)
      for (int j = 0; j < N; ++j) {
        if (vis[size_t(c) * N + j] == T(0)) continue;
        T r[2], Jc[12], Jp[6];
        obs(x, c, j, r, g ? Jc : nullptr, g ? Jp : nullptr);
        const T n2 = r[0] * r[0] + r[1] * r[1];
        const auto ls = robust::Apply<T>(loss, n2, th2);   // loss 0: {n2, 1}
        cost += ls.l;
        nres += 2;
        ninl += (loss == 0 || n2 <= th2) ? 2 : 0;
        if (!g) continue;
        int idx[9];
        for (int k = 0; k < 6; ++k) idx[k] = 6 * c + k;
        for (int k = 0; k < 3; ++k) idx[6 + k] = 6 * C + 3 * j + k;
        for (int row = 0; row < 2; ++row) {
          T J9[9];
          for (int k = 0; k < 6; ++k) J9[k] = Jc[6 * row + k];
          for (int k = 0; k < 3; ++k) J9[6 + k] = Jp[3 * row + k];
          for (int a = 0; a < 9; ++a) {
            const T sJ = ls.s * J9[a];
            g[idx[a]] += sJ * r[row];
            if (H) for (int b = 0; b < 9; ++b) H[size_t(idx[b]) * n + idx[a]] += sJ * J9[b];
          }
        }
      }
    return Cost(double(cost), nres, nres ? float(ninl) / float(nres) : 1.0f);
  }
};

}  // namespace ba
}  // namespace oracle
