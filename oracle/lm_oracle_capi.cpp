// TEST INFRASTRUCTURE ONLY.  C entry points (ctypes) over oracle/lm_oracle.hpp for tests/,
// bench.py's cpu_baseline leg and __graft_entry__.smoke().  Never linked into the product.
// Uses the POD `toa_options` from include/tinyopt_amd.h purely as a data contract.
#include <chrono>
#include <cstring>
#include <vector>

#include "../include/tinyopt_amd.h"
#include "lm_oracle.hpp"
#include "robust.hpp"
#include "ba.hpp"
#include "se3.hpp"
#include "testfns.hpp"
#include "synth.hpp"

#ifdef _OPENMP
#include <omp.h>
#endif

using namespace oracle;

static Options from_pod(const toa_options& p) {
  Options o;
  o.solver_type = p.solver_type;
  o.max_iters = uint16_t(p.max_iters);
  o.min_error = p.min_error;
  o.min_rerr_dec = p.min_rerr_dec;
  o.min_step_norm2 = p.min_step_norm2;
  o.min_grad_norm2 = p.min_grad_norm2;
  o.max_total_failures = uint8_t(p.max_total_failures);
  o.max_consec_failures = uint8_t(p.max_consec_failures);
  o.damping_init = p.damping_init;
  o.damping_range = {{p.damping_min, p.damping_max}};
  o.good_factor = p.good_factor;
  o.bad_factor = p.bad_factor;
  o.grad_clipping = p.grad_clipping;
  o.check_min_H_diag = p.check_min_H_diag;
  o.check_final_cost = p.check_final_cost;
  o.use_step_quality_approx = p.use_step_quality_approx;
  o.use_ldlt = p.use_ldlt;
  o.H_is_full = p.H_is_full;
  o.save_last = p.save_last;
  o.use_squared_norm = p.use_squared_norm;
  o.downscale_by_2 = p.downscale_by_2;
  o.normalize = p.normalize;
  return o;
}

// ---- DenseRow model: r_i = a_i.x + 0.1 sin(a_i.x) - b_i, J_i = (1 + 0.1 cos(a_i.x)) a_i.
// Folded exactly like the AD bridge folds a residual VECTOR (diff/optimize_autodiff.h:123-164):
// grad = J^T r, H = J^T J (full), cost = (||r||^2, m); all arithmetic in T.
// Loop order is i-outer (rank-1 updates) — same per-entry summation order as a naive
// triple loop, but vectorisable, so the timed CPU baseline is not artificially slow.
// M-estimator applied by the DenseRow / circle-fit accumulators below (set by oracle_set_loss): what a tinyopt user does
// by wrapping the residual's squared norm in `losses::Huber(n2, th2, true)` inside the cost functor
// (losses/robust_norms.h:20-26, docs/API.md:396-411): cost += l, the residual's J^T J and J^T r scaled by s = dl/dn2,
// inliers = residuals with n2 <= th2 (cost.h:84-95).  kind 0 = plain squared L2.
static int g_loss_kind = 0;
static double g_loss_th2 = 0;
static float* g_inlier_out = nullptr;   // [P] final inlier ratios of the next oracle_dense_row_lm call (nullptr = not wanted)

template <typename T>
struct DenseRowAcc {
  int n, m;
  const T* A;  // m×n row-major
  const T* b;  // m
  mutable std::vector<T> Jrow;
  mutable std::vector<double> gd, Hd;   // (ORACLE_GRAM_DOUBLE builds only)
  DenseRowAcc(int n_, int m_, const T* A_, const T* b_) : n(n_), m(m_), A(A_), b(b_), Jrow(n_) {}
  Cost operator()(const std::vector<T>& x, T* g, T* H) const {
    T c = 0;
    const int kind = g_loss_kind;
    const T th2 = T(g_loss_th2);
    int inliers = 0;
    // ORACLE_COST_LANES (default 1 = the sequential sum this oracle is pinned with): tools/iter_inflation.py builds copies with
    // L > 1 partial sums folded by a tree — the shape of Eigen's packet reduction and of the device's blocked sums — or with a
    // double accumulator (ORACLE_COST_DOUBLE), to show how the iteration count at the float floor depends on the ORDER of
    // this one sum.  Never defined in the library the tests load.
#ifndef ORACLE_COST_LANES
#define ORACLE_COST_LANES 1
#endif
    T cl[ORACLE_COST_LANES] = {};
    double cd = 0;
    for (int i = 0; i < m; ++i) {
      const T* a = A + size_t(i) * n;
      T t = 0;
#if defined(ORACLE_DOT_DOUBLE)     // (tools/iter_inflation.py only: a_i . x summed in double, rounded once)
      { double td = 0; for (int j = 0; j < n; ++j) td += double(a[j]) * double(x[j]); t = T(td); }
#elif defined(ORACLE_DOT_TREE)     // (... or as a 16-lane tree of partial sums, the shape of the device's reduction)
      { T tl[16] = {}; for (int j = 0; j < n; ++j) tl[j & 15] = std::fma(a[j], x[j], tl[j & 15]);
        for (int w2 = 8; w2 >= 1; w2 /= 2) for (int k = 0; k < w2; ++k) tl[k] += tl[k + w2];
        t = tl[0]; }
#else
      for (int j = 0; j < n; ++j) t += a[j] * x[j];
#endif
      const T r = t + T(0.1) * std::sin(t) - b[i];
      T w = T(1);
      if (kind == 0) {
#if defined(ORACLE_COST_DOUBLE)
        cd += double(r) * double(r);
#elif defined(ORACLE_COST_BY_PASS_KIND)   // the tree sum on passes that want the gradient, the sequential one on cost-only passes
        if (g) cl[i % ORACLE_COST_LANES] += r * r; else c += r * r;
#elif ORACLE_COST_LANES > 1
        cl[i % ORACLE_COST_LANES] += r * r;
#else
        c += r * r;
#endif
      } else {
        const auto ls = robust::Apply<T>(kind, r * r, th2);
        c += ls.l;
        w = ls.s;
        inliers += (r * r <= th2) ? 1 : 0;
      }
      if (g) {
        const T s = T(1) + T(0.1) * std::cos(t);
        T* J = Jrow.data();
        for (int j = 0; j < n; ++j) J[j] = s * a[j];
#if defined(ORACLE_GRAM_DOUBLE)   // (tools/iter_inflation.py only: g and H summed in double, rounded to T once at the end)
        if (gd.empty()) { gd.assign(n, 0.0); Hd.assign(size_t(n) * n, 0.0); }
        for (int j = 0; j < n; ++j) gd[j] += double(w * J[j] * r);
        if (H)
          for (int q = 0; q < n; ++q) {
            const T Jq = w * J[q];
            for (int p = 0; p < n; ++p) Hd[size_t(q) * n + p] += double(J[p] * Jq);
          }
#else
        for (int j = 0; j < n; ++j) g[j] += w * J[j] * r;
        if (H) {
          for (int q = 0; q < n; ++q) {  // column q of col-major H: H[q*n + p] += J[p]*J[q]
            const T Jq = w * J[q];
            T* Hq = H + size_t(q) * n;
            for (int p = 0; p < n; ++p) Hq[p] += J[p] * Jq;
          }
        }
#endif
      }
    }
#if defined(ORACLE_GRAM_DOUBLE)
    if (g && !gd.empty()) {
      for (int j = 0; j < n; ++j) g[j] += T(gd[j]);
      if (H) for (size_t e = 0; e < size_t(n) * n; ++e) H[e] += T(Hd[e]);
      gd.clear(); Hd.clear();
    }
#endif
    if (kind == 0) {
#if defined(ORACLE_COST_DOUBLE)
      c = T(cd);
#elif defined(ORACLE_COST_BY_PASS_KIND)
      if (g) {
        for (int w = ORACLE_COST_LANES / 2; w >= 1; w /= 2)
          for (int k = 0; k < w; ++k) cl[k] += cl[k + w];
        c = cl[0];
      }
#elif ORACLE_COST_LANES > 1
      for (int w = ORACLE_COST_LANES / 2; w >= 1; w /= 2)
        for (int k = 0; k < w; ++k) cl[k] += cl[k + w];
      c = cl[0];
#endif
    }
    (void)cl; (void)cd;
    return kind == 0 ? Cost(double(c), m) : Cost(double(c), m, m ? float(inliers) / float(m) : 1.0f);
  }
};

// ---- GaussianPrior, the manual callback of benchmarks/dense.cpp:57-66 / :90-99:
// res = (x-y)/sigma; grad = J*res with J = diag(1/sigma); H.diagonal() = sigma^-2;
// returns res.squaredNorm() as a SCALAR => Cost(v, 1) (cost.h:22).
template <typename T>
struct GaussianPriorAcc {
  int n;
  const T* y;
  const T* sigma;
  Cost operator()(const std::vector<T>& x, T* g, T* H) const {
    T c = 0;
    for (int j = 0; j < n; ++j) {
      const T res = (x[j] - y[j]) / sigma[j];
      c += res * res;
      if (g) {
        g[j] = (T(1) / sigma[j]) * res;
        const T is = T(1) / sigma[j];
        H[size_t(j) * n + j] = is * is;
      }
    }
    return Cost(double(c));
  }
};

template <typename T>
static void run_batch_dense_row(int64_t P, int n, int m, const T* A, const T* b, T* x, const Options& o,
                                int32_t* stop, int32_t* iters, int32_t* fails, double* cost, double* rerr,
                                double* finalH, double* errs, double* deltas2, uint8_t* succ, int hist_stride,
                                int nthreads) {
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (int64_t p = 0; p < P; ++p) {
    DenseRowAcc<T> acc(n, m, A + size_t(p) * m * n, b + size_t(p) * m);
    std::vector<T> xv(x + p * n, x + (p + 1) * n);
    Optimizer<T> opt(o, n);
    Output out = opt.OptimizeAcc(xv, acc, EuclidPlus<T>());
    std::memcpy(x + p * n, xv.data(), sizeof(T) * n);
    if (stop) stop[p] = out.stop_reason;
    if (iters) iters[p] = out.num_iters;
    if (fails) fails[p] = out.num_failures;
    if (cost) cost[p] = out.final_cost.cost;
    if (rerr) rerr[p] = out.final_rerr_dec;
    if (g_inlier_out) g_inlier_out[p] = out.final_cost.inlier_ratio;
    if (finalH && !out.final_hessian.empty())
      std::memcpy(finalH + size_t(p) * n * n, out.final_hessian.data(), sizeof(double) * n * n);
    if (errs) {
      for (size_t k = 0; k < out.errs.size() && int(k) < hist_stride; ++k) {
        errs[size_t(p) * hist_stride + k] = out.errs[k];
        if (deltas2) deltas2[size_t(p) * hist_stride + k] = out.deltas2[k];
        if (succ) succ[size_t(p) * hist_stride + k] = out.successes[k];
      }
    }
  }
  (void)nthreads;
}

template <typename T>
static void se3_lm_t(int64_t P, int npts, const T* data, T* poses, const Options& o, int32_t* stop, int32_t* iters,
                     double* cost, double* finalH, float* inlier_ratio) {
  for (int64_t p = 0; p < P; ++p) {
    const T* d = data + size_t(p) * (8 + 5 * size_t(npts));
    se3::ReprojAcc<T> acc{npts, d, d + 8};
    se3::Pose<T> x;
    for (int i = 0; i < 12; ++i) x[i] = poses[p * 12 + i];
    Optimizer<T> opt(o, 6);
    Output out = opt.OptimizeAcc(x, acc, se3::Plus<T>());
    for (int i = 0; i < 12; ++i) poses[p * 12 + i] = x[i];
    if (stop) stop[p] = out.stop_reason;
    if (iters) iters[p] = out.num_iters;
    if (cost) cost[p] = out.final_cost.cost;
    if (inlier_ratio) inlier_ratio[p] = out.final_cost.inlier_ratio;
    if (finalH && !out.final_hessian.empty()) std::memcpy(finalH + size_t(p) * 36, out.final_hessian.data(), sizeof(double) * 36);
  }
}
// AccumulateFromJ (oracle/lm_oracle.hpp) with the M-estimator of oracle_set_loss on every scalar residual: what the AD bridge
// computes when the user's functor returns residuals already passed through `losses::X(n2, th2, true)`.
template <typename T>
static Cost AccumulateFromJLoss(int m, int n, const T* r, const T* J, T* g, T* H) {
  if (g_loss_kind == 0) return AccumulateFromJ<T>(m, n, r, J, g, H);
  const T th2 = T(g_loss_th2);
  T c = 0;
  int inliers = 0;
  if (g) std::fill(g, g + n, T(0));
  if (g && H) std::fill(H, H + size_t(n) * n, T(0));
  for (int i = 0; i < m; ++i) {
    const auto ls = robust::Apply<T>(g_loss_kind, r[i] * r[i], th2);
    c += ls.l;
    inliers += (r[i] * r[i] <= th2) ? 1 : 0;
    if (g) {
      for (int a = 0; a < n; ++a) {
        const T sw = ls.s * J[size_t(i) * n + a];
        g[a] += sw * r[i];
        if (H) for (int b2 = 0; b2 < n; ++b2) H[size_t(b2) * n + a] += sw * J[size_t(i) * n + b2];
      }
    }
  }
  return Cost(double(c), m, m ? float(inliers) / float(m) : 1.0f);
}

// Gaussian prior with a GENERAL covariance, whitened by the upper Cholesky factor U of the information matrix:
// res = U (x - y), J = U  (losses/mahalanobis.h:160-171 MahaWhitenedInfoU; the AD form of tests/cov.cpp:127-146
// `res = Lt * (x - y)`), folded as the AD bridge folds a residual vector: grad = J^T res, H = J^T J, cost = ||res||^2
// over n residuals.  data: [P][n + n*n] = y, then U row-major (strictly upper triangular + diagonal).
template <typename T>
struct MahaPriorAcc {
  int n;
  const T* y;
  const T* U;
  Cost operator()(const std::vector<T>& x, T* g, T* H) const {
    std::vector<T> r(n), J(size_t(n) * n);
    for (int i = 0; i < n; ++i) {
      T s = 0;
      for (int j = i; j < n; ++j) s += U[size_t(i) * n + j] * (x[j] - y[j]);  // triangularView<Upper>
      r[i] = s;
      for (int j = 0; j < n; ++j) J[size_t(i) * n + j] = U[size_t(i) * n + j];
    }
    return AccumulateFromJ<T>(n, n, r.data(), J.data(), g, H);
  }
};
template <typename T>
static void maha_prior_lm_t(int64_t P, int n, const T* data, T* x, const Options& o, int32_t* stop, int32_t* iters,
                            double* cost, double* finalH) {
  for (int64_t p = 0; p < P; ++p) {
    const T* d = data + size_t(p) * (n + size_t(n) * n);
    std::vector<T> xv(x + p * n, x + p * n + n);
    Optimizer<T> opt(o, n);
    Output out = opt.OptimizeAcc(xv, MahaPriorAcc<T>{n, d, d + n}, EuclidPlus<T>());
    std::memcpy(x + p * n, xv.data(), sizeof(T) * n);
    if (stop) stop[p] = out.stop_reason;
    if (iters) iters[p] = out.num_iters;
    if (cost) cost[p] = out.final_cost.cost;
    if (finalH && !out.final_hessian.empty()) std::memcpy(finalH + size_t(p) * n * n, out.final_hessian.data(), sizeof(double) * n * n);
  }
}
// Bundle adjustment through the reference's own route: the FULL dense (6C + 3N)^2 system + dense LDL^T (oracle/ba.hpp).
// x: [P][12 C + 3 N] in place; data: [P][8 + 3 C N]; history arrays as for the other families.
template <typename T>
static void ba_lm_t(int64_t P, int C, int N, const T* data, T* x, const Options& o, int32_t* stop, int32_t* iters, int32_t* fails,
                    double* cost, int32_t* nres, double* errs, double* deltas2, uint8_t* succ, int hs) {
  const size_t xs = size_t(12) * C + size_t(3) * N, ds = size_t(8) + size_t(3) * C * N;
  for (int64_t p = 0; p < P; ++p) {
    ba::Params<T> X;
    X.C = C; X.N = N;
    X.v.assign(x + p * xs, x + (p + 1) * xs);
    Optimizer<T> opt(o, 6 * C + 3 * N);
    Output out = opt.OptimizeAcc(X, ba::Acc<T>{C, N, data + p * ds, g_loss_kind, T(g_loss_th2)}, ba::Plus<T>());
    if (g_inlier_out) g_inlier_out[p] = out.final_cost.inlier_ratio;
    std::memcpy(x + p * xs, X.v.data(), sizeof(T) * xs);
    if (stop) stop[p] = out.stop_reason;
    if (iters) iters[p] = out.num_iters;
    if (fails) fails[p] = out.num_failures;
    if (cost) cost[p] = out.final_cost.cost;
    if (nres) nres[p] = out.final_cost.num_residuals;
    if (errs)
      for (size_t k = 0; k < out.errs.size() && int(k) < hs; ++k) {
        errs[size_t(p) * hs + k] = out.errs[k];
        if (deltas2) deltas2[size_t(p) * hs + k] = out.deltas2[k];
        if (succ) succ[size_t(p) * hs + k] = out.successes[k];
      }
  }
}
extern "C" {

void oracle_set_loss(int kind, double th2, float* inlier_ratio_out) {
  g_loss_kind = kind;
  g_loss_th2 = th2;
  g_inlier_out = inlier_ratio_out;
}

int oracle_num_threads_max() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

// Synthetic DenseRow batch (natural layout): A [P][m][n], b [P][m], x0 [P][n] of T; xstar [P][n] double.
void oracle_synth_dense_row(int dtype, uint64_t seed, int64_t problem0, int64_t P, int n, int m,
                            void* A, void* b, void* x0, double* xstar) {
  for (int64_t p = 0; p < P; ++p) {
    if (dtype == TOA_F32)
      synth::dense_row_problem<float>(seed, uint64_t(problem0 + p), n, m,
                                      A ? (float*)A + size_t(p) * m * n : nullptr, b ? (float*)b + size_t(p) * m : nullptr,
                                      x0 ? (float*)x0 + size_t(p) * n : nullptr, xstar ? xstar + size_t(p) * n : nullptr);
    else
      synth::dense_row_problem<double>(seed, uint64_t(problem0 + p), n, m,
                                       A ? (double*)A + size_t(p) * m * n : nullptr, b ? (double*)b + size_t(p) * m : nullptr,
                                       x0 ? (double*)x0 + size_t(p) * n : nullptr, xstar ? xstar + size_t(p) * n : nullptr);
  }
}

void oracle_synth_gaussian_prior(int dtype, uint64_t seed, int64_t problem0, int64_t P, int n,
                                 void* y, void* sigma, void* x0) {
  for (int64_t p = 0; p < P; ++p) {
    if (dtype == TOA_F32)
      synth::gaussian_prior_problem<float>(seed, uint64_t(problem0 + p), n, (float*)y + p * n, (float*)sigma + p * n, (float*)x0 + p * n);
    else
      synth::gaussian_prior_problem<double>(seed, uint64_t(problem0 + p), n, (double*)y + p * n, (double*)sigma + p * n, (double*)x0 + p * n);
  }
}

// One Accumulate call per problem: g [P][n], H [P][n*n] col-major full, cost [P], nres [P].
void oracle_dense_row_accumulate(int dtype, int64_t P, int n, int m, const void* A, const void* b, const void* x,
                                 int want_grad, void* g, void* H, double* cost, int32_t* nres) {
  for (int64_t p = 0; p < P; ++p) {
    if (dtype == TOA_F32) {
      DenseRowAcc<float> acc(n, m, (const float*)A + size_t(p) * m * n, (const float*)b + size_t(p) * m);
      std::vector<float> xv((const float*)x + p * n, (const float*)x + (p + 1) * n);
      float* gp = want_grad ? (float*)g + p * n : nullptr;
      float* Hp = want_grad ? (float*)H + size_t(p) * n * n : nullptr;
      if (gp) { std::fill(gp, gp + n, 0.f); std::fill(Hp, Hp + size_t(n) * n, 0.f); }
      Cost c = acc(xv, gp, Hp);
      cost[p] = c.cost; if (nres) nres[p] = c.num_residuals;
      if (g_inlier_out) g_inlier_out[p] = c.inlier_ratio;
    } else {
      DenseRowAcc<double> acc(n, m, (const double*)A + size_t(p) * m * n, (const double*)b + size_t(p) * m);
      std::vector<double> xv((const double*)x + p * n, (const double*)x + (p + 1) * n);
      double* gp = want_grad ? (double*)g + p * n : nullptr;
      double* Hp = want_grad ? (double*)H + size_t(p) * n * n : nullptr;
      if (gp) { std::fill(gp, gp + n, 0.0); std::fill(Hp, Hp + size_t(n) * n, 0.0); }
      Cost c = acc(xv, gp, Hp);
      cost[p] = c.cost; if (nres) nres[p] = c.num_residuals;
      if (g_inlier_out) g_inlier_out[p] = c.inlier_ratio;
    }
  }
}

// Damped solve per problem, exactly lm.h:108-117 (H_ii *= scale, in double) + gn.h:150-171.
void oracle_solve_damped(int dtype, int64_t P, int n, const void* H, const void* g, double scale, void* dx, int32_t* ok) {
  for (int64_t p = 0; p < P; ++p) {
    if (dtype == TOA_F32) {
      std::vector<float> Hp((const float*)H + size_t(p) * n * n, (const float*)H + size_t(p + 1) * n * n), mg(n);
      for (int i = 0; i < n; ++i) { Hp[size_t(i) * n + i] = float(Hp[size_t(i) * n + i] * scale); mg[i] = -((const float*)g)[p * n + i]; }
      ok[p] = SolveLDLT<float>(n, Hp.data(), mg.data(), (float*)dx + p * n) ? 1 : 0;
    } else {
      std::vector<double> Hp((const double*)H + size_t(p) * n * n, (const double*)H + size_t(p + 1) * n * n), mg(n);
      for (int i = 0; i < n; ++i) { Hp[size_t(i) * n + i] = Hp[size_t(i) * n + i] * scale; mg[i] = -((const double*)g)[p * n + i]; }
      ok[p] = SolveLDLT<double>(n, Hp.data(), mg.data(), (double*)dx + p * n) ? 1 : 0;
    }
  }
}

// Batched LM on DenseRow problems; returns wall seconds.  x [P][n] updated in place.
double oracle_dense_row_lm(int dtype, int64_t P, int n, int m, const void* A, const void* b, void* x,
                           const toa_options* opts, int32_t* stop, int32_t* iters, int32_t* fails, double* cost,
                           double* rerr, double* finalH, double* errs, double* deltas2, uint8_t* succ,
                           int hist_stride, int nthreads) {
  const Options o = from_pod(*opts);
  const auto t0 = std::chrono::steady_clock::now();
  if (dtype == TOA_F32)
    run_batch_dense_row<float>(P, n, m, (const float*)A, (const float*)b, (float*)x, o, stop, iters, fails, cost, rerr,
                               finalH, errs, deltas2, succ, hist_stride, nthreads);
  else
    run_batch_dense_row<double>(P, n, m, (const double*)A, (const double*)b, (double*)x, o, stop, iters, fails, cost,
                                rerr, finalH, errs, deltas2, succ, hist_stride, nthreads);
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// Batched LM on GaussianPrior problems (benchmarks/dense.cpp manual-callback semantics).
double oracle_gaussian_prior_lm_hist(int dtype, int64_t P, int n, const void* y, const void* sigma, void* x,
                                     const toa_options* opts, int32_t* stop, int32_t* iters, int32_t* fails,
                                     double* cost, double* finalH, double* errs, double* deltas2, uint8_t* succ,
                                     int hist_stride);
double oracle_gaussian_prior_lm(int dtype, int64_t P, int n, const void* y, const void* sigma, void* x,
                                const toa_options* opts, int32_t* stop, int32_t* iters, int32_t* fails,
                                double* cost, double* finalH) {
  return oracle_gaussian_prior_lm_hist(dtype, P, n, y, sigma, x, opts, stop, iters, fails, cost, finalH, nullptr, nullptr,
                                       nullptr, 0);
}
// the same with the per-iteration history (Output::errs / deltas2 / successes) for the trajectory comparator
double oracle_gaussian_prior_lm_hist(int dtype, int64_t P, int n, const void* y, const void* sigma, void* x,
                                     const toa_options* opts, int32_t* stop, int32_t* iters, int32_t* fails,
                                     double* cost, double* finalH, double* errs, double* deltas2, uint8_t* succ,
                                     int hist_stride) {
  const Options o = from_pod(*opts);
  const auto t0 = std::chrono::steady_clock::now();
  for (int64_t p = 0; p < P; ++p) {
    Output out;
    if (dtype == TOA_F32) {
      GaussianPriorAcc<float> acc{n, (const float*)y + p * n, (const float*)sigma + p * n};
      std::vector<float> xv((float*)x + p * n, (float*)x + (p + 1) * n);
      Optimizer<float> opt(o, n);
      out = opt.OptimizeAcc(xv, acc, EuclidPlus<float>());
      std::memcpy((float*)x + p * n, xv.data(), sizeof(float) * n);
    } else {
      GaussianPriorAcc<double> acc{n, (const double*)y + p * n, (const double*)sigma + p * n};
      std::vector<double> xv((double*)x + p * n, (double*)x + (p + 1) * n);
      Optimizer<double> opt(o, n);
      out = opt.OptimizeAcc(xv, acc, EuclidPlus<double>());
      std::memcpy((double*)x + p * n, xv.data(), sizeof(double) * n);
    }
    if (stop) stop[p] = out.stop_reason;
    if (iters) iters[p] = out.num_iters;
    if (fails) fails[p] = out.num_failures;
    if (cost) cost[p] = out.final_cost.cost;
    if (finalH && !out.final_hessian.empty())
      std::memcpy(finalH + size_t(p) * n * n, out.final_hessian.data(), sizeof(double) * n * n);
    if (errs)
      for (size_t k = 0; k < out.errs.size() && int(k) < hist_stride; ++k) {
        errs[size_t(p) * hist_stride + k] = out.errs[k];
        if (deltas2) deltas2[size_t(p) * hist_stride + k] = out.deltas2[k];
        if (succ) succ[size_t(p) * hist_stride + k] = out.successes[k];
      }
  }
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// Scalar sqrt2 problems (tests/sqrt2.cpp:30-56 manual float / :58-70 AD double): r = x*x-2,
// grad = J r, H = J^2, cost = r^2 (1 residual).
void oracle_sqrt2_lm(int dtype, int64_t P, void* x, const toa_options* opts, int32_t* stop, int32_t* iters,
                     double* cost, double* errs, double* deltas2, uint8_t* succ, int hist_stride) {
  const Options o = from_pod(*opts);
  for (int64_t p = 0; p < P; ++p) {
    Output out;
    if (dtype == TOA_F32) {
      std::vector<float> xv{((float*)x)[p]};
      auto acc = [](const std::vector<float>& x, float* g, float* H) {
        float r = x[0] * x[0] - 2, J = 2 * x[0];
        if (g) { g[0] = J * r; H[0] = J * J; }
        return Cost(double(r * r));
      };
      Optimizer<float> opt(o, 1);
      out = opt.OptimizeAcc(xv, acc, EuclidPlus<float>());
      ((float*)x)[p] = xv[0];
    } else {
      std::vector<double> xv{((double*)x)[p]};
      auto acc = [](const std::vector<double>& x, double* g, double* H) {
        double r = x[0] * x[0] - 2, J = 2 * x[0];
        if (g) { g[0] = J * r; H[0] = J * J; }
        return Cost(r * r);
      };
      Optimizer<double> opt(o, 1);
      out = opt.OptimizeAcc(xv, acc, EuclidPlus<double>());
      ((double*)x)[p] = xv[0];
    }
    if (stop) stop[p] = out.stop_reason;
    if (iters) iters[p] = out.num_iters;
    if (cost) cost[p] = out.final_cost.cost;
    if (errs)
      for (size_t k = 0; k < out.errs.size() && int(k) < hist_stride; ++k) {
        errs[size_t(p) * hist_stride + k] = out.errs[k];
        if (deltas2) deltas2[size_t(p) * hist_stride + k] = out.deltas2[k];
        if (succ) succ[size_t(p) * hist_stride + k] = out.successes[k];
      }
  }
}


// ---- SE3 reprojection.  poses: [P][12] (R row-major, t), data: [P][8 + 5*npts]
//      ([f cx cy 0 0 0 0 0 | x y z u v ...]), updated in place.
void oracle_se3_reproj_lm(int dtype, int64_t P, int npts, const void* data, void* poses, const toa_options* opts,
                          int32_t* stop, int32_t* iters, double* cost, double* finalH, float* inlier_ratio) {
  const Options o = from_pod(*opts);
  if (dtype == TOA_F32) se3_lm_t<float>(P, npts, (const float*)data, (float*)poses, o, stop, iters, cost, finalH, inlier_ratio);
  else se3_lm_t<double>(P, npts, (const double*)data, (double*)poses, o, stop, iters, cost, finalH, inlier_ratio);
}
// one Accumulate call: g [P][6], H [P][36], cost [P]
void oracle_se3_reproj_accumulate(int dtype, int64_t P, int npts, const void* data, const void* poses, void* g, void* H,
                                  double* cost) {
  for (int64_t p = 0; p < P; ++p) {
    if (dtype == TOA_F32) {
      const float* d = (const float*)data + size_t(p) * (8 + 5 * size_t(npts));
      se3::ReprojAcc<float> acc{npts, d, d + 8};
      se3::Pose<float> x;
      for (int i = 0; i < 12; ++i) x[i] = ((const float*)poses)[p * 12 + i];
      float* gp = (float*)g + p * 6; float* Hp = (float*)H + p * 36;
      std::fill(gp, gp + 6, 0.f); std::fill(Hp, Hp + 36, 0.f);
      cost[p] = acc(x, gp, Hp).cost;
    } else {
      const double* d = (const double*)data + size_t(p) * (8 + 5 * size_t(npts));
      se3::ReprojAcc<double> acc{npts, d, d + 8};
      se3::Pose<double> x;
      for (int i = 0; i < 12; ++i) x[i] = ((const double*)poses)[p * 12 + i];
      double* gp = (double*)g + p * 6; double* Hp = (double*)H + p * 36;
      std::fill(gp, gp + 6, 0.0); std::fill(Hp, Hp + 36, 0.0);
      cost[p] = acc(x, gp, Hp).cost;
    }
  }
}
void oracle_maha_prior_lm(int dtype, int64_t P, int n, const void* data, void* x, const toa_options* opts, int32_t* stop,
                          int32_t* iters, double* cost, double* finalH) {
  const Options o = from_pod(*opts);
  if (dtype == TOA_F32) maha_prior_lm_t<float>(P, n, (const float*)data, (float*)x, o, stop, iters, cost, finalH);
  else maha_prior_lm_t<double>(P, n, (const double*)data, (double*)x, o, stop, iters, cost, finalH);
}

// The reference's analytic optimizer test functions (oracle/testfns.hpp) for a batch of starts.  x: [P][n] in place.
void oracle_testfn_lm(int fn, int dtype, int64_t P, void* x, const toa_options* opts, int32_t* stop, int32_t* iters,
                      int32_t* fails, double* cost, double* errs, double* deltas2, uint8_t* succ, int hist_stride) {
  const Options o = from_pod(*opts);
  const int n = testfn::dims(fn);
  for (int64_t p = 0; p < P; ++p) {
    Output out;
    if (dtype == TOA_F32) {
      std::vector<float> xv((float*)x + p * n, (float*)x + p * n + n);
      Optimizer<float> opt(o, n);
      out = opt.OptimizeAcc(xv, testfn::Acc<float>{fn}, EuclidPlus<float>());
      std::memcpy((float*)x + p * n, xv.data(), sizeof(float) * n);
    } else {
      std::vector<double> xv((double*)x + p * n, (double*)x + p * n + n);
      Optimizer<double> opt(o, n);
      out = opt.OptimizeAcc(xv, testfn::Acc<double>{fn}, EuclidPlus<double>());
      std::memcpy((double*)x + p * n, xv.data(), sizeof(double) * n);
    }
    if (stop) stop[p] = out.stop_reason;
    if (iters) iters[p] = out.num_iters;
    if (fails) fails[p] = out.num_failures;
    if (cost) cost[p] = out.final_cost.cost;
    if (errs)
      for (size_t k = 0; k < out.errs.size() && int(k) < hist_stride; ++k) {
        errs[size_t(p) * hist_stride + k] = out.errs[k];
        if (deltas2) deltas2[size_t(p) * hist_stride + k] = out.deltas2[k];
        if (succ) succ[size_t(p) * hist_stride + k] = out.successes[k];
      }
  }
}
// one Accumulate call of a test function: g [P][n], H [P][n*n] (col-major), cost [P]
void oracle_testfn_accumulate(int fn, int dtype, int64_t P, const void* x, void* g, void* H, double* cost) {
  const int n = testfn::dims(fn);
  for (int64_t p = 0; p < P; ++p) {
    if (dtype == TOA_F32) {
      std::vector<float> xv((const float*)x + p * n, (const float*)x + p * n + n);
      cost[p] = testfn::accumulate<float>(fn, xv, (float*)g + p * n, (float*)H + p * n * n).cost;
    } else {
      std::vector<double> xv((const double*)x + p * n, (const double*)x + p * n + n);
      cost[p] = testfn::accumulate<double>(fn, xv, (double*)g + p * n, (double*)H + p * n * n).cost;
    }
  }
}

// the M-estimators alone (oracle/robust.hpp): loss[i], scale[i] = rho(n2[i], th2)
void oracle_robust_norm(int kind, int dtype, int64_t count, const void* n2, double th2, void* loss, void* scale) {
  for (int64_t i = 0; i < count; ++i) {
    if (dtype == TOA_F32) {
      const auto ls = robust::Apply<float>(kind, ((const float*)n2)[i], float(th2));
      ((float*)loss)[i] = ls.l; ((float*)scale)[i] = ls.s;
    } else {
      const auto ls = robust::Apply<double>(kind, ((const double*)n2)[i], th2);
      ((double*)loss)[i] = ls.l; ((double*)scale)[i] = ls.s;
    }
  }
}
// SE3 pose prior (tests/sophus.cpp:26-44): residual log(prior_inv * x).  prior_inv, poses: [P][12] (R row-major, t).
void oracle_se3_prior_lm(int dtype, int64_t P, const void* prior_inv, void* poses, const toa_options* opts, int32_t* stop,
                         int32_t* iters, double* cost) {
  const Options o = from_pod(*opts);
  for (int64_t p = 0; p < P; ++p) {
    Output out;
    if (dtype == TOA_F32) {
      se3::Pose<float> x;
      for (int i = 0; i < 12; ++i) x[i] = ((float*)poses)[p * 12 + i];
      Optimizer<float> opt(o, 6);
      out = opt.OptimizeAcc(x, se3::PosePriorAcc<float>{(const float*)prior_inv + p * 12}, se3::Plus<float>());
      for (int i = 0; i < 12; ++i) ((float*)poses)[p * 12 + i] = x[i];
    } else {
      se3::Pose<double> x;
      for (int i = 0; i < 12; ++i) x[i] = ((double*)poses)[p * 12 + i];
      Optimizer<double> opt(o, 6);
      out = opt.OptimizeAcc(x, se3::PosePriorAcc<double>{(const double*)prior_inv + p * 12}, se3::Plus<double>());
      for (int i = 0; i < 12; ++i) ((double*)poses)[p * 12 + i] = x[i];
    }
    if (stop) stop[p] = out.stop_reason;
    if (iters) iters[p] = out.num_iters;
    if (cost) cost[p] = out.final_cost.cost;
  }
}
// one Accumulate call of the pose prior: g [P][6], H [P][36] col-major, cost [P]
void oracle_se3_prior_accumulate(int64_t P, const double* prior_inv, const double* poses, double* g, double* H, double* cost) {
  for (int64_t p = 0; p < P; ++p) {
    se3::Pose<double> x;
    for (int i = 0; i < 12; ++i) x[i] = poses[p * 12 + i];
    std::fill(g + p * 6, g + p * 6 + 6, 0.0);
    std::fill(H + p * 36, H + p * 36 + 36, 0.0);
    cost[p] = se3::PosePriorAcc<double>{prior_inv + p * 12}(x, g + p * 6, H + p * 36).cost;
  }
}
// xi = log(pose) for a batch: [P][6] (upsilon, omega)
void oracle_se3_log(int64_t P, const double* poses, double* xi) {
  for (int64_t p = 0; p < P; ++p) se3::se3_log<double, double>(poses + p * 12, poses + p * 12 + 9, xi + p * 6);
}
// pose <- pose * exp(delta) for a batch (tests of the manifold update)
void oracle_se3_plus(int dtype, int64_t P, void* poses, const void* delta) {
  for (int64_t p = 0; p < P; ++p) {
    if (dtype == TOA_F32) {
      se3::Pose<float> x; for (int i = 0; i < 12; ++i) x[i] = ((float*)poses)[p * 12 + i];
      std::vector<float> d((const float*)delta + p * 6, (const float*)delta + p * 6 + 6);
      se3::plus_eq(x, d, 1.f);
      for (int i = 0; i < 12; ++i) ((float*)poses)[p * 12 + i] = x[i];
    } else {
      se3::Pose<double> x; for (int i = 0; i < 12; ++i) x[i] = ((double*)poses)[p * 12 + i];
      std::vector<double> d((const double*)delta + p * 6, (const double*)delta + p * 6 + 6);
      se3::plus_eq(x, d, 1.0);
      for (int i = 0; i < 12; ++i) ((double*)poses)[p * 12 + i] = x[i];
    }
  }
}


// Circle fit (tests/circle.cpp:32-68): x = (cx, cy, radius); r_i = ||p_i - c||^2 - radius^2; the reference
// differentiates it with Jets — here the analytic Jacobian (identical to rounding) folded as the AD bridge does.
// obs: [P][npts][2], x: [P][3] in place.
void oracle_circle_fit_lm(int dtype, int64_t P, int npts, const void* obs, void* x, const toa_options* opts,
                          int32_t* stop, int32_t* iters, double* cost) {
  const Options o = from_pod(*opts);
  for (int64_t p = 0; p < P; ++p) {
    Output out;
    if (dtype == TOA_F32) {
      const float* ob = (const float*)obs + size_t(p) * npts * 2;
      std::vector<float> xv((float*)x + p * 3, (float*)x + p * 3 + 3);
      auto acc = [&](const std::vector<float>& v, float* g, float* H) {
        std::vector<float> r(npts), J(size_t(npts) * 3);
        for (int i = 0; i < npts; ++i) {
          const float dx = ob[2 * i] - v[0], dy = ob[2 * i + 1] - v[1];
          r[i] = dx * dx + dy * dy - v[2] * v[2];
          J[i * 3] = -2 * dx; J[i * 3 + 1] = -2 * dy; J[i * 3 + 2] = -2 * v[2];
        }
        return AccumulateFromJLoss<float>(npts, 3, r.data(), J.data(), g, H);
      };
      Optimizer<float> opt(o, 3);
      out = opt.OptimizeAcc(xv, acc, EuclidPlus<float>());
      std::memcpy((float*)x + p * 3, xv.data(), 12);
    } else {
      const double* ob = (const double*)obs + size_t(p) * npts * 2;
      std::vector<double> xv((double*)x + p * 3, (double*)x + p * 3 + 3);
      auto acc = [&](const std::vector<double>& v, double* g, double* H) {
        std::vector<double> r(npts), J(size_t(npts) * 3);
        for (int i = 0; i < npts; ++i) {
          const double dx = ob[2 * i] - v[0], dy = ob[2 * i + 1] - v[1];
          r[i] = dx * dx + dy * dy - v[2] * v[2];
          J[i * 3] = -2 * dx; J[i * 3 + 1] = -2 * dy; J[i * 3 + 2] = -2 * v[2];
        }
        return AccumulateFromJLoss<double>(npts, 3, r.data(), J.data(), g, H);
      };
      Optimizer<double> opt(o, 3);
      out = opt.OptimizeAcc(xv, acc, EuclidPlus<double>());
      std::memcpy((double*)x + p * 3, xv.data(), 24);
    }
    if (stop) stop[p] = out.stop_reason;
    if (iters) iters[p] = out.num_iters;
    if (cost) cost[p] = out.final_cost.cost;
    if (g_inlier_out) g_inlier_out[p] = out.final_cost.inlier_ratio;
  }
}

void oracle_ba_lm(int dtype, int64_t P, int C, int N, const void* data, void* x, const toa_options* opts, int32_t* stop,
                  int32_t* iters, int32_t* fails, double* cost, int32_t* nres, double* errs, double* deltas2, uint8_t* succ, int hs) {
  const Options o = from_pod(*opts);
  if (dtype == TOA_F32) ba_lm_t<float>(P, C, N, (const float*)data, (float*)x, o, stop, iters, fails, cost, nres, errs, deltas2, succ, hs);
  else ba_lm_t<double>(P, C, N, (const double*)data, (double*)x, o, stop, iters, fails, cost, nres, errs, deltas2, succ, hs);
}

}  // extern "C"
