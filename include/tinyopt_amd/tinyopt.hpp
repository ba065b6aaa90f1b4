// tinyopt_amd/tinyopt.hpp — header-only C++ host adaptor over the C-ABI (include/tinyopt_amd.h).
//
// Mirrors the reference's user surface for the LM hot path so that a tinyopt call site changes by a
// namespace and a cost-functor type only:
//
//   reference (include/tinyopt/optimize.h:16-17)          here
//   ---------------------------------------------          ----------------------------------------
//   tinyopt::Options options;                               tinyopt_amd::Options options;
//   auto out = tinyopt::Optimize(x, cost, options);         auto out = tinyopt_amd::Optimize(x, cost, options);
//     x    : one parameter block, updated in place            x    : P parameter blocks ([P][n] contiguous), in place
//     cost : any C++ callable (residuals or Accumulate)        cost : a DEVICE model (DenseRow, GaussianPrior, Sqrt2, CircleFit, SE3Reproj) — host
//                                                                       lambdas cannot run on the GPU
//     out  : tinyopt::Output                                   out  : BatchOutput (one Output row per problem)
//
// Option names / nesting / defaults follow include/tinyopt/optimizers/options.h:18-156; StopReason
// values follow include/tinyopt/stop_reasons.h:14-43; Output fields follow include/tinyopt/output.h.
// No Eigen, no HIP headers: link with -ltinyopt_amd only.  Errors of use throw std::invalid_argument
// (as the reference does, optimize.h:47,55,75); numeric failures are StopReason values, never exceptions.
#pragma once

#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "../tinyopt_amd.h"

namespace tinyopt_amd {

enum StopReason : int {  // include/tinyopt/stop_reasons.h:14-43
  kOutOfMemory = -4, kSolverFailed = -3, kSystemHasNaNOrInf = -2, kSkipped = -1, kNone = 0, kMinError, kMinRelError,
  kMinDeltaNorm, kMinGradNorm, kMaxIters, kMaxNoDecr, kMaxConsecNoDecr, kTimedOut, kUserStopped
};

struct Options {  // include/tinyopt/optimizers/options.h:18-156 (numeric knobs)
  enum Solver { LevenbergMarquardt = 0, GaussNewton = 1 };
  Solver solver_type = LevenbergMarquardt;
  bool check_final_cost = false;
  bool use_step_quality_approx = false;
  float grad_clipping = 0;
  struct Hessian {
    bool use_ldlt = true;
    bool H_is_full = true;
    float check_min_H_diag = 0;
    bool save_last = true;
  } hessian;
  struct CostScaling {
    bool use_squared_norm = true;
    bool downscale_by_2 = false;
    bool normalize = false;
  } cost;
  uint16_t max_iters = 50;
  float min_error = 1e-12f;
  float min_rerr_dec = 1e-10f;
  float min_step_norm2 = 1e-14f;
  float min_grad_norm2 = 1e-18f;
  uint8_t max_total_failures = 0;
  uint8_t max_consec_failures = 5;
  struct LM {
    float damping_init = 1e-4f;
    std::array<float, 2> damping_range{{1e-9f, 1e9f}};
    float good_factor = 1.0f / 3.0f;
    float bad_factor = 2.0f;
  } lm;

  toa_options to_pod() const {
    toa_options p;
    toa_options_default(&p);
    p.solver_type = solver_type;
    p.max_iters = max_iters;
    p.min_error = min_error;
    p.min_rerr_dec = min_rerr_dec;
    p.min_step_norm2 = min_step_norm2;
    p.min_grad_norm2 = min_grad_norm2;
    p.max_total_failures = max_total_failures;
    p.max_consec_failures = max_consec_failures;
    p.damping_init = lm.damping_init;
    p.damping_min = lm.damping_range[0];
    p.damping_max = lm.damping_range[1];
    p.good_factor = lm.good_factor;
    p.bad_factor = lm.bad_factor;
    p.grad_clipping = grad_clipping;
    p.check_min_H_diag = hessian.check_min_H_diag;
    p.check_final_cost = check_final_cost;
    p.use_step_quality_approx = use_step_quality_approx;
    p.use_ldlt = hessian.use_ldlt;
    p.H_is_full = hessian.H_is_full;
    p.save_last = hessian.save_last;
    p.use_squared_norm = cost.use_squared_norm;
    p.downscale_by_2 = cost.downscale_by_2;
    p.normalize = cost.normalize;
    return p;
  }
};

// benchmarks/options.h:10-27
inline Options CreateBenchmarkOptions() {
  Options o;
  o.max_iters = 10;
  o.min_error = 0;
  o.min_rerr_dec = 1e-12f;
  o.min_step_norm2 = 1e-16f;
  o.max_consec_failures = 3;
  o.hessian.save_last = false;
  return o;
}

inline void check(int rc) {
  if (rc == TOA_OK) return;
  const std::string msg = std::string("tinyopt_amd: ") + toa_last_error();
  if (rc == TOA_E_ARG || rc == TOA_E_UNSUPPORTED) throw std::invalid_argument(msg);
  if (rc == TOA_E_NOMEM) throw std::bad_alloc();
  throw std::runtime_error(msg);
}

// RAII over toa_handle: one per host thread per GPU (the reference's Optimizer_ is equally stateful).
class Context {
 public:
  explicit Context(int device = 0, void* stream = nullptr) { check(toa_create(&h_, device, stream)); }
  ~Context() { toa_destroy(h_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  toa_handle get() const { return h_; }

 private:
  toa_handle h_ = nullptr;
};

template <typename T>
class DeviceBuffer {
 public:
  DeviceBuffer() = default;
  DeviceBuffer(const Context& c, size_t count) : c_(&c), n_(count) {
    void* p = nullptr;
    check(toa_malloc(c.get(), &p, count * sizeof(T)));
    p_ = static_cast<T*>(p);
  }
  ~DeviceBuffer() { if (p_) toa_free(c_->get(), p_); }
  DeviceBuffer(DeviceBuffer&& o) noexcept : c_(o.c_), p_(o.p_), n_(o.n_) { o.p_ = nullptr; }
  DeviceBuffer& operator=(DeviceBuffer&& o) noexcept {
    if (this != &o) { if (p_) toa_free(c_->get(), p_); c_ = o.c_; p_ = o.p_; n_ = o.n_; o.p_ = nullptr; }
    return *this;
  }
  T* data() const { return p_; }
  size_t size() const { return n_; }
  void upload(const T* host) { check(toa_memcpy_h2d(c_->get(), p_, host, n_ * sizeof(T))); }
  void download(T* host) const { check(toa_memcpy_d2h(c_->get(), host, p_, n_ * sizeof(T))); }
  void zero() { check(toa_memset(c_->get(), p_, 0, n_ * sizeof(T))); }

 private:
  const Context* c_ = nullptr;
  T* p_ = nullptr;
  size_t n_ = 0;
};

template <typename Scalar>
constexpr int dtype_of() {
  static_assert(std::is_same<Scalar, float>::value || std::is_same<Scalar, double>::value,
                "Scalar must be float or double (the reference's solver Scalar)");
  return std::is_same<Scalar, float>::value ? TOA_F32 : TOA_F64;
}

// Device cost model: r_i(x) = a_i.x + 0.1 sin(a_i.x) - b_i for P problems.  Takes the place of the
// residual functor in Optimize(x, cost) (e.g. benchmarks/dense.cpp:56,71-74).
template <typename Scalar>
class DenseRow {
 public:
  // A: [P][m][n] row-major, b: [P][m] — host arrays; uploaded and packed into the HBM layout once.
  DenseRow(const Context& ctx, int64_t P, int n, int m, const Scalar* A, const Scalar* b) : ctx_(&ctx), P_(P), n_(n), m_(m) {
    size_t bytes = 0;
    check(toa_dense_row_layout(dtype_of<Scalar>(), n, m, nullptr, nullptr, nullptr, nullptr, &bytes));
    packed_ = DeviceBuffer<Scalar>(ctx, size_t(P) * bytes / sizeof(Scalar));
    DeviceBuffer<Scalar> dA(ctx, size_t(P) * m * n), db(ctx, size_t(P) * m);
    dA.upload(A);
    db.upload(b);
    check(toa_dense_row_pack(ctx.get(), dtype_of<Scalar>(), n, m, P, dA.data(), db.data(), packed_.data()));
    check(toa_synchronize(ctx.get()));
  }
  static constexpr int model_id = TOA_MODEL_DENSE_ROW;
  int64_t P() const { return P_; }
  int n() const { return n_; }
  int m() const { return m_; }
  int xdim() const { return n_; }
  const Scalar* data() const { return packed_.data(); }
  const Context& ctx() const { return *ctx_; }

 private:
  const Context* ctx_;
  int64_t P_;
  int n_, m_;
  DeviceBuffer<Scalar> packed_;
};

// Generic holder for the models whose device data is a plain host array uploaded as is.
template <typename Scalar, int ModelId>
class PlainModel {
 public:
  PlainModel(const Context& ctx, int64_t P, int n, int m, int xdim, const Scalar* host, size_t count)
      : ctx_(&ctx), P_(P), n_(n), m_(m), xdim_(xdim), data_(ctx, count ? count : 1) {
    if (count) data_.upload(host);
  }
  static constexpr int model_id = ModelId;
  int64_t P() const { return P_; }
  int n() const { return n_; }
  int m() const { return m_; }
  int xdim() const { return xdim_; }
  const Scalar* data() const { return data_.data(); }
  const Context& ctx() const { return *ctx_; }

 private:
  const Context* ctx_;
  int64_t P_;
  int n_, m_, xdim_;
  DeviceBuffer<Scalar> data_;
};

// r = (x - y) / sigma, m = n: the reference's published dense benchmark (benchmarks/dense.cpp:53-66).
// ys: [P][2][n] = y then sigma per problem.
template <typename Scalar>
struct GaussianPrior : PlainModel<Scalar, TOA_MODEL_GAUSSIAN_PRIOR> {
  GaussianPrior(const Context& ctx, int64_t P, int n, const Scalar* ys)
      : PlainModel<Scalar, TOA_MODEL_GAUSSIAN_PRIOR>(ctx, P, n, n, n, ys, size_t(P) * 2 * n) {}
};
// r = x*x - 2 (tests/sqrt2.cpp:30-70)
template <typename Scalar>
struct Sqrt2 : PlainModel<Scalar, TOA_MODEL_SQRT2> {
  Sqrt2(const Context& ctx, int64_t P) : PlainModel<Scalar, TOA_MODEL_SQRT2>(ctx, P, 1, 1, 1, nullptr, 0) {}
};
// tests/circle.cpp:32-68, differentiated on the device by dual numbers.  obs: [P][npts][2]; x = (cx, cy, radius).
template <typename Scalar>
struct CircleFit : PlainModel<Scalar, TOA_MODEL_CIRCLE_FIT> {
  CircleFit(const Context& ctx, int64_t P, int npts, const Scalar* obs)
      : PlainModel<Scalar, TOA_MODEL_CIRCLE_FIT>(ctx, P, 3, npts, 3, obs, size_t(P) * npts * 2) {}
};
// The DenseRow residual r_i = a_i.x + 0.1 sin(a_i.x) - b_i for parameter blocks beyond one wavefront (1 <= n <= 1024;
// `Dims == Dynamic` in the reference, optimize.h:27-33): batched library GEMM + rocSOLVER Cholesky under the same LM state
// machine.  data: per problem A row-major [m][n] followed by b [m].
template <typename Scalar>
struct DenseRowNatural : PlainModel<Scalar, TOA_MODEL_DENSE_ROW_NATURAL> {
  DenseRowNatural(const Context& ctx, int64_t P, int n, int m, const Scalar* A_then_b)
      : PlainModel<Scalar, TOA_MODEL_DENSE_ROW_NATURAL>(ctx, P, n, m, n, A_then_b, size_t(P) * m * (size_t(n) + 1)) {}
};
// SE3 pinhole reprojection (3rdparty/traits/sophus.h:13-27 update); x: [P][12] = R (row-major) | t; n = 6.
// data: [P][8 + 5*npts] = [f cx cy 0 0 0 0 0 | x y z u v ...].
// Optional M-estimator on each point's squared reprojection error: data[3] = TOA_LOSS_*, data[4] = th^2
// (losses/robust_norms.h:32-316); set them in the host array before constructing the model.
template <typename Scalar>
struct SE3Reproj : PlainModel<Scalar, TOA_MODEL_SE3_REPROJ> {
  SE3Reproj(const Context& ctx, int64_t P, int npts, const Scalar* data)
      : PlainModel<Scalar, TOA_MODEL_SE3_REPROJ>(ctx, P, 6, 2 * npts, 12, data, size_t(P) * (8 + 5 * size_t(npts))) {}
};

// include/tinyopt/output.h:26-145, one entry per problem.
struct BatchOutput {
  std::vector<int32_t> stop_reason, num_iters, num_failures, num_consec_failures, final_num_residuals;
  std::vector<double> final_cost, final_rerr_dec;
  std::vector<float> final_inlier_ratio;      // Output::final_cost.inlier_ratio (cost.h:84-95); 1 without a robust loss
  std::vector<double> final_hessian;          // [P][n*n], undamped; empty unless options.hessian.save_last
  std::vector<double> errs, deltas2;          // [P][hist_stride] (only with history = true)
  std::vector<uint8_t> successes;
  int hist_stride = 0;
  bool Succeeded(size_t p) const { return stop_reason[p] >= kNone; }                                  // output.h:30
  bool Converged(size_t p) const { return stop_reason[p] >= kMinError && stop_reason[p] < kMaxIters; } // output.h:33-35
};

// tinyopt::Optimize(x, cost, options) for a batch.  x: [P][n] contiguous host scalars, updated in place.
template <typename Scalar, typename Cost>
BatchOutput Optimize(std::vector<Scalar>& x, const Cost& cost, const Options& options = {}, bool history = false) {
  const int64_t P = cost.P();
  const int n = cost.n();
  if (int64_t(x.size()) != P * cost.xdim())
    throw std::invalid_argument("tinyopt_amd::Optimize: x must hold P * (parameters per problem) scalars");
  const Context& ctx = cost.ctx();
  DeviceBuffer<Scalar> dx(ctx, x.size());
  dx.upload(x.data());
  DeviceBuffer<int32_t> stop(ctx, P), iters(ctx, P), fails(ctx, P), cfails(ctx, P), nres(ctx, P);
  DeviceBuffer<double> fc(ctx, P), fr(ctx, P);
  DeviceBuffer<float> inl(ctx, P);
  DeviceBuffer<double> fH, errs, d2;
  DeviceBuffer<uint8_t> succ;
  toa_results r{};
  r.stop_reason = stop.data(); r.num_iters = iters.data(); r.num_failures = fails.data();
  r.num_consec_failures = cfails.data(); r.final_cost = fc.data(); r.final_num_residuals = nres.data();
  r.final_rerr_dec = fr.data();
  r.final_inlier_ratio = inl.data();
  BatchOutput out;
  if (options.hessian.save_last) { fH = DeviceBuffer<double>(ctx, size_t(P) * n * n); fH.zero(); r.final_hessian = fH.data(); }
  if (history) {
    out.hist_stride = options.max_iters + 2;
    errs = DeviceBuffer<double>(ctx, size_t(P) * out.hist_stride); errs.zero();
    d2 = DeviceBuffer<double>(ctx, size_t(P) * out.hist_stride); d2.zero();
    succ = DeviceBuffer<uint8_t>(ctx, size_t(P) * out.hist_stride); succ.zero();
    r.errs = errs.data(); r.deltas2 = d2.data(); r.successes = succ.data(); r.hist_stride = out.hist_stride;
  }
  const toa_options pod = options.to_pod();
  check(toa_lm_run(ctx.get(), Cost::model_id, dtype_of<Scalar>(), n, cost.m(), P, cost.data(), dx.data(), &pod, &r, nullptr));
  check(toa_synchronize(ctx.get()));
  dx.download(x.data());
  auto get = [&](auto& vec, const auto& buf) { vec.resize(buf.size()); buf.download(vec.data()); };
  get(out.stop_reason, stop); get(out.num_iters, iters); get(out.num_failures, fails);
  get(out.num_consec_failures, cfails); get(out.final_num_residuals, nres); get(out.final_cost, fc);
  get(out.final_rerr_dec, fr); get(out.final_inlier_ratio, inl);
  if (options.hessian.save_last) get(out.final_hessian, fH);
  if (history) { get(out.errs, errs); get(out.deltas2, d2); get(out.successes, succ); }
  return out;
}

// The reference's class / stepping form (`lm::Optimizer<H_t> optimizer(options)`; `optimizer.Step(x, acc, out)` one loop
// pass at a time; `optimizer(x, f, max_iters)` — include/tinyopt/optimizers/optimizer.h:199,331-539) for a batch.
// x stays on the device between steps; `x()` downloads the current iterate, `output()` the results so far.
template <typename Scalar, typename Cost>
class Optimizer {
 public:
  Optimizer(std::vector<Scalar>& x, const Cost& cost, const Options& options = {})
      : x_(&x), cost_(&cost), options_(options), pod_(options.to_pod()), P_(cost.P()), n_(cost.n()),
        dx_(cost.ctx(), x.size()), stop_(cost.ctx(), P_), iters_(cost.ctx(), P_), fails_(cost.ctx(), P_), cfails_(cost.ctx(), P_),
        nres_(cost.ctx(), P_), fc_(cost.ctx(), P_), fr_(cost.ctx(), P_), inl_(cost.ctx(), P_), active_(cost.ctx(), 1),
        state_(cost.ctx(), toa_lm_state_bytes(dtype_of<Scalar>(), n_, P_)) {
    if (int64_t(x.size()) != P_ * cost.xdim())
      throw std::invalid_argument("tinyopt_amd::Optimizer: x must hold P * (parameters per problem) scalars");
    dx_.upload(x.data());
    stop_.zero(); iters_.zero(); fc_.zero();
    r_ = toa_results{};
    r_.stop_reason = stop_.data(); r_.num_iters = iters_.data(); r_.num_failures = fails_.data();
    r_.num_consec_failures = cfails_.data(); r_.final_cost = fc_.data(); r_.final_num_residuals = nres_.data();
    r_.final_rerr_dec = fr_.data(); r_.final_inlier_ratio = inl_.data();
    if (options.hessian.save_last) { fH_ = DeviceBuffer<double>(cost.ctx(), size_t(P_) * n_ * n_); fH_.zero(); r_.final_hessian = fH_.data(); }
    check(toa_lm_begin(cost.ctx().get(), Cost::model_id, dtype_of<Scalar>(), n_, cost.m(), P_, cost.data(), dx_.data(), &pod_, &r_,
                       state_.data()));
  }
  // One pass of the loop body for every running problem; returns how many are still running.
  int64_t Step() {
    const Context& ctx = cost_->ctx();
    active_.zero();
    check(toa_lm_step(ctx.get(), Cost::model_id, dtype_of<Scalar>(), n_, cost_->m(), P_, cost_->data(), dx_.data(), &pod_, &r_,
                      nullptr, state_.data(), active_.data()));
    check(toa_synchronize(ctx.get()));
    int32_t a = 0;
    active_.download(&a);
    dx_.download(x_->data());  // x by reference, updated in place at every Step (optimizer.h:271-279)
    return a;
  }
  // `optimizer(x, f, max_iters)`
  BatchOutput operator()(int max_iters = -1) {
    const int limit = max_iters < 0 ? int(options_.max_iters) + 2 : max_iters;
    for (int i = 0; i < limit; ++i)
      if (Step() == 0) break;
    return output();
  }
  BatchOutput output() const {
    BatchOutput out;
    auto get = [&](auto& vec, const auto& buf) { vec.resize(buf.size()); buf.download(vec.data()); };
    get(out.stop_reason, stop_); get(out.num_iters, iters_); get(out.num_failures, fails_);
    get(out.num_consec_failures, cfails_); get(out.final_num_residuals, nres_); get(out.final_cost, fc_);
    get(out.final_rerr_dec, fr_); get(out.final_inlier_ratio, inl_);
    if (options_.hessian.save_last) get(out.final_hessian, fH_);
    return out;
  }

 private:
  std::vector<Scalar>* x_;
  const Cost* cost_;
  Options options_;
  toa_options pod_;
  int64_t P_;
  int n_;
  DeviceBuffer<Scalar> dx_;
  DeviceBuffer<int32_t> stop_, iters_, fails_, cfails_, nres_;
  DeviceBuffer<double> fc_, fr_;
  DeviceBuffer<float> inl_;
  DeviceBuffer<int32_t> active_;
  DeviceBuffer<unsigned char> state_;
  DeviceBuffer<double> fH_;
  toa_results r_;
};

// The Accumulate-callback seam `acc(x, grad, H) -> Cost` (docs/API.md:37-57) for a batch; grad == nullptr
// (cost only) when g / H are null.  g: [P][n], H: [P][n*n], cost: [P] (= ||r||^2), all host.
template <typename Scalar, typename Cost>
void Accumulate(const Cost& cost, const std::vector<Scalar>& x, std::vector<Scalar>* g, std::vector<Scalar>* H,
                std::vector<double>& cost_out) {
  const int64_t P = cost.P();
  const int n = cost.n();
  const Context& ctx = cost.ctx();
  DeviceBuffer<Scalar> dx(ctx, x.size());
  dx.upload(x.data());
  DeviceBuffer<double> dc(ctx, P);
  const bool want = g && H;
  DeviceBuffer<Scalar> dg, dH;
  if (want) { dg = DeviceBuffer<Scalar>(ctx, size_t(P) * n); dH = DeviceBuffer<Scalar>(ctx, size_t(P) * n * n); }
  check(toa_accumulate(ctx.get(), Cost::model_id, dtype_of<Scalar>(), n, cost.m(), P, cost.data(), dx.data(), want ? 1 : 0,
                       want ? dg.data() : nullptr, want ? dH.data() : nullptr, dc.data(), nullptr));
  check(toa_synchronize(ctx.get()));
  cost_out.resize(P);
  dc.download(cost_out.data());
  if (want) { g->resize(size_t(P) * n); dg.download(g->data()); H->resize(size_t(P) * n * n); dH.download(H->data()); }
}

// losses::Huber(n2, th2, true) & co. (losses/robust_norms.h:32-316) for an array of squared norms: kind = TOA_LOSS_*.
template <typename Scalar>
void RobustNorm(const Context& ctx, int kind, const std::vector<Scalar>& n2, Scalar th2, std::vector<Scalar>& loss,
                std::vector<Scalar>& scale) {
  DeviceBuffer<Scalar> dn(ctx, n2.size()), dl(ctx, n2.size()), ds(ctx, n2.size());
  dn.upload(n2.data());
  check(toa_robust_norm(ctx.get(), kind, dtype_of<Scalar>(), int64_t(n2.size()), dn.data(), double(th2), dl.data(), ds.data()));
  check(toa_synchronize(ctx.get()));
  loss.resize(n2.size());
  scale.resize(n2.size());
  dl.download(loss.data());
  ds.download(scale.data());
}

// tinyopt::InvCov (include/tinyopt/math.h:41-91) for a batch of n x n matrices; ok[p] == 0 <=> std::nullopt.
template <typename Scalar>
void InvCov(const Context& ctx, int64_t P, int n, const std::vector<Scalar>& H, std::vector<Scalar>& C, std::vector<int32_t>& ok) {
  DeviceBuffer<Scalar> dH(ctx, H.size()), dC(ctx, H.size());
  DeviceBuffer<int32_t> dok(ctx, P);
  dH.upload(H.data());
  check(toa_inv_cov(ctx.get(), dtype_of<Scalar>(), n, P, dH.data(), dC.data(), dok.data()));
  check(toa_synchronize(ctx.get()));
  C.resize(H.size());
  dC.download(C.data());
  ok.resize(P);
  dok.download(ok.data());
}

}  // namespace tinyopt_amd
