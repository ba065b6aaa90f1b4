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
#include <chrono>
#include <cstdint>
#include <functional>
#include <iostream>
#include <stdexcept>
#include <string>
#include <algorithm>
#include <type_traits>
#include <utility>
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
  // Host-side stop controls (options.h:96-106).  When any is set, Optimize() drives the loop through the stepping form
  // and evaluates them between iterations (optimizer.h:302-305, 529-534); per problem of the batch.
  double max_duration_ms = 0;                                          // 0 = no limit -> kTimedOut
  std::function<bool(double, double, double)> stop_callback;           // (err, |dx|^2, |g|^2) -> kUserStopped
  std::function<bool(float, const std::vector<float>&, const std::vector<float>&)> stop_callback2;   // (err, dx, g)
  // The per-iteration log line of Optimizer_::Step (options.h:113-125, optimizer.h:463-516), through the same stepping form.  Off by
  // default here (the reference logs by default; a batched solve would print one line per problem and iteration): the problems
  // named in `problems` (empty = problem 0) are logged to `sink` (empty = std::cout, like TINYOPT_LOG: log.h:23).
  // print_max_stdev / print_J_jet / print_failure are not mirrored.
  struct Log {
    bool enable = false;
    std::string e = "\xCE\xB5\xC2\xB2";   // "ε²"
    bool print_emoji = true;
    bool print_x = false;
    bool print_dx = false;
    bool print_inliers = false;
    bool print_t = true;
    std::vector<int64_t> problems;
    std::function<void(const std::string&)> sink;
  } log;
  bool has_host_controls() const { return max_duration_ms > 0 || stop_callback || stop_callback2 || log.enable; }
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
  explicit Context(int device = 0, void* stream = nullptr) {
    // the header and the library must agree on the layout of what crosses the boundary (counters_dev is [TOA_NUM_COUNTERS = 8]
    // since ABI 4; a header compiled against an older library — or the reverse — must not run)
    if (toa_abi_version() != TOA_ABI_VERSION)
      throw std::runtime_error("tinyopt_amd: libtinyopt_amd.so has ABI version " + std::to_string(toa_abi_version()) + ", this header expects " +
                               std::to_string(TOA_ABI_VERSION));
    check(toa_create(&h_, device, stream));
  }
  ~Context() { toa_destroy(h_); }
  // A/B arms of the library as typed per-handle state (toa_tuning; nullptr = the library's own choices)
  void set_tuning(const toa_tuning* t) const { check(toa_set_tuning(h_, t)); }
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

// The M-estimator part of a cost functor (losses/robust_norms.h:32-316; docs/API.md:396-411 wraps a residual's squared norm
// in `losses::Huber(n2, th2, true)`): `cost.set_loss(TOA_LOSS_HUBER, th)` makes every launch with this model pass each
// residual item through rho — cost += l, the item's J^T J / J^T r scaled by s, inliers in final_inlier_ratio.  Honoured by
// DenseRow, CircleFit (the families toa_set_loss lists); SE3Reproj carries its loss in its data header.
struct LossTag {
  int loss_kind = TOA_LOSS_L2;
  double loss_th2 = 0;
  void set_loss(int kind, double th) { loss_kind = kind; loss_th2 = th * th; }
  bool se3_header_l2 = false;   // SE3Reproj: no problem's data header names a loss -> the kernels without the M-estimator branch (toa_tuning)
};
template <typename Cost>
inline void apply_loss(const Cost& cost) {
  check(toa_set_loss(cost.ctx().get(), cost.loss_kind, cost.loss_th2));
  toa_tuning t;
  check(toa_get_tuning(cost.ctx().get(), &t));
  if ((t.se3_reproj_header_l2 != 0) != cost.se3_header_l2) {   // (the field follows the model: a second host call only when it changes)
    t.se3_reproj_header_l2 = cost.se3_header_l2 ? 1 : 0;
    check(toa_set_tuning(cost.ctx().get(), &t));
  }
}

// Device cost model: r_i(x) = a_i.x + 0.1 sin(a_i.x) - b_i for P problems.  Takes the place of the
// residual functor in Optimize(x, cost) (e.g. benchmarks/dense.cpp:56,71-74).
template <typename Scalar>
class DenseRow : public LossTag {
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
class PlainModel : public LossTag {
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
// The analytic functions of the reference's optimizer tests as manual Accumulate callbacks (tests/optimize_easy.cpp,
// optimize_hard.cpp, basic.cpp:41-87): fn = 0 Rosenbrock, 1 plateau, 2 Powell (n = 4), 3 Beale, 4 Himmelblau, 5 `x - 2` (n = 1).
template <typename Scalar>
struct TestFn : PlainModel<Scalar, TOA_MODEL_TESTFN> {
  static int dims(int fn) { return fn == 5 ? 1 : (fn == 2 ? 4 : 2); }
  static int residuals(int fn) { return fn == 3 ? 3 : (fn == 4 ? 2 : 1); }
  TestFn(const Context& ctx, int64_t P, int fn) : PlainModel<Scalar, TOA_MODEL_TESTFN>(ctx, P, dims(fn), residuals(fn), dims(fn), id(fn), 1) {}

 private:
  const Scalar* id(int fn) { fn_ = Scalar(fn); return &fn_; }
  Scalar fn_ = 0;
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
      : PlainModel<Scalar, TOA_MODEL_SE3_REPROJ>(ctx, P, 6, 2 * npts, 12, data, size_t(P) * (8 + 5 * size_t(npts))) {
    bool any = false;   // (a header that names a loss keeps the kernels with the M-estimator branch)
    for (int64_t p = 0; p < P && !any; ++p) any = data[size_t(p) * (8 + 5 * size_t(npts)) + 3] != Scalar(0);
    this->se3_header_l2 = !any;
  }
};

// A residual supplied as C++ source text at RUN time (toa_model_compile: hiprtc + hipModuleLoad, no rebuild of the library) —
// the device-side form of `tinyopt::Optimize(x, [](const auto& x) { return r(x); })` (optimize.h:16-33, optimizer.h:145-160).
// body: generic in its scalar type S; x[j] parameter j, p[k] the item's scalars, h[k] the header scalars, r[q] the residuals:
//   JitResidual<double> fit(ctx, "const S dx = p[0] - x[0]; const S dy = p[1] - x[1]; r[0] = dx*dx + dy*dy - x[2]*x[2];", 3, 2);
//   auto out = Optimize(x, fit.bind(P, npts, obs));                                            // tests/circle.cpp:32-68
template <typename Scalar>
class JitModel;
template <typename Scalar>
class JitResidual {
 public:
  // manifold: TOA_MANIFOLD_EUCLID, or TOA_MANIFOLD_SE3 — x is ONE pose stored as 12 scalars (R row-major, t), n = 6, the body reads
  //   it through x[0..11] and may call se3_log<S, T>(R, t, xi)  (tests/sophus.cpp:26-44 `Optimize(pose, lambda)`);
  // kind: TOA_JIT_RESIDUAL (differentiated on the device), or TOA_JIT_ACCUMULATE — a manual Accumulate callback: the body fills
  //   r[q] and, inside `if (want_grad)`, its own Jacobian rows J[q][a]  (docs/API.md:37-57);
  // n up to 63, up to 8 residuals per item (beyond 12 parameters a row is a lane on the matrix-core Gram: both kinds, M-estimators, user manifolds —
  //   round 6).  Compiled code objects are cached on disk (toa_jit_set_cache_dir).
  // manifold = TOA_MANIFOLD_USER (round 5): the caller's own parameter container — tinyopt's traits::params_trait<T> (traits.h:103-359) as text:
  //   x_scalars = the container as stored, n = the dimension of its tangent, plus_body = the body of
  //   `template <class S> void plus(const T* x, const S* d, S* xp)`, xp = x (+) d (PlusEq on plain T; differentiated through Jets seeded on d).
  JitResidual(const Context& ctx, const std::string& body, int n, int item_scalars, int residuals_per_item = 1, int header_scalars = 0,
              int manifold = TOA_MANIFOLD_EUCLID, int kind = TOA_JIT_RESIDUAL, const std::string& plus_body = std::string(), int x_scalars = 0)
      : ctx_(&ctx), n_(n), kR_(residuals_per_item), kD_(item_scalars), kH_(header_scalars),
        xdim_(manifold == TOA_MANIFOLD_SE3 ? 12 : (manifold == TOA_MANIFOLD_USER ? x_scalars : n)) {
    std::vector<char> log(1 << 16);
    toa_jit_spec spec{};
    spec.dtype = dtype_of<Scalar>(); spec.num_params = n; spec.residuals_per_item = residuals_per_item;
    spec.scalars_per_item = item_scalars; spec.header_scalars = header_scalars; spec.manifold = manifold; spec.kind = kind;
    if (manifold == TOA_MANIFOLD_USER) { spec.x_scalars = x_scalars; spec.plus_body = plus_body.c_str(); }
    const int rc = toa_model_compile_ex(ctx.get(), &spec, body.c_str(), &h_, log.data(), log.size());
    log_ = log.data();
    check(rc);
  }
  ~JitResidual() { if (h_) (void)toa_model_destroy(h_); }
  JitResidual(const JitResidual&) = delete;
  JitResidual& operator=(const JitResidual&) = delete;
  // data: [P][header_scalars + items * item_scalars] host scalars
  JitModel<Scalar> bind(int64_t P, int items, const Scalar* data) const { return JitModel<Scalar>(*this, P, items, data); }
  const std::string& compile_log() const { return log_; }
  toa_jit_model handle() const { return h_; }
  const Context& ctx() const { return *ctx_; }
  int n() const { return n_; }
  int residuals_per_item() const { return kR_; }
  int item_scalars() const { return kD_; }
  int header_scalars() const { return kH_; }
  int xdim() const { return xdim_; }   // stored scalars of x per problem (12 for an SE3 pose)
  // What the run-time build came out as (toa_jit_model_stats): resident workgroups per compute unit, LDS per workgroup, vector
  // registers per lane, scratch bytes per lane — a body heavy enough to spill or to drop to one workgroup shows up here.
  struct BuildStats { int wg_per_cu = 0, lds_bytes_per_wg = 0, num_regs = 0, scratch_bytes = 0; };
  BuildStats stats() const {
    BuildStats s;
    check(toa_jit_model_stats(h_, &s.wg_per_cu, &s.lds_bytes_per_wg, &s.num_regs, &s.scratch_bytes));
    return s;
  }

 private:
  const Context* ctx_;
  toa_jit_model h_ = nullptr;
  int n_, kR_, kD_, kH_, xdim_;
  std::string log_;
};
template <typename Scalar>
class JitModel : public LossTag {
 public:
  JitModel(const JitResidual<Scalar>& res, int64_t P, int items, const Scalar* host)
      : res_(&res), P_(P), items_(items), data_(res.ctx(), size_t(P) * (res.header_scalars() + size_t(items) * res.item_scalars())) {
    data_.upload(host);
  }
  static constexpr int model_id = -1;
  int64_t P() const { return P_; }
  int n() const { return res_->n(); }
  int m() const { return items_ * res_->residuals_per_item(); }
  int xdim() const { return res_->xdim(); }
  int items() const { return items_; }
  const Scalar* data() const { return data_.data(); }
  const Context& ctx() const { return res_->ctx(); }
  toa_jit_model jit_handle() const { return res_->handle(); }

 private:
  const JitResidual<Scalar>* res_;
  int64_t P_;
  int items_;
  DeviceBuffer<Scalar> data_;
};
namespace detail {
template <typename C, typename = void> struct is_jit : std::false_type {};
template <typename C> struct is_jit<C, std::void_t<decltype(std::declval<const C&>().jit_handle())>> : std::true_type {};
}  // namespace detail

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

// include/tinyopt/output.h:26-145 for ONE problem — what the P == 1 overload of Optimize returns, so that a call site of
// the reference (`const auto& out = Optimize(x, loss, options); REQUIRE(out.Succeeded());`) keeps compiling.
struct Output {
  int32_t stop_reason = kNone;
  int32_t num_iters = 0, num_failures = 0, num_consec_failures = 0;
  struct Cost_ { double cost = 0; int32_t num_resisuals = 0; float inlier_ratio = 1; } final_cost;   // cost.h:18-97
  double final_rerr_dec = 0;
  std::vector<double> final_hessian;   // n*n, undamped; empty unless options.hessian.save_last
  std::vector<double> errs, deltas2;   // one entry per iteration (only with history = true)
  std::vector<bool> successes;
  bool Succeeded() const { return stop_reason >= kNone; }
  bool Converged() const { return stop_reason >= kMinError && stop_reason < kMaxIters; }
};

template <typename Scalar, typename Cost>
BatchOutput OptimizeWithHostControls(std::vector<Scalar>& x, const Cost& cost, const Options& options, bool history);

// tinyopt::Optimize(x, cost, options) for a batch.  x: [P][n] contiguous host scalars, updated in place.
template <typename Scalar, typename Cost>
BatchOutput Optimize(std::vector<Scalar>& x, const Cost& cost, const Options& options = {}, bool history = false) {
  const int64_t P = cost.P();
  const int n = cost.n();
  if (int64_t(x.size()) != P * cost.xdim())
    throw std::invalid_argument("tinyopt_amd::Optimize: x must hold P * (parameters per problem) scalars");
  if (options.has_host_controls()) {
    return OptimizeWithHostControls(x, cost, options, history);   // (run-time models too: toa_jit_lm_begin / step / stop)
  }
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
  apply_loss(cost);
  if constexpr (detail::is_jit<Cost>::value)
    check(toa_jit_lm_run(ctx.get(), cost.jit_handle(), cost.items(), P, cost.data(), dx.data(), &pod, &r, nullptr));
  else
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
  Optimizer(std::vector<Scalar>& x, const Cost& cost, const Options& options = {}, bool history = false)
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
    if (history) {
      hist_stride_ = options.max_iters + 2;
      errs_ = DeviceBuffer<double>(cost.ctx(), size_t(P_) * hist_stride_); errs_.zero();
      d2_ = DeviceBuffer<double>(cost.ctx(), size_t(P_) * hist_stride_); d2_.zero();
      succ_ = DeviceBuffer<uint8_t>(cost.ctx(), size_t(P_) * hist_stride_); succ_.zero();
      r_.errs = errs_.data(); r_.deltas2 = d2_.data(); r_.successes = succ_.data(); r_.hist_stride = hist_stride_;
    }
    apply_loss(cost);
    if constexpr (detail::is_jit<Cost>::value)
      check(toa_jit_lm_begin(cost.ctx().get(), cost.jit_handle(), cost.items(), P_, cost.data(), dx_.data(), &pod_, &r_, state_.data()));
    else
      check(toa_lm_begin(cost.ctx().get(), Cost::model_id, dtype_of<Scalar>(), n_, cost.m(), P_, cost.data(), dx_.data(), &pod_, &r_,
                         state_.data()));
  }
  // One pass of the loop body for every running problem; returns how many are still running.
  int64_t Step() {
    const Context& ctx = cost_->ctx();
    active_.zero();
    apply_loss(*cost_);
    if constexpr (detail::is_jit<Cost>::value)
      check(toa_jit_lm_step(ctx.get(), cost_->jit_handle(), cost_->items(), P_, cost_->data(), dx_.data(), &pod_, &r_, nullptr,
                            state_.data(), active_.data()));
    else
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
    if (hist_stride_) { out.hist_stride = hist_stride_; get(out.errs, errs_); get(out.deltas2, d2_); get(out.successes, succ_); }
    return out;
  }
  // What the reference hands its stop callbacks after an iteration (optimizer.h:529-534), per problem: the cost, |dx|^2,
  // |g|^2 and — when dx / g are given — the step and gradient vectors ([P][n]).
  void StepInfo(std::vector<double>& err, std::vector<double>& dx2, std::vector<double>& g2, std::vector<Scalar>* dx = nullptr,
                std::vector<Scalar>* g = nullptr) const {
    const Context& ctx = cost_->ctx();
    DeviceBuffer<double> de(ctx, P_), dd(ctx, P_), dg(ctx, P_);
    DeviceBuffer<Scalar> vdx, vg;
    if (dx) vdx = DeviceBuffer<Scalar>(ctx, size_t(P_) * n_);
    if (g) vg = DeviceBuffer<Scalar>(ctx, size_t(P_) * n_);
    check(toa_lm_step_info(ctx.get(), dtype_of<Scalar>(), n_, P_, state_.data(), de.data(), dd.data(), dg.data(),
                           dx ? vdx.data() : nullptr, g ? vg.data() : nullptr));
    check(toa_synchronize(ctx.get()));
    err.resize(P_); dx2.resize(P_); g2.resize(P_);
    de.download(err.data()); dd.download(dx2.data()); dg.download(g2.data());
    if (dx) { dx->resize(size_t(P_) * n_); vdx.download(dx->data()); }
    if (g) { g->resize(size_t(P_) * n_); vg.download(g->data()); }
  }
  // The rest of what the reference's log line prints (optimizer.h:463-516): damping, residual count, inlier residuals.
  void StepLog(std::vector<double>& lambda, std::vector<int32_t>& nres, std::vector<int32_t>& ninl) const {
    const Context& ctx = cost_->ctx();
    DeviceBuffer<double> dl(ctx, P_);
    DeviceBuffer<int32_t> dn(ctx, P_), di(ctx, P_);
    check(toa_lm_step_log(ctx.get(), dtype_of<Scalar>(), n_, P_, state_.data(), dl.data(), dn.data(), di.data()));
    check(toa_synchronize(ctx.get()));
    lambda.resize(P_); nres.resize(P_); ninl.resize(P_);
    dl.download(lambda.data()); dn.download(nres.data()); di.download(ninl.data());
  }
  // Ends the still-running problems p with request[p] != 0 with that StopReason (kUserStopped, kTimedOut).
  void Stop(const std::vector<int32_t>& request) {
    if (int64_t(request.size()) != P_) throw std::invalid_argument("tinyopt_amd::Optimizer::Stop: one request per problem");
    const Context& ctx = cost_->ctx();
    DeviceBuffer<int32_t> dr(ctx, P_);
    dr.upload(request.data());
    if constexpr (detail::is_jit<Cost>::value)
      check(toa_jit_lm_stop(ctx.get(), cost_->jit_handle(), cost_->items(), P_, cost_->data(), dx_.data(), &pod_, &r_, nullptr,
                            state_.data(), dr.data()));
    else
      check(toa_lm_stop(ctx.get(), Cost::model_id, dtype_of<Scalar>(), n_, cost_->m(), P_, cost_->data(), dx_.data(), &pod_, &r_,
                        nullptr, state_.data(), dr.data()));
    check(toa_synchronize(ctx.get()));
  }
  void SetStopReason(const std::vector<int32_t>& stop) { stop_.upload(stop.data()); }

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
  DeviceBuffer<double> fH_, errs_, d2_;
  DeviceBuffer<uint8_t> succ_;
  int hist_stride_ = 0;
  toa_results r_;
};

// Optimize() when Options carries host-side stop controls: the loop of OptimizeAcc (optimizer.h:266-310) driven from the
// host over the stepping form.  After every pass the callbacks see (err, |dx|^2, |g|^2) / (err, dx, g) of each problem no
// numeric stop test has ended (the else-if chain of optimizer.h:519-534); the accumulated wall time is checked against
// max_duration_ms (one clock for the batch) and kTimedOut overrides whatever reason a problem picked up in that same
// pass, as the unconditional assignment at optimizer.h:303-305 does.
template <typename Scalar, typename Cost>
BatchOutput OptimizeWithHostControls(std::vector<Scalar>& x, const Cost& cost, const Options& options, bool history) {
  Optimizer<Scalar, Cost> opt(x, cost, options, history);
  const int64_t P = cost.P();
  const int n = cost.n();
  std::vector<char> running_before(P, 1);
  std::vector<double> err, dx2, g2;
  std::vector<Scalar> dxv, gv;
  std::vector<float> dxf(n), gf(n);
  double duration_ms = 0;
  // the log line's running state (optimizer.h:428-446): the last accepted cost of every logged problem
  std::vector<int64_t> logged;
  if (options.log.enable) {
    if (options.log.problems.empty()) logged.push_back(0);
    for (int64_t q : options.log.problems) if (q >= 0 && q < P) logged.push_back(q);
  }
  const double big = double(std::numeric_limits<Scalar>::max()), eps = sizeof(Scalar) == 4 ? 1e-4 : double(1e-7f);   // math.h:297-301
  std::vector<double> final_cost(logged.size(), big);
  for (int it = 0; it < int(options.max_iters) + 2; ++it) {
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<Scalar> x_before;   // (the line shows the x the iteration STARTED from: the reference forms it inside Step)
    if (!logged.empty() && options.log.print_x) x_before = x;
    const int64_t active = opt.Step();
    if (!logged.empty()) {   // optimizer.h:463-516: the same fields in the same order, one line per logged problem that made this iteration
      std::vector<double> lam;
      std::vector<int32_t> nres, ninl;
      opt.StepInfo(err, dx2, g2, options.log.print_dx ? &dxv : nullptr, nullptr);
      opt.StepLog(lam, nres, ninl);
      const double took = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      const std::vector<Scalar>& xnow = x_before;
      for (size_t li = 0; li < logged.size(); ++li) {
        const int64_t q = logged[li];
        if (!running_before[q]) continue;
        const double fc = final_cost[li], derr = err[q] - fc;
        const bool good = derr < 0.0;
        const double rel = (fc > eps && fc < big) ? (fc - err[q]) / fc : 0.0;
        char buf[256];
        std::string line;
        if (options.log.print_emoji) line += (good || it == 0) ? (it == 0 ? "\xE2\x84\xB9\xEF\xB8\x8F" : "\xE2\x9C\x85") : "\xE2\x9D\x8C";
        std::snprintf(buf, sizeof(buf), "#%d ", it);
        line += buf;
        if (options.log.print_x) {
          const size_t xd = xnow.size() / size_t(P);
          line += "x:[";
          for (size_t j = 0; j < xd; ++j) { std::snprintf(buf, sizeof(buf), j ? " %.6g" : "%.6g", double(xnow[size_t(q) * xd + j])); line += buf; }
          line += "] ";
        }
        std::snprintf(buf, sizeof(buf), "%s:%.4e n:%d d%s:%+.2e r%s:%+.1e |\xCE\xB4x|:%.2e ", options.log.e.c_str(), err[q], int(nres[q]),
                      options.log.e.c_str(), it == 0 ? 0.0 : derr, options.log.e.c_str(), rel, std::sqrt(dx2[q]));
        line += buf;
        if (options.log.print_dx) {
          line += "\xCE\xB4x:[";
          for (int j = 0; j < n; ++j) { std::snprintf(buf, sizeof(buf), j ? " %.6g" : "%.6g", double(dxv[size_t(q) * n + j])); line += buf; }
          line += "] ";
        }
        if (options.min_grad_norm2 > 0) { std::snprintf(buf, sizeof(buf), "|\xE2\x88\x87|:%.2e ", std::sqrt(g2[q])); line += buf; }
        if (options.solver_type == Options::LevenbergMarquardt && lam[q] > 0) {   // SolverLM::stateAsString (lm.h:150-154)
          std::snprintf(buf, sizeof(buf), "\xE2\x97\x8B:%.2e ", 1.0 / lam[q]);
          line += buf;
        }
        if (options.log.print_inliers) {
          std::snprintf(buf, sizeof(buf), "in:%.2f%% (%d) ", 100.0 * ninl[q] / std::max(1, int(nres[q])), int(ninl[q]));
          line += buf;
        }
        if (options.log.print_t) { std::snprintf(buf, sizeof(buf), "\xCF\x84:%.2f ", duration_ms + took); line += buf; }
        if (options.log.sink) options.log.sink(line);
        else std::cout << line << std::endl;
        if (good || it == 0) final_cost[li] = err[q];
      }
    }
    BatchOutput now = opt.output();
    std::vector<char> running(P);
    for (int64_t p = 0; p < P; ++p) running[p] = running_before[p] && now.stop_reason[p] == kNone;
    std::vector<int32_t> req(P, 0);
    bool any = false;
    // The reference evaluates the callbacks inside Step on EVERY iteration (optimizer.h:529-534) and labels kMaxIters only after
    // the loop, when stop_reason is still kNone (:320-321): a problem that used up its iterations in this very pass is
    // consulted too, and a callback returning true makes it kUserStopped, not kMaxIters.
    std::vector<char> just_max(P, 0);
    bool consult = false, relabel = false;
    for (int64_t p = 0; p < P; ++p) {
      just_max[p] = running_before[p] && now.stop_reason[p] == kMaxIters;
      consult = consult || running[p] || just_max[p];
    }
    if (consult && (options.stop_callback || options.stop_callback2)) {
      opt.StepInfo(err, dx2, g2, options.stop_callback2 ? &dxv : nullptr, options.stop_callback2 ? &gv : nullptr);
      for (int64_t p = 0; p < P; ++p) {
        if (!running[p] && !just_max[p]) continue;
        bool stop = options.stop_callback && options.stop_callback(err[p], dx2[p], g2[p]);
        if (!stop && options.stop_callback2) {
          for (int j = 0; j < n; ++j) { dxf[j] = float(dxv[size_t(p) * n + j]); gf[j] = float(gv[size_t(p) * n + j]); }
          stop = options.stop_callback2(float(err[p]), dxf, gf);
        }
        if (stop) {
          if (just_max[p]) { now.stop_reason[p] = kUserStopped; relabel = true; }   // already finalised: only the label changes
          else { req[p] = kUserStopped; any = true; }
        }
      }
      if (relabel) opt.SetStopReason(now.stop_reason);
    }
    (void)active;
    duration_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    const bool timed_out = options.max_duration_ms > 0 && duration_ms > options.max_duration_ms;
    if (timed_out)
      for (int64_t p = 0; p < P; ++p)
        if (running[p]) { req[p] = kTimedOut; any = true; }
    if (any) opt.Stop(req);
    if (timed_out) {
      BatchOutput fin = opt.output();
      bool changed = false;
      for (int64_t p = 0; p < P; ++p)
        if (running_before[p] && !running[p]) { fin.stop_reason[p] = kTimedOut; changed = true; }
      if (changed) opt.SetStopReason(fin.stop_reason);
      break;
    }
    bool left = false;
    for (int64_t p = 0; p < P; ++p) { running_before[p] = running[p] && req[p] == 0; left = left || running_before[p]; }
    if (!left) break;
  }
  return opt.output();
}

// ---- C1: a batch sharded over the GPUs of a node, one process per GPU ---------------------------------------
// RAII over toa_comm.  The 128-byte id comes from ONE rank (`Communicator::UniqueId()`) and is handed to the others by the
// host program (MPI_Bcast, a file, a socket): the library has no side channel of its own.
class Communicator {
 public:
  using Id = std::array<char, TOA_COMM_ID_BYTES>;
  static Id UniqueId() { Id id; check(toa_comm_unique_id(id.data())); return id; }
  Communicator(const Context& ctx, const Id& id, int nranks, int rank) : ctx_(&ctx), nranks_(nranks), rank_(rank) {
    check(toa_comm_init_rank(ctx.get(), id.data(), nranks, rank, &c_));
  }
  ~Communicator() { toa_comm_destroy(c_); }
  Communicator(const Communicator&) = delete;
  Communicator& operator=(const Communicator&) = delete;
  toa_comm get() const { return c_; }
  int nranks() const { return nranks_; }
  int rank() const { return rank_; }
  const Context& ctx() const { return *ctx_; }
  // this rank's contiguous block [lo, hi) of the P_total problem ids
  std::pair<int64_t, int64_t> shard(int64_t P_total) const {
    int64_t lo = 0, hi = 0;
    check(toa_shard_range(P_total, rank_, nranks_, &lo, &hi));
    return {lo, hi};
  }

 private:
  const Context* ctx_;
  toa_comm c_ = nullptr;
  int nranks_, rank_;
};

// Optimize() of this rank's shard followed by the ONE collective of the path: x_local / local_cost hold the problems
// comm.shard(P_total) of a batch of P_total.  On the root the returned BatchOutput carries stop_reason / num_iters /
// final_cost of ALL P_total problems in problem-id order and *x_all their parameters; the other ranks get their local
// Output (x_all untouched).  No communication happens during the solve.
template <typename Scalar, typename Cost>
BatchOutput ShardedOptimize(std::vector<Scalar>& x_local, const Cost& local_cost, const Options& options, const Communicator& comm,
                            int64_t P_total, std::vector<Scalar>* x_all, int root = 0) {
  const auto range = comm.shard(P_total);
  if (local_cost.P() != range.second - range.first)
    throw std::invalid_argument("tinyopt_amd::ShardedOptimize: the local model must hold this rank's shard of P_total problems");
  BatchOutput local = Optimize(x_local, local_cost, options);
  const Context& ctx = local_cost.ctx();
  const int xd = local_cost.xdim();
  const int64_t Pl = local_cost.P();
  DeviceBuffer<Scalar> dx(ctx, x_local.size() ? x_local.size() : 1);
  DeviceBuffer<int32_t> ds(ctx, Pl ? Pl : 1), di(ctx, Pl ? Pl : 1);
  DeviceBuffer<double> dc(ctx, Pl ? Pl : 1);
  if (Pl) { dx.upload(x_local.data()); ds.upload(local.stop_reason.data()); di.upload(local.num_iters.data()); dc.upload(local.final_cost.data()); }
  toa_results lr{}, ar{};
  lr.stop_reason = ds.data(); lr.num_iters = di.data(); lr.final_cost = dc.data();
  const bool is_root = comm.rank() == root;
  DeviceBuffer<Scalar> ax;
  DeviceBuffer<int32_t> as, ai;
  DeviceBuffer<double> ac;
  if (is_root) {
    ax = DeviceBuffer<Scalar>(ctx, size_t(P_total) * xd); as = DeviceBuffer<int32_t>(ctx, P_total); ai = DeviceBuffer<int32_t>(ctx, P_total);
    ac = DeviceBuffer<double>(ctx, P_total);
    ar.stop_reason = as.data(); ar.num_iters = ai.data(); ar.final_cost = ac.data();
  }
  check(toa_gather(ctx.get(), comm.get(), dtype_of<Scalar>(), xd, P_total, dx.data(), &lr, root, is_root ? ax.data() : nullptr, &ar));
  check(toa_synchronize(ctx.get()));
  if (!is_root) return local;
  BatchOutput all;
  all.stop_reason.resize(P_total); all.num_iters.resize(P_total); all.final_cost.resize(P_total);
  as.download(all.stop_reason.data()); ai.download(all.num_iters.data()); ac.download(all.final_cost.data());
  if (x_all) { x_all->resize(size_t(P_total) * xd); ax.download(x_all->data()); }
  return all;
}

// ---- the reference's own call shape: ONE parameter block, `Output out = Optimize(x, cost, options)` --------------------
// (include/tinyopt/optimize.h:16-17 takes any `T& x`).  x: a scalar, or any contiguous container with data() / size()
// (std::array, std::vector of another scalar type is excluded by the batch overload above, an Eigen::Matrix, ...) holding
// the parameters of the single problem of `cost` (cost.P() == 1).  Updated in place; returns one Output.
namespace detail {
template <typename X, typename = void> struct is_container : std::false_type {};
template <typename X> struct is_container<X, std::void_t<decltype(std::declval<X&>().data()), decltype(std::declval<X&>().size())>> : std::true_type {};
template <typename S> struct is_std_vector : std::false_type {};
template <typename S, typename A> struct is_std_vector<std::vector<S, A>> : std::true_type {};
inline Output single(const BatchOutput& b, int n, bool history) {
  Output o;
  o.stop_reason = b.stop_reason[0]; o.num_iters = b.num_iters[0]; o.num_failures = b.num_failures[0];
  o.num_consec_failures = b.num_consec_failures[0];
  o.final_cost.cost = b.final_cost[0]; o.final_cost.num_resisuals = b.final_num_residuals[0];
  o.final_cost.inlier_ratio = b.final_inlier_ratio.empty() ? 1.0f : b.final_inlier_ratio[0];
  o.final_rerr_dec = b.final_rerr_dec[0];
  if (!b.final_hessian.empty()) o.final_hessian.assign(b.final_hessian.begin(), b.final_hessian.begin() + size_t(n) * n);
  if (history && b.hist_stride > 0) {
    const int k = std::min<int>(o.num_iters, b.hist_stride);
    o.errs.assign(b.errs.begin(), b.errs.begin() + k);
    o.deltas2.assign(b.deltas2.begin(), b.deltas2.begin() + k);
    for (int i = 0; i < k; ++i) o.successes.push_back(b.successes[i] != 0);
  }
  return o;
}
}  // namespace detail

template <typename X, typename Cost, typename Scalar = std::decay_t<decltype(*std::declval<const Cost&>().data())>,
          std::enable_if_t<!detail::is_std_vector<X>::value && (detail::is_container<X>::value || std::is_arithmetic<X>::value), int> = 0>
Output Optimize(X& x, const Cost& cost, const Options& options = {}, bool history = false) {
  if (cost.P() != 1) throw std::invalid_argument("tinyopt_amd::Optimize(x, cost): this overload solves ONE problem (cost.P() == 1)");
  std::vector<Scalar> xv;
  if constexpr (std::is_arithmetic<X>::value) {
    xv.assign(1, Scalar(x));
  } else {
    xv.resize(x.size());
    for (size_t i = 0; i < xv.size(); ++i) xv[i] = Scalar(x.data()[i]);
  }
  const BatchOutput b = Optimize(xv, cost, options, history);   // throws std::invalid_argument on a size mismatch
  if constexpr (std::is_arithmetic<X>::value) {
    x = X(xv[0]);
  } else {
    for (size_t i = 0; i < xv.size(); ++i) x.data()[i] = static_cast<std::decay_t<decltype(x.data()[0])>>(xv[i]);
  }
  return detail::single(b, cost.n(), history);
}

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
  apply_loss(cost);
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
