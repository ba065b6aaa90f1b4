// Row-split ("wide") execution for few, huge problems, the stepping form's state kernels, the one-workgroup-per-problem team
// kernel and the persistent form.
#pragma once
#include "fused_kernels.hpp"

namespace toa {

// ================================================================================================
// Row-split ("wide") execution for few, huge problems (BASELINE configs C2 / C5: P = 1, m = 10^3..5*10^4).
// One wavefront per problem would leave the chip idle, so the rows of every problem are split over S chunks:
//   wide_partial_kernel  (P*S waves)  K1/K2 on one chunk each -> partial (H, g, cost) in HBM scratch
//   wide_step_kernel     (P waves)    sums the S partials in a fixed order (deterministic), then runs ONE
//                                     iteration of the same state machine (lm_iteration) with the state parked
//                                     in global memory between launches
// The host enqueues init + (partial, step) x max_iters on the stream without reading anything back: problems
// that have stopped make their later launches no-ops.
// ================================================================================================
struct WideParams {
  const void* data;
  void* x;
  long long P;
  int n, m, splits, chunk_rows;
  toa_options opt;
  toa_results res;
  unsigned long long* counters;
  void* state;     // WideState<T>[P]
  void* partials;  // T[P][splits][n*n + n + 2]  (H, g, cost, inlier residuals)
  void* hsum;      // T[P][n*n]
  int step_mode;   // stepping form: publish x and the running results at every pass
  int* active;     // stepping form (optional): += 1 per problem still running after the pass
  const int* stop_request;  // stepping form, wide_stop_kernel: [P] StopReason to impose (0 = none)
  int loss;                 // toa_set_loss
  double loss_th2;
  unsigned* sync;  // persistent form: [P][2] = (arrive, go) generation counters, then [1] abort flag; zeroed per launch
  int lds_per_wave;
};

template <typename T>
__device__ __forceinline__ void wide_copy_pods(WaveLds<T>& L, const WideParams* prm, int lane) {
  const int* src_o = reinterpret_cast<const int*>(&prm->opt);
  int* dst_o = reinterpret_cast<int*>(L.opt);
  for (int i = lane; i < int(sizeof(toa_options) / 4); i += 64) dst_o[i] = src_o[i];
  const int* src_r = reinterpret_cast<const int*>(&prm->res);
  int* dst_r = reinterpret_cast<int*>(L.res);
  for (int i = lane; i < int(sizeof(toa_results) / 4); i += 64) dst_r[i] = src_r[i];
}
template <typename T, int XD>
__global__ void __launch_bounds__(256) wide_init_kernel(const WideParams* __restrict__ prm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long p = (long long)blockIdx.x * 4 + wave;
  if (p >= prm->P) return;
  const int n = prm->n;
  WaveLds<T> L = WaveLds<T>::carve(smem + size_t(wave) * prm->lds_per_wave, n);
  wide_copy_pods(L, prm, lane);
  wave_sync();
  const int xd = XD ? XD : n;
  const T* X = static_cast<const T*>(prm->x);
  L.xs[lane] = lane < xd ? X[size_t(p) * xd + lane] : T(0);
  L.g[lane] = T(0);
  L.hd[lane] = T(0);
  L.st->acc_passes = 0; L.st->eval_passes = 0; L.st->solves = 0; L.st->problems = 0;
  lm_init<T>(L, lane);
  wide_store_state(L, static_cast<WideState<T>*>(prm->state) + p, lane);
}

template <typename Model>
__global__ void __launch_bounds__(256) wide_partial_kernel(const WideParams* __restrict__ prm) {
  using T = typename Model::Scalar;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = prm->n, S = prm->splits;
  const long long unit = (long long)blockIdx.x * 4 + wave;
  if (unit >= prm->P * S) return;
  const long long p = unit / S;
  const int sidx = int(unit % S);
  const WideState<T>* ws = static_cast<const WideState<T>*>(prm->state) + p;
  if (ws->st.stop != TOA_STOP_NONE || ws->st.iter >= ws->st.max_iters) return;  // this problem is finished
  Model model;
  model.init(n, prm->m, prm->data);
  model.set_loss(prm->loss, prm->loss_th2);
  WaveLds<T> L = WaveLds<T>::carve(model_bind_stage(model, smem + size_t(wave) * prm->lds_per_wave, n), n);
  L.xs[lane] = ws->xs[lane];
  wave_sync();
  const bool do_acc = prm->opt.solver_type != 0 || ws->st.rebuild;
  const int m4 = (prm->m + 3) & ~3;
  const int row0 = sidx * prm->chunk_rows;
  const int rows = min(prm->chunk_rows, m4 - row0);
  model.bind_chunk(p, row0, rows, n);
  const int stride = n * n + n + 2;
  T* part = static_cast<T*>(prm->partials) + (size_t(p) * S + sidx) * stride;
  T c;
  int nr;
  if (do_acc) {
    model.accumulate(L, n, lane, c, nr);
    model.write_sym(part, n, n, lane);
    wave_sync();
    if (lane < n) {
      part[lane * n + lane] = L.hd[lane];
      part[n * n + lane] = L.g[lane];
    }
  } else {
    model.evaluate(L, n, lane, c, nr);
  }
  if (lane == 0) { part[n * n + n] = c; part[n * n + n + 1] = T(model_inliers(model, -1, 0)); }  // -1: model has no robust loss
}

// Model for the step kernel: "accumulate" = fold the S chunk partials (fixed order => deterministic).
template <typename T, int NPAD, typename Manifold>
struct PartialSumModel {
  using Scalar = T;
  __device__ __forceinline__ void set_loss(int, double) {}  // no M-estimator on this family
  static constexpr int kNpad = NPAD;
  static constexpr int kXdim = Manifold::kXdim;
  const T* part;
  T* hsum;
  int S, n_, m;
  int ninl;
  bool direct = false;  // the partials live in LDS (team form, S <= 8): lane e sums element e over the chunks itself — S LDS reads
                        // and S - 1 additions instead of the transposed wave reductions that hide HBM latency in fold_small
  __device__ __forceinline__ T fold(int off) const {
    T s = 0;
    const int stride = n_ * n_ + n_ + 2;
    int k = 0;
    for (; k + 4 <= S; k += 4) {  // four loads in flight, added in index order (fixed association => deterministic)
      const T a0 = part[size_t(k) * stride + off], a1 = part[size_t(k + 1) * stride + off];
      const T a2 = part[size_t(k + 2) * stride + off], a3 = part[size_t(k + 3) * stride + off];
      s += a0; s += a1; s += a2; s += a3;
    }
    for (; k < S; ++k) s += part[size_t(k) * stride + off];
    return s;
  }
  // Small systems (n*n + n + 2 <= 64, i.e. n <= 6: BASELINE configs C2 / C5): lane j fetches the WHOLE partial of
  // chunk j with all its loads in flight at once, then each element is summed across the lanes by a fixed reduction
  // tree — ~3 us for 64 chunks, where a per-element serial walk over the chunks is 64 dependent HBM round trips.
  // Returns this lane's element total (lane e <-> element e of [H | g | cost | inliers]).
  __device__ __forceinline__ T fold_small(int lane, bool full) const {
    const int stride = n_ * n_ + n_ + 2;
    const int e0 = full ? 0 : n_ * n_ + n_;  // cost-only: just the last two elements
    T tot = 0;
    for (int base = 0; base < S; base += 64) {
      const int k = base + lane;
      const bool valid = k < S;
      const T* src = part + size_t(valid ? k : 0) * stride;
      T v[64];
#pragma unroll
      for (int e = 0; e < 64; ++e) v[e] = (valid && e >= e0 && e < stride) ? src[e] : T(0);
      if (full) {
        // two transposed 32-value reductions (~230 instructions each in fp64) instead of one all-reduce per element
        // (n^2 + n + 2 = 44 of them at n = 6, ~1 100 instructions): this fold is on the critical path of EVERY iteration of a
        // single-problem solve, executed by one wave on an otherwise idle CU
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          if (32 * half < stride) {  // wave-uniform
            T p32[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) p32[i] = v[32 * half + i];
            const T r = wave_transposed_reduce32(p32, lane);   // lane l: total of element 32 half + (l & 31)
            tot += ((lane >> 5) == half) ? r : T(0);
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 64; ++e) {
          if (e >= e0 && e < stride) {  // wave-uniform
            const T t = wave_allreduce_sum(v[e]);
            tot += (lane == e) ? t : T(0);
          }
        }
      }
    }
    return tot;
  }
  __device__ __forceinline__ bool small() const { return n_ * n_ + n_ + 2 <= 64; }
  __device__ __forceinline__ void accumulate(WaveLds<T>& L, int n, int lane, T& cost, int& nres) {
    if (small()) {
      const T tot = direct ? (lane < n * n + n + 2 ? fold(lane) : T(0)) : fold_small(lane, true);
      const int nn = n * n;
      if (lane < nn) {
        hsum[lane] = tot;
        if (lane / n == lane % n) L.hd[lane / n] = tot;
      } else if (lane < nn + n) {
        L.g[lane - nn] = tot;
      }
      cost = wave_bcast(tot, nn + n);
      ninl = int(wave_bcast(tot, nn + n + 1));
    } else {
      for (int e = lane; e < n * n; e += 64) hsum[e] = fold(e);
      if (lane < n) { L.g[lane] = fold(n * n + lane); L.hd[lane] = fold(lane * n + lane); }
      cost = fold(n * n + n);
      ninl = int(fold(n * n + n + 1));
    }
    if (ninl < 0) ninl = m;
    nres = m;
    wave_sync();
  }
  __device__ __forceinline__ void evaluate(WaveLds<T>&, int n, int lane, T& cost, int& nres) {
    if (small()) {
      const T tot = direct ? (lane >= n * n + n && lane < n * n + n + 2 ? fold(lane) : T(0)) : fold_small(lane, false);
      cost = wave_bcast(tot, n * n + n);
      ninl = int(wave_bcast(tot, n * n + n + 1));
    } else {
      cost = fold(n * n + n);
      ninl = int(fold(n * n + n + 1));
    }
    if (ninl < 0) ninl = m;
    nres = m;
  }
  template <typename O>
  __device__ __forceinline__ void write_sym(O* M, int LD, int n, int lane) const {
    for (int e = lane; e < n * n; e += 64) M[(e / n) * LD + (e % n)] = O(hsum[e]);
  }
  __device__ __forceinline__ void plus_eq(WaveLds<T>& L, const T* d, T sign, int n, int lane) const {
    Manifold::plus_eq(L, d, sign, n, lane);
  }
};

template <typename T, int NPAD, typename Manifold>
__global__ void __launch_bounds__(256) wide_step_kernel(const WideParams* __restrict__ prm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long p = (long long)blockIdx.x * 4 + wave;
  if (p >= prm->P) return;
  const int n = prm->n;
  WideState<T>* ws = static_cast<WideState<T>*>(prm->state) + p;
  if (ws->st.stop != TOA_STOP_NONE || ws->st.iter >= ws->st.max_iters) return;
  WaveLds<T> L = WaveLds<T>::carve(smem + size_t(wave) * prm->lds_per_wave, n);
  wide_copy_pods(L, prm, lane);
  wide_load_state(L, ws, lane);
  PartialSumModel<T, NPAD, Manifold> model;
  model.S = prm->splits;
  model.n_ = n;
  model.m = prm->m;
  model.part = static_cast<const T*>(prm->partials) + size_t(p) * prm->splits * (n * n + n + 2);
  model.hsum = static_cast<T*>(prm->hsum) + size_t(p) * n * n;
  const bool more = lm_iteration<T>(model, L, n, lane, p);
  if (more && prm->step_mode) {  // the reference's Step updates x in place every time (optimizer.h:271-279)
    const int xd = Manifold::kXdim ? Manifold::kXdim : n;
    T* X = static_cast<T*>(prm->x);
    if (lane < xd) X[size_t(p) * xd + lane] = L.xs[lane];
    if (lane == 0) {
      prm->res.num_iters[p] = L.st->num_iters;
      prm->res.final_cost[p] = L.st->final_cost;
      prm->res.stop_reason[p] = TOA_STOP_NONE;
      if (prm->active) atomicAdd(prm->active, 1);
    }
  }
  if (!more) {
    lm_finalize<T>(model, L, n, lane, p);  // sets a non-zero StopReason: later launches skip this problem
    const int xd = Manifold::kXdim ? Manifold::kXdim : n;
    T* X = static_cast<T*>(prm->x);
    if (lane < xd) X[size_t(p) * xd + lane] = L.xs[lane];
    if (prm->counters && lane == 0) {
      atomicAdd(&prm->counters[0], L.st->acc_passes);
      atomicAdd(&prm->counters[1], L.st->eval_passes);
      atomicAdd(&prm->counters[2], L.st->solves);
      atomicAdd(&prm->counters[3], L.st->problems);
    }
  }
  wide_store_state(L, ws, lane);
}

// Host-side stop controls of the stepping form (`Options::stop_callback`, `stop_callback2`, `max_duration_ms`;
// optimizer.h:302-305, 529-534): the host evaluates them between two toa_lm_step calls on what toa_lm_step_info reads
// back and names the problems to stop; this kernel ends those problems exactly as the loop would have — StopReason set,
// then the finalisation of OptimizeAcc (undamped final Hessian, Output fields; optimizer.h:313-321).  x already holds
// the iterate the reference would return: its Step sets the StopReason first and OptimizeAcc still applies the step
// before leaving the loop (optimizer.h:271-309), which is what the completed toa_lm_step has done.
template <typename T, int NPAD, typename Manifold>
__global__ void __launch_bounds__(256) wide_stop_kernel(const WideParams* __restrict__ prm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long p = (long long)blockIdx.x * 4 + wave;
  if (p >= prm->P) return;
  const int req = prm->stop_request[p];
  if (req == TOA_STOP_NONE) return;
  const int n = prm->n;
  WideState<T>* ws = static_cast<WideState<T>*>(prm->state) + p;
  if (ws->st.stop != TOA_STOP_NONE || ws->st.iter >= ws->st.max_iters) return;  // already finished on its own
  WaveLds<T> L = WaveLds<T>::carve(smem + size_t(wave) * prm->lds_per_wave, n);
  wide_copy_pods(L, prm, lane);
  wide_load_state(L, ws, lane);
  PartialSumModel<T, NPAD, Manifold> model;
  model.S = prm->splits;
  model.n_ = n;
  model.m = prm->m;
  model.part = static_cast<const T*>(prm->partials) + size_t(p) * prm->splits * (n * n + n + 2);
  model.hsum = static_cast<T*>(prm->hsum) + size_t(p) * n * n;
  L.st->stop = req;
  wave_sync();
  lm_finalize<T>(model, L, n, lane, p);
  if (prm->counters && lane == 0) {
    atomicAdd(&prm->counters[0], L.st->acc_passes);
    atomicAdd(&prm->counters[1], L.st->eval_passes);
    atomicAdd(&prm->counters[2], L.st->solves);
    atomicAdd(&prm->counters[3], L.st->problems);
  }
  wide_store_state(L, ws, lane);
}

// What the host-side stop controls look at after a step (optimizer.h:529-534: `stop_callback(err, |dx|^2, |g|^2)`,
// `stop_callback2(err, dx, g)`): the cost, step and gradient of each problem's LAST iteration, out of the state block.
template <typename T>
__global__ void __launch_bounds__(256) step_info_kernel(const void* state_, long long P, int n, double* err, double* dx2,
                                                        double* g2, T* dx_out, T* g_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long p = (long long)blockIdx.x * 4 + wave;
  if (p >= P) return;
  const WideState<T>* ws = static_cast<const WideState<T>*>(state_) + p;
  const T d = lane < n ? ws->dx[lane] : T(0);
  const T g = lane < n ? ws->g[lane] : T(0);
  const double sd = double(wave_allreduce_sum(d * d));   // same arithmetic as lm_judge_step (optimizer.h:412-415)
  const double sg = double(wave_allreduce_sum(g * g));
  if (lane == 0) {
    if (err) err[p] = ws->st.cost_val;
    if (dx2) dx2[p] = sd;
    if (g2) g2[p] = sg;
  }
  if (dx_out && lane < n) dx_out[size_t(p) * n + lane] = d;
  if (g_out && lane < n) g_out[size_t(p) * n + lane] = g;
}

// toa_lm_step_log: what the per-iteration log line prints besides step_info's numbers (optimizer.h:463-516)
template <typename T>
__global__ void __launch_bounds__(256) step_log_kernel(const void* state_, long long P, double* lambda, int* nres, int* ninl) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const WideState<T>* ws = static_cast<const WideState<T>*>(state_) + p;
  if (lambda) lambda[p] = double(ws->st.lambda);
  if (nres) nres[p] = ws->st.cost_nres;
  if (ninl) ninl[p] = ws->st.cost_ninl;
}

// ------------------------------------------------------------------------------------------------
// Persistent form of the row-split solve: ONE launch for the whole solve instead of 1 + 2 x (max_iters + 1).
// The multi-launch form is bound by the GPU's kernel-to-kernel dependency latency (~10 us per launch, the same
// eager or replayed from a hipGraph); here the S chunk-waves of a problem stay resident and hand over through two
// generation counters in HBM:
//     every chunk wave: partial (H, g, cost) of its rows -> HBM, release, arrive += 1
//     leader (chunk 0): waits arrive == S * gen, acquire, folds the partials in fixed order, runs ONE lm_iteration
//                       (state in its LDS for the whole solve), publishes x + flags, release, go = gen + 1
//     the others      : wait go > gen, acquire, pick up x (or leave when the problem has stopped)
// Agent-scope release/acquire fences order the HBM hand-over across XCDs (separate L2s).  The launcher uses this
// form only when every workgroup is certainly co-resident (P * S <= #CUs, one 64-thread workgroup each); the waits
// poll with s_sleep and give up after ~5 s (abort flag -> StopReason kTimedOut) instead of hanging the device.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool persistent_wait(unsigned* addr, const unsigned target, unsigned* abort_flag) {
  const unsigned long long t0 = wall_clock64();  // constant 100 MHz
  for (;;) {
    if (__hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
    if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
    if (wall_clock64() - t0 > 500000000ull) {  // 5 s at the constant 100 MHz
      __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
    __builtin_amdgcn_s_sleep(2);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return true;
}

// Team form for ONE SMALL problem per workgroup (BASELINE config C2: n = 6, 1000 residuals, 56 KB): the S chunk-waves
// of a problem are the waves of ONE workgroup, their partials and the leader's x / flags live in LDS, and the two
// hand-overs of an iteration are workgroup barriers (~1 us) instead of release / acquire round trips through HBM
// (~10 us each across XCDs).  Same arithmetic in the same order as the persistent form (fixed-order fold of S partials).
template <typename Model, int NPAD, typename Manifold>
__global__ void __launch_bounds__(512) wide_team_kernel(const WideParams* __restrict__ prm) {
  using T = typename Model::Scalar;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = prm->n, S = prm->splits;  // S == blockDim.x / 64
  const long long p = blockIdx.x;
  const bool leader = wave == 0;
  const size_t pw = size_t(prm->lds_per_wave);
  WaveLds<T> L = WaveLds<T>::carve(smem + size_t(wave) * pw, n);
  const int stride = n * n + n + 2;
  T* parts = reinterpret_cast<T*>(smem + size_t(S) * pw);
  T* xshare = parts + size_t(S) * stride;
  int* flags = reinterpret_cast<int*>(xshare + 64);  // [0] stopped, [1] rebuild
  const int xd = Manifold::kXdim ? Manifold::kXdim : n;

  const int m4 = (prm->m + 3) & ~3;
  const int row0 = wave * prm->chunk_rows;
  const int rows = min(prm->chunk_rows, m4 - row0);
  Model model;
  model.init(n, prm->m, prm->data);
  model.set_loss(prm->loss, prm->loss_th2);
  static_assert(ModelStageBytes<Model>::value == 0, "the team form of the row-split kernels has no room for a model's LDS stage");
  model.bind_chunk(p, row0, rows, n);
  T* part = parts + size_t(wave) * stride;

  PartialSumModel<T, NPAD, Manifold> fold;
  fold.S = S; fold.n_ = n; fold.m = prm->m;
  fold.direct = true;
  fold.part = parts;
  fold.hsum = (n * n <= 64) ? L.aux : static_cast<T*>(prm->hsum) + size_t(p) * n * n;

  if (leader) {
    wide_copy_pods(L, prm, lane);
    wave_sync();
    const T* X = static_cast<const T*>(prm->x);
    L.xs[lane] = lane < xd ? X[size_t(p) * xd + lane] : T(0);
    L.g[lane] = T(0);
    L.hd[lane] = T(0);
    L.st->acc_passes = 0; L.st->eval_passes = 0; L.st->solves = 0; L.st->problems = 0;
    lm_init<T>(L, lane);
    xshare[lane] = L.xs[lane];
    if (lane == 0) { flags[0] = 0; flags[1] = L.st->rebuild; }
  }
  __syncthreads();
#ifdef TOA_TEAM_TIMING
  unsigned long long tt_[4] = {0, 0, 0, 0}, ttp_ = wall_clock64();
#define TEAM_TICK(i) { const unsigned long long n_ = wall_clock64(); tt_[i] += n_ - ttp_; ttp_ = n_; }
#else
#define TEAM_TICK(i)
#endif
  for (;;) {
    if (flags[0] != 0) break;  // workgroup-uniform: written before the barrier every wave has just passed
    if (!leader) { L.xs[lane] = xshare[lane]; wave_sync(); }
    const bool do_acc = prm->opt.solver_type != 0 || flags[1] != 0;
    T c;
    int nr;
    if (do_acc) {
      model.accumulate(L, n, lane, c, nr);
      model.write_sym(part, n, n, lane);
      wave_sync();
      if (lane < n) {
        part[lane * n + lane] = L.hd[lane];
        part[n * n + lane] = L.g[lane];
      }
    } else {
      model.evaluate(L, n, lane, c, nr);
    }
    if (lane == 0) { part[n * n + n] = c; part[n * n + n + 1] = T(model_inliers(model, -1, 0)); }
    TEAM_TICK(0)
    __syncthreads();
    TEAM_TICK(1)
    if (leader) {
      const bool more = lm_iteration<T>(fold, L, n, lane, p);
      TEAM_TICK(2)
      if (!more) {
        lm_finalize<T>(fold, L, n, lane, p);
        T* X = static_cast<T*>(prm->x);
        if (lane < xd) X[size_t(p) * xd + lane] = L.xs[lane];
        if (prm->counters && lane == 0) {
          atomicAdd(&prm->counters[0], L.st->acc_passes);
          atomicAdd(&prm->counters[1], L.st->eval_passes);
          atomicAdd(&prm->counters[2], L.st->solves);
          atomicAdd(&prm->counters[3], L.st->problems);
        }
      }
      xshare[lane] = L.xs[lane];
      if (lane == 0) { flags[0] = more ? 0 : 1; flags[1] = L.st->rebuild; }
    }
    __syncthreads();
    TEAM_TICK(3)
  }
#ifdef TOA_TEAM_TIMING
  if (threadIdx.x == 0 && blockIdx.x == 0)
    printf("team p=0 S=%d: data pass %.1f us  barrier %.1f us  iteration %.1f us  publish + barrier %.1f us\n", S, tt_[0] * 0.01, tt_[1] * 0.01,
           tt_[2] * 0.01, tt_[3] * 0.01);
#endif
}

template <typename Model, int NPAD, typename Manifold>
__global__ void __launch_bounds__(64) wide_persistent_kernel(const WideParams* __restrict__ prm) {
  using T = typename Model::Scalar;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const int n = prm->n, S = prm->splits;
  const long long p = (long long)blockIdx.x / S;
  const int sidx = int((long long)blockIdx.x % S);
  const bool leader = sidx == 0;
  WaveLds<T> L = WaveLds<T>::carve(smem + stage_carve_off<T>(ModelStageBytes<Model>::value, n), n);
  WideState<T>* ws = static_cast<WideState<T>*>(prm->state) + p;  // mailbox: x + flags published by the leader
  unsigned* arrive = prm->sync + 2 * p;
  unsigned* go = arrive + 1;
  unsigned* abort_flag = prm->sync + 2 * prm->P;
  const int xd = Manifold::kXdim ? Manifold::kXdim : n;

  const int m4 = (prm->m + 3) & ~3;
  const int row0 = sidx * prm->chunk_rows;
  const int rows = min(prm->chunk_rows, m4 - row0);
  Model model;
  model.init(n, prm->m, prm->data);
  model.set_loss(prm->loss, prm->loss_th2);
  (void)model_bind_stage(model, smem, n);
  model.bind_chunk(p, row0, rows, n);
  const int stride = n * n + n + 2;
  T* part = static_cast<T*>(prm->partials) + (size_t(p) * S + sidx) * stride;

  PartialSumModel<T, NPAD, Manifold> fold;
  fold.S = S; fold.n_ = n; fold.m = prm->m;
  fold.part = static_cast<const T*>(prm->partials) + size_t(p) * S * stride;
  // n <= 8: the folded H stays in LDS (L.aux) instead of making an HBM round trip between fold and factorisation
  fold.hsum = (n * n <= 64) ? L.aux : static_cast<T*>(prm->hsum) + size_t(p) * n * n;

  auto publish = [&](int stop) __attribute__((always_inline)) {
    ws->xs[lane] = L.xs[lane];
    if (lane == 0) { ws->st.rebuild = L.st->rebuild; ws->st.stop = stop; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  };

  if (leader) {
    wide_copy_pods(L, prm, lane);
    wave_sync();
    const T* X = static_cast<const T*>(prm->x);
    L.xs[lane] = lane < xd ? X[size_t(p) * xd + lane] : T(0);
    L.g[lane] = T(0);
    L.hd[lane] = T(0);
    L.st->acc_passes = 0; L.st->eval_passes = 0; L.st->solves = 0; L.st->problems = 0;
    lm_init<T>(L, lane);
    publish(TOA_STOP_NONE);
    if (lane == 0) __hip_atomic_store(go, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

#ifdef TOA_PERSIST_TIMING
  unsigned long long tk[4] = {0, 0, 0, 0}, tprev = wall_clock64();
#define TOA_TICK(i) { const unsigned long long tn = wall_clock64(); tk[i] += tn - tprev; tprev = tn; }
#else
#define TOA_TICK(i)
#endif
  for (unsigned gen = 1;; ++gen) {
    bool do_acc;
    if (leader) {
      do_acc = prm->opt.solver_type != 0 || L.st->rebuild;
    } else {
      if (!persistent_wait(go, gen, abort_flag)) return;
      if (ws->st.stop != TOA_STOP_NONE) return;  // the problem has finished
      L.xs[lane] = ws->xs[lane];
      wave_sync();
      do_acc = prm->opt.solver_type != 0 || ws->st.rebuild;
    }
    T c;
    int nr;
    if (do_acc) {
      model.accumulate(L, n, lane, c, nr);
      model.write_sym(part, n, n, lane);
      wave_sync();
      if (lane < n) {
        part[lane * n + lane] = L.hd[lane];
        part[n * n + lane] = L.g[lane];
      }
    } else {
      model.evaluate(L, n, lane, c, nr);
    }
    if (lane == 0) { part[n * n + n] = c; part[n * n + n + 1] = T(model_inliers(model, -1, 0)); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!leader) continue;
    TOA_TICK(0)
    if (!persistent_wait(arrive, unsigned(S) * gen, abort_flag)) {
      if (lane == 0) prm->res.stop_reason[p] = TOA_STOP_TIMED_OUT;
      return;
    }
    TOA_TICK(1)
    const bool more = lm_iteration<T>(fold, L, n, lane, p);
    TOA_TICK(2)
    if (!more) {
      lm_finalize<T>(fold, L, n, lane, p);
      T* X = static_cast<T*>(prm->x);
      if (lane < xd) X[size_t(p) * xd + lane] = L.xs[lane];
      if (prm->counters && lane == 0) {
        atomicAdd(&prm->counters[0], L.st->acc_passes);
        atomicAdd(&prm->counters[1], L.st->eval_passes);
        atomicAdd(&prm->counters[2], L.st->solves);
        atomicAdd(&prm->counters[3], L.st->problems);
      }
    }
    publish(more ? TOA_STOP_NONE : (L.st->stop != TOA_STOP_NONE ? L.st->stop : TOA_STOP_MAX_ITERS));
    if (lane == 0) __hip_atomic_store(go, gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    TOA_TICK(3)
#ifdef TOA_PERSIST_TIMING
    if (!more && lane == 0)
      printf("persistent p=%lld S=%d gens=%u  partial %.1f us  wait %.1f us  iteration %.1f us  publish+finalize %.1f us\n", p, S, gen,
             tk[0] * 0.01, tk[1] * 0.01, tk[2] * 0.01, tk[3] * 0.01);
#endif
    if (!more) return;
  }
}

}  // namespace toa
