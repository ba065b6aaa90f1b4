// The one-wavefront-per-problem kernels: lm_fused_kernel (whole LM solves in one launch: K1 + K2 + K3 + K4, dynamic problem
// queue, memo, cooperative tail) and the seams accumulate_kernel (K1 / K2), solve_damped_kernel (K3), inv_cov_kernel.
#pragma once
#include "models_jet.hpp"

namespace toa {

// Per-problem LM state parked in HBM between launches: the stepping form (`Optimizer_::Step`, optimizer.h:331-539, one
// loop pass per call) and the launch-per-iteration row-split path both resume the same state machine from it.
template <typename T>
struct WideState {
  LmState<T> st;
  T xs[64], g[64], hd[64], dx[64], ldx[64];
};

template <typename T>
__device__ __forceinline__ void wide_load_state(WaveLds<T>& L, const WideState<T>* ws, int lane) {
  const int* src = reinterpret_cast<const int*>(&ws->st);
  int* dst = reinterpret_cast<int*>(L.st);
  for (int i = lane; i < int(sizeof(LmState<T>) / 4); i += 64) dst[i] = src[i];
  L.xs[lane] = ws->xs[lane]; L.g[lane] = ws->g[lane]; L.hd[lane] = ws->hd[lane];
  L.dx[lane] = ws->dx[lane]; L.ldx[lane] = ws->ldx[lane];
  wave_sync();
}
template <typename T>
__device__ __forceinline__ void wide_store_state(const WaveLds<T>& L, WideState<T>* ws, int lane) {
  wave_sync();
  const int* src = reinterpret_cast<const int*>(L.st);
  int* dst = reinterpret_cast<int*>(&ws->st);
  for (int i = lane; i < int(sizeof(LmState<T>) / 4); i += 64) dst[i] = src[i];
  ws->xs[lane] = L.xs[lane]; ws->g[lane] = L.g[lane]; ws->hd[lane] = L.hd[lane];
  ws->dx[lane] = L.dx[lane]; ws->ldx[lane] = L.ldx[lane];
}

struct FusedParams {
  const void* data;
  void* x;
  long long P;
  int n, m;
  toa_options opt;
  toa_results res;
  unsigned long long* counters;  // [4] or null
  int* queue;                    // [0] pop counter, [16] waves that have left the kernel (separate cache lines)
  unsigned long long* timeline;  // debug (toa_debug_timeline): [P][2] start / end of every problem in 100 MHz ticks
  int lds_per_wave;
  int mode;                      // 0: whole solve; 1: begin (state <- x0, lm_init); 2: ONE loop pass per problem (stepping form);
                                 // 3: finalise the problems named in stop_request with that StopReason (host-side stop controls)
  void* state;                   // modes 1, 2, 3: caller's state block (see launch_wide)
  int* active;                   // mode 2 (optional): += 1 per problem that is still running after this pass
  const int* stop_request;       // mode 3: [P] StopReason to impose on a still-running problem (0 = leave it running)
  int loss;                      // TOA_LOSS_* of the handle (toa_set_loss): applied per residual by the DenseRow / Jet families
  double loss_th2;
  void* memo;                    // mode 0, models with kMemo: one slot of memo_stride bytes per resident wave (null = off)
  unsigned long long memo_stride;
  int stage_off;                 // row-per-lane fp64 pass: byte offset of the wave's LDS stage in its LDS region
  int carve_off;                 // byte offset of the wave's carve in its LDS region (> 0: a stage in front of it, overlapping the carve's
                                 // pass-dead head — WaveLds::pass_dead_bytes)
  int coop_tot_off;              // cooperative passes: byte offset of the chunk-partial total in a wave's carve (0 = its M)
  int memo_lds_off;              // != 0: the memo slot is in LDS instead, at this byte offset of the wave's carve (small Grams)
  int coop_K;                    // cooperative passes (CoopCtl): chunks per pass, 0 = off
  int coop_cs;                   // steps (of 4 rows) per chunk, a multiple of the load ring's period
  int reserved_[2];              // (the team form's two fields, round 5: the block's layout is unchanged)
};

template <typename M, typename = void>
struct ModelStageBytes { static constexpr size_t value = 0; };
template <typename M>
struct ModelStageBytes<M, std::enable_if_t<(M::kStageBytes > 0)>> { static constexpr size_t value = M::kStageBytes; };
// A staged model's LDS stage begins at the wave's region and ends INSIDE its carve, over the part of it that is dead while a
// pass runs (WaveLds::pass_dead_bytes): the carve starts stage_carve_off bytes into the region.  One rule for every kernel and
// for the host's LDS sizing (lds_fit): a function of the stage size and n only.
template <typename T>
__host__ __device__ inline size_t stage_carve_off(size_t stage_bytes, int n) {
  const size_t st = (stage_bytes + 15) & ~size_t(15), dead = WaveLds<T>::pass_dead_bytes(n);
  return st > dead ? st - dead : 0;
}
// launch-per-iteration kernels: binds the model's stage, returns where the wave's carve starts
template <typename Model>
__device__ __forceinline__ char* model_bind_stage(Model& model, char* wave_base, int n) {
  if constexpr (ModelStageBytes<Model>::value > 0) {
    model.stage = reinterpret_cast<unsigned char*>(wave_base);
    return wave_base + stage_carve_off<typename Model::Scalar>(ModelStageBytes<Model>::value, n);
  } else {
    (void)model; (void)n;
    return wave_base;
  }
}
template <typename M, typename = void>
struct ModelWaves { static constexpr int value = 4; };
template <typename M>
struct ModelWaves<M, std::enable_if_t<(M::kWaves > 0)>> { static constexpr int value = M::kWaves; };
template <typename M, typename = void>
struct ModelCoop { static constexpr bool value = false; };
template <typename M>
struct ModelCoop<M, std::enable_if_t<M::kCoop>> { static constexpr bool value = true; };

// (An occupancy request via __launch_bounds__'s second argument is NOT usable here: under the tighter register budget
// hipcc parks the destination registers of the in-flight asm loads in AGPRs right after issuing them — tools/isa_lint.py
// caught exactly that when 5 waves/SIMD were requested for the fp64 n <= 15 kernel.)
template <typename Model>
#ifndef TOA_FUSED_ATTR
#define TOA_FUSED_ATTR   // run-time builds may ask for an occupancy here (jit.hip: __attribute__((amdgpu_waves_per_eu(3, 3))))
#endif
__global__ void __launch_bounds__(64 * ModelWaves<Model>::value) TOA_FUSED_ATTR lm_fused_kernel(const FusedParams* __restrict__ prm_g) {
  using T = typename Model::Scalar;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int kW = ModelWaves<Model>::value;   // waves per workgroup
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int n = prm_g->n;
  constexpr int NO = kW;                         // every wave of a workgroup pulls problems
  constexpr bool owner = true;
  WaveLds<T> L = WaveLds<T>::carve(smem + size_t(wave) * prm_g->lds_per_wave + prm_g->carve_off, n);
  // private per-wave copies of the option / result PODs (no inter-wave synchronisation anywhere)
  {
    const int* src_o = reinterpret_cast<const int*>(&prm_g->opt);
    int* dst_o = reinterpret_cast<int*>(L.opt);
    for (int i = lane; i < int(sizeof(toa_options) / 4); i += 64) dst_o[i] = src_o[i];
    const int* src_r = reinterpret_cast<const int*>(&prm_g->res);
    int* dst_r = reinterpret_cast<int*>(L.res);
    for (int i = lane; i < int(sizeof(toa_results) / 4); i += 64) dst_r[i] = src_r[i];
    L.st->acc_passes = 0; L.st->eval_passes = 0; L.st->solves = 0; L.st->problems = 0; L.st->reused_passes = 0;
    L.st->memo_slot = 0;
    if constexpr (ModelMemo<Model>::value) {
      if (prm_g->memo_lds_off)   // (a generic pointer into LDS: the flat stores / loads of memo_save / memo_load reach it too)
        L.st->memo_slot = reinterpret_cast<unsigned long long>(static_cast<void*>(smem + size_t(wave) * prm_g->lds_per_wave + prm_g->memo_lds_off));
      else if (prm_g->memo)
        L.st->memo_slot = owner ? reinterpret_cast<unsigned long long>(prm_g->memo) + (size_t(blockIdx.x) * NO + wave) * prm_g->memo_stride : 0ull;
    }
  }
  wave_sync();
  const long long P = prm_g->P;
  Model model;
  model.init(n, prm_g->m, prm_g->data);
  model.set_loss(prm_g->loss, prm_g->loss_th2);
  if constexpr (ModelStageBytes<Model>::value > 0) model.stage = reinterpret_cast<unsigned char*>(smem) + size_t(wave) * prm_g->lds_per_wave + prm_g->stage_off;
  T* X = static_cast<T*>(prm_g->x);
  const int xd = Model::kXdim ? Model::kXdim : n;  // stored parameters per problem (SE3: 12 for n = 6)
  int* queue = prm_g->queue;
  if constexpr (ModelCoop<Model>::value) {   // the workgroup's control block: every wave marks itself as an owner, no pass open
    model.coop_init(prm_g->coop_K, prm_g->coop_cs, prm_g->lds_per_wave, prm_g->coop_tot_off);
    CoopCtl* ctl = reinterpret_cast<CoopCtl*>(smem + size_t(kW) * prm_g->lds_per_wave);
    if (lane == 0) {
      ctl->slot[wave].ticket = prm_g->coop_K;
      ctl->slot[wave].turn = prm_g->coop_K;
      ctl->active[wave] = owner ? 1 : 0;
    }
    __syncthreads();   // the only workgroup barrier of the kernel: nobody scans the slots before they exist
  }
  int solved = 0;
  // The first problem of every wave is assigned statically (wave w of the launch takes problem w); the shared counter hands
  // out the rest.  4 096 waves popping the same address at launch time serialise in the L2 (~5 ns per atomic = 20-30 us
  // before the last wave has its first problem: 4 % of a C3 launch, visible in the launch timeline).
  const int nwaves = int(gridDim.x) * NO;   // (owners of the launch)
  bool first = owner, dry = !owner;           // (a helper of the team form starts where an owner ends up: queue dry, looking for tickets)
  for (;;) {  // one work item = one whole problem
    int p = 0;
    if (first) {
      p = int(blockIdx.x) * NO + wave;
      first = false;
    } else if (dry) {
      p = int(P);
    } else {
      // (An "end game" that stops two — or three — waves of every workgroup from pulling once fewer than two problems per
      //  workgroup are left, so that each remaining problem is worked by an owner plus helpers, was measured and rejected: C4
      //  12.93 -> 12.60 M it/s with two pullers, 12.07 with one (profiles/r03_ab_log.md).  A helper shares the Accumulate
      //  passes only; the solve, the step test and the evaluate-only passes stay with the owner, and two waves on one problem
      //  are well short of twice as fast.)
      if (lane == 0) p = atomicAdd(queue, 1) + nwaves;
      p = __builtin_amdgcn_readfirstlane(p);
    }
    bool ghost = false;
    if (p >= P) {
      // The queue is dry.  A wave of a cooperative model does not leave yet: it looks for a sibling's open Accumulate pass,
      // takes a chunk ticket of it (DenseRowModel::coop_find) and runs a GHOST problem through the very same state-machine
      // code — whose single Accumulate call is where the chunk loop lives (DenseRowModel::coop_acc: hipcc tolerates
      // exactly one MFMA loop per kernel).  That call works the ticket (and the pass's remaining ones) off, then reports
      // "no residuals", which ends the ghost at once (kSkipped, optimizer.h:372-375) with nothing written anywhere (p < 0);
      // the wave comes back here for the next ticket until no sibling is active any more.
      dry = true;
      if constexpr (ModelCoop<Model>::value) {
        if (model.coop_K > 1) ghost = model.coop_find(lane);
      }
      if (!ghost) break;
    }
    // Fairness between the waves of a SIMD.  The issue arbiter serves the OLDEST wave first, and a wave keeps its age for
    // the whole (persistent) kernel: the launch timeline shows the oldest wave of each SIMD solving a problem in 0.8 ms
    // while the youngest needs up to 6.9 ms for its first one and is still far from done when the queue runs dry — the
    // drain is then as long as those starved problems.  Priority outranks age, so the waves that are behind are given
    // the issue slots: a wave drops one level per problem it has finished.  (Measured and rejected, profiles/r02_ab_log.md:
    // no priorities; a level that follows the lag behind the average wave; re-queueing unfinished problems iteration by
    // iteration through HBM during the drain.)
    {
      const int lag = 1 - solved;
      if (lag >= 1) __builtin_amdgcn_s_setprio(3);
      else if (lag == 0) __builtin_amdgcn_s_setprio(2);
      else if (lag == -1) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
    ++solved;
    if (ghost) p = 0;
    model.bind(p);
    wave_sync();
    L.xs[lane] = (lane < xd && !ghost) ? X[size_t(p) * xd + lane] : T(0);
    wave_sync();
    const unsigned long long tl0 = prm_g->timeline ? wall_clock64() : 0ull;
    lm_solve_problem<T>(model, L, n, lane, ghost ? -1ll : (long long)p);
    if (ghost) {
      if (lane == 0) L.st->acc_passes -= 1;   // the ghost's Build streamed nothing of its own
      continue;
    }
    if (lane < xd) X[size_t(p) * xd + lane] = L.xs[lane];
    if (prm_g->timeline && lane == 0) { prm_g->timeline[2 * size_t(p)] = tl0; prm_g->timeline[2 * size_t(p) + 1] = wall_clock64(); }
  }
  unsigned long long* counters = prm_g->counters;
  if (counters && lane == 0) {
    atomicAdd(&counters[0], L.st->acc_passes);
    atomicAdd(&counters[1], L.st->eval_passes);
    atomicAdd(&counters[2], L.st->solves);
    atomicAdd(&counters[3], L.st->problems);
    if (L.st->reused_passes) atomicAdd(&counters[4], L.st->reused_passes);
  }
  // The work queue cleans itself: the last wave to leave puts the pop counter (and this exit counter) back to zero, so the
  // next launch on the stream needs no memset in front of it (one stream operation, ~5 us, per solve: 1 % of a C3 launch).
  if (lane == 0) {
    const int gone = atomicAdd(&queue[16], 1);
    if (gone == int(gridDim.x) * kW - 1) {
      __hip_atomic_store(&queue[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&queue[16], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// K1/K2 seam: one wave per problem (grid-stride), writes g [P][n], H [P][n*n], cost, nres.
template <typename Model>
__global__ void __launch_bounds__(256) accumulate_kernel(const void* data_, const void* x_, long long P, int n, int m,
                                                         int want_grad, void* g_, void* H_, double* cost, int* nres,
                                                         int lds_per_wave, int loss, double loss_th2) {
  using T = typename Model::Scalar;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T* X = static_cast<const T*>(x_);
  Model model;
  model.init(n, m, data_);
  model.set_loss(loss, loss_th2);
  WaveLds<T> L = WaveLds<T>::carve(model_bind_stage(model, smem + size_t(wave) * lds_per_wave, n), n);
  const int xd = Model::kXdim ? Model::kXdim : n;
  for (long long p = (long long)blockIdx.x * 4 + wave; p < P; p += (long long)gridDim.x * 4) {
    wave_sync();
    L.xs[lane] = lane < xd ? X[size_t(p) * xd + lane] : T(0);
    wave_sync();
    model.bind(p);
    T c;
    int nr;
    if (want_grad) {
      model.accumulate(L, n, lane, c, nr);
      T* G = static_cast<T*>(g_) + size_t(p) * n;
      T* H = static_cast<T*>(H_) + size_t(p) * n * n;
      if (lane < n) G[lane] = L.g[lane];
      model.write_sym(H, n, n, lane);
      wave_sync();
      if (lane < n) H[lane * n + lane] = L.hd[lane];  // the (undamped) diagonal always comes from hd
    } else {
      model.evaluate(L, n, lane, c, nr);
    }
    if (lane == 0) { cost[p] = double(c); if (nres) nres[p] = nr; }
  }
}

// K3 seam: H_ii *= scale (double), dx = -H^-1 g with Eigen's acceptance rule.
template <typename T, int NPAD>
__global__ void __launch_bounds__(256) solve_damped_kernel(const void* H_, const void* g_, long long P, int n,
                                                           double scale, void* dx_, int* ok_, int lds_per_wave) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  WaveLds<T> L = WaveLds<T>::carve(smem + size_t(wave) * lds_per_wave, n);
  const T* Hg = static_cast<const T*>(H_);
  const T* gg = static_cast<const T*>(g_);
  T* dxg = static_cast<T*>(dx_);
  for (long long p = (long long)blockIdx.x * 4 + wave; p < P; p += (long long)gridDim.x * 4) {
    wave_sync();
    const T* H = Hg + size_t(p) * n * n;
    // upper triangle is authoritative (math.h:235 selfadjointView<Upper>): M[i][j] = H(min,max) (col-major)
    auto fill = [&]() __attribute__((always_inline)) {
      for (int e = lane; e < n * n; e += 64) {
        const int i = e / n, j = e % n;
        const int a = i < j ? i : j, b = i < j ? j : i;
        T v = H[size_t(b) * n + a];
        if (i == j) v = T(double(v) * scale);
        L.M[i * L.LD + j] = v;
      }
      wave_sync();
    };
    fill();
    const T gl = lane < n ? gg[size_t(p) * n + lane] : T(0);
    bool ok;
    T dx = 0;
    {
      LdltFast<T, NPAD> F;
      ok = F.factor(L.M, L.LD, n, lane);
      if (ok) dx = F.solve(L.M, L.LD, n, lane, -gl);
      else if (LdltFast<T, NPAD>::kClobbersM) { wave_sync(); fill(); }
    }
    if (!ok) {
      ok = ldlt_factor_wave<T>(L.M, L.LD, L.perm, L.tmp, n, lane);
      if (ok) dx = ldlt_solve_wave<T>(L.M, L.LD, L.perm, L.vec, n, lane, -gl);
    }
    if (lane < n) dxg[size_t(p) * n + lane] = dx;
    if (lane == 0) ok_[p] = ok ? 1 : 0;
  }
}

// Covariance seam: C = H^-1 by LDL^T against the identity — tinyopt::InvCov / DenseInvCov (include/tinyopt/math.h:41-57:
// `chol = m.selfadjointView<Upper>().ldlt(); if (Success && isPositive()) return chol.solve(Identity)`; cols()==1:
// unprotected 1/m), used by Output::Covariance (output.h:80-94) and SolverLM::Covariance (lm.h:174).
template <typename T, int NPAD>
__global__ void __launch_bounds__(256) inv_cov_kernel(const void* H_, long long P, int n, void* C_, int* ok_, int lds_per_wave) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  WaveLds<T> L = WaveLds<T>::carve(smem + size_t(wave) * lds_per_wave, n);
  const T* Hg = static_cast<const T*>(H_);
  T* Cg = static_cast<T*>(C_);
  for (long long p = (long long)blockIdx.x * 4 + wave; p < P; p += (long long)gridDim.x * 4) {
    wave_sync();
    const T* H = Hg + size_t(p) * n * n;
    T* C = Cg + size_t(p) * n * n;
    if (n == 1) {  // math.h:49-50
      if (lane == 0) { C[0] = T(1) / H[0]; ok_[p] = 1; }
      continue;
    }
    auto fill = [&]() __attribute__((always_inline)) {
      for (int e = lane; e < n * n; e += 64) {
        const int i = e / n, j = e % n;
        const int a = i < j ? i : j, b = i < j ? j : i;
        L.M[i * L.LD + j] = H[size_t(b) * n + a];  // upper triangle is authoritative
      }
      wave_sync();
    };
    fill();
    LdltFast<T, NPAD> F;
    bool ok = F.factor(L.M, L.LD, n, lane);
    if (ok) {
      for (int j = 0; j < n; ++j) {
        const T x = F.solve(L.M, L.LD, n, lane, lane == j ? T(1) : T(0));
        if (lane < n) C[size_t(j) * n + lane] = x;  // column j (symmetric: row j)
      }
    } else {
      if (LdltFast<T, NPAD>::kClobbersM) { wave_sync(); fill(); }
      ok = ldlt_factor_wave<T>(L.M, L.LD, L.perm, L.tmp, n, lane);
      if (ok)
        for (int j = 0; j < n; ++j) {
          const T x = ldlt_solve_wave<T>(L.M, L.LD, L.perm, L.vec, n, lane, lane == j ? T(1) : T(0));
          if (lane < n) C[size_t(j) * n + lane] = x;
        }
    }
    if (!ok)  // rejected (std::nullopt in the reference): define the output instead of leaving caller memory untouched
      for (int e = lane; e < n * n; e += 64) C[e] = T(0);
    if (lane == 0) ok_[p] = ok ? 1 : 0;
  }
}

}  // namespace toa
