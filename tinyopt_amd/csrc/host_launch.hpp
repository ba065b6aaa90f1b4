// Host side shared by the translation units: the handle (toa_context), error plumbing, the device guard, LDS budgeting and the
// launchers of the kernels in fused_kernels.hpp / wide_kernels.hpp.  Never seen by hiprtc.
#pragma once
#ifdef __HIPCC_RTC__
#error "host_launch.hpp is host code: run-time builds include kernels.hpp, which leaves it out"
#endif
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "../../include/tinyopt_amd.h"
#include "wide_kernels.hpp"

// ================================================================================================
// host side shared by the translation units
// ================================================================================================
struct toa_context {
  int device = 0;
  hipStream_t stream = nullptr;
  int num_cus = 0;
  int clock_khz = 0;
  int max_lds = 0;
  char name[128] = {0};
  int* queue = nullptr;  // device work-queue head
  bool queue_dirty = true;  // the queue block needs a memset before the next fused launch (first use, or after a failure)
  void* params_dev = nullptr;  // device copy of the fused kernel's parameter block
  int loss = TOA_LOSS_L2;      // toa_set_loss: the M-estimator of this handle's cost functor (DenseRow / Jet families)
  double loss_th2 = 0;
  toa_tuning tune = {};        // toa_set_tuning: A/B arms (all-zero = the library's choices)
  std::string timeline_path;   // toa_debug_timeline
  unsigned char params_shadow[1024] = {0};  // what params_dev holds (or will hold, in stream order): see upload_params
  size_t params_shadow_bytes = 0;
  void* scratch = nullptr;     // row-split path: state + partials + folded H (grown on demand)
  size_t scratch_bytes = 0;
  std::vector<std::unique_ptr<char[]>> captured_blocks;   // parameter blocks of launches captured into hipGraphs (upload_params)
  bool shadow_retired = false;
  // Workspaces a captured hipGraph may still point into.  Once ANY launch of this handle has been captured (shadow_retired),
  // a workspace that has to grow is not freed but parked here until toa_destroy: a graph bakes the raw device pointers of the
  // scratch / memo / aux blocks of its capture time into its nodes, and a later eager call with a larger shape must not pull
  // them from under a replay.  toa_release_workspace() is the only way a handle's workspace is given up.
  std::vector<void*> retired_blocks;
  void* memo = nullptr;        // fused kernel: one parked linearisation per resident wave (lm_device.hpp; grown on demand)
  size_t memo_bytes = 0;
  void* aux = nullptr;         // bundle adjustment with visibility lists: its work arrays (`scratch` belongs to the solver it calls)
  size_t aux_bytes = 0;
  // launch-per-stage pipelines (BA lists, n > 128): the ring through which the host reads "is anything still running" a few
  // passes late — pinned flags + one event per slot, created on first use and kept (hipHostMalloc costs ~1 ms per call)
  // set by a pipeline around its toa_large_solve call (bundle adjustment with lists): matrix p is factorised only where
  // solve_mask[p * solve_mask_stride] != 0 — the workgroups of finished scenes leave at once (own kernels only; the library
  // path solves everything, as before).  Device pointer; NULL = solve all.
  const int32_t* solve_mask = nullptr;
  int64_t solve_mask_stride = 0;
  static constexpr int kPassRing = 4, kLanes = 4;   // (lanes: the n > 128 pipeline runs the batch as up to four lanes on as many streams)
  int* pass_flags = nullptr;   // [kLanes][kPassRing][2], pinned host memory
  hipEvent_t pass_done[kLanes * kPassRing] = {};
  hipStream_t lane_stream[kLanes - 1] = {};
  hipEvent_t lane_fork = nullptr, lane_join[kLanes - 1] = {}, lane_gram[kLanes * kPassRing] = {};
  // row-split path: optional hipGraph of the (init, [partial, step] x iters) launch sequence (toa_tuning::wide_graph)
  struct WideGraph { const void* k_init; const void* k_part; const void* k_step; unsigned g_p, g_u; size_t lds; int iters; hipGraphExec_t exec; };
  WideGraph wgraphs[16];
  int nwgraphs = 0;
  // launch-configuration cache: (kernel, dynamic LDS bytes) -> resident workgroups per CU.
  // hipFuncSetAttribute / hipOccupancy* cost milliseconds per call; pay them once per variant.
  struct Cfg { const void* fn; size_t lds; int wg_per_cu; };
  Cfg cfg[256];
  int ncfg = 0;
  // large-n K3 (large_n.hip): rocBLAS handle created on the first n > 63 solve, and how to destroy it
  void* blas = nullptr;
  int (*blas_destroy)(void*) = nullptr;
  // one-matrix-per-call solves spread over side streams (toa_large_solve_each): streams, their rocBLAS handles, events
  static constexpr int kSide = 8;
  hipStream_t side_stream[kSide] = {};
  void* side_blas[kSide] = {};
  hipEvent_t side_done[kSide] = {};
  hipEvent_t side_fork = nullptr;
  int nside = 0;
};

// large_fused.hip: the n in [64, 128] loop as one persistent kernel (called by toa_large_lm_run when eligible)
bool toa_large_fused_eligible(toa_context* h, int dtype, int n, int m);
int toa_large_fused_lm_run(toa_context* h, int dtype, int n, int m, int64_t P, const void* data, void* x, const toa_options* options,
                           const toa_results* results, uint64_t* counters);

int toa_large_accumulate(toa_context* h, int dtype, int n, int m, int64_t P, const void* data, const void* x, int want_grad, void* g,
                         void* H, double* cost, int32_t* nres);
int toa_large_accumulate_pipeline(toa_context* h, int dtype, int n, int m, int64_t P, const void* data, const void* x, int want_grad, void* g,
                                  void* H, double* cost, int32_t* nres);

// error reporting lives in capi.hip (one thread_local message for the whole library)
int toa_fail(int code, const std::string& msg);
#define HIP_TRY(expr)                                                                           \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return toa_fail(e_ == hipErrorOutOfMemory ? TOA_E_NOMEM : TOA_E_HIP,                      \
                      std::string(#expr) + ": " + hipGetErrorString(e_));                       \
  } while (0)

// Give up a workspace block of the handle that is about to be replaced by a larger one: freed at once — unless a launch of
// this handle has ever been captured into a hipGraph, whose nodes may hold pointers into it (kept until toa_destroy then).
inline void toa_release_workspace(toa_context* h, void* block) {
  if (!block) return;
  if (h->shadow_retired) h->retired_blocks.push_back(block);
  else (void)hipFree(block);
}

// Before a device workspace is re-allocated: everything queued on the stream may still use the old block, so the stream is
// drained first — which, like the hipMalloc that follows, cannot happen while the stream is being CAPTURED into a hipGraph.
// Workspaces only ever grow and are kept, so one un-captured call of the same shape beforehand is all a capturing caller
// needs; without it the call is refused here instead of failing inside the runtime with the capture invalidated.
inline int grow_sync(toa_context* h, const char* what) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(h->stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
    return toa_fail(TOA_E_UNSUPPORTED, std::string(what) + ": a device workspace has to grow, which cannot happen while the stream is being captured; "
                                       "run this shape once before hipStreamBeginCapture (workspaces only grow and are kept by the handle)");
  HIP_TRY(hipStreamSynchronize(h->stream));
  return TOA_OK;
}

namespace toa {
// Every C entry point runs on its handle's GPU and leaves the CALLER's current device as it found it: torch (and any
// other HIP user of the process) reads its "current device" through hipGetDevice, so a library that switched it as a
// side effect would silently redirect the caller's later allocations in a single-process multi-GPU program.
struct DeviceGuard {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    int cur = -1;
    err = hipGetDevice(&cur);
    if (err == hipSuccess && cur != dev) {
      err = hipSetDevice(dev);
      if (err == hipSuccess) prev = cur;
    }
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define TOA_ON_DEVICE(dev)          \
  toa::DeviceGuard guard_(dev);     \
  HIP_TRY(guard_.err)

// Raise a kernel's dynamic-LDS limit once per (kernel, size): hipFuncSetAttribute costs ~1 ms per call.
inline int ensure_lds_attr(toa_handle h, const void* fn, size_t bytes) {
  size_t max_set = 0;
  for (int i = 0; i < h->ncfg; ++i)
    if (h->cfg[i].fn == fn) {
      if (h->cfg[i].lds == bytes) return TOA_OK;
      if (h->cfg[i].lds > max_set) max_set = h->cfg[i].lds;
    }
  // the limit only ever grows: a smaller request (another n on the same instantiation) must not lower it under a larger
  // size whose cache entry would make later launches skip this call
  if (bytes > max_set) HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  if (h->ncfg < 256) h->cfg[h->ncfg++] = {fn, bytes, -1};
  return TOA_OK;
}

// Stream-ordered upload of a kernel's parameter block into the context's device copy.  Repeated solves over the same
// buffers (an outer loop re-solving, the stepping form, the benchmark) present byte-identical blocks: the upload — a
// staged ~10 us stream operation in front of every launch — is skipped when the block already there is the same.
inline int upload_params(toa_handle h, const void* blk, size_t bytes) {
  // Under stream capture the copy below is only RECORDED, host POINTER included: the graph reads the block when it is
  // launched, long after the caller's stack copy is gone — so the block is parked in host memory the handle keeps for its
  // lifetime (1 KB per captured launch).  And once a graph of ours exists, a replay can rewrite the device block behind the
  // shadow's back at any time: from then on every eager call uploads (~10 us), the shadow is retired for this handle.
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(h->stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
    h->captured_blocks.emplace_back(new char[bytes]);
    std::memcpy(h->captured_blocks.back().get(), blk, bytes);
    HIP_TRY(hipMemcpyAsync(h->params_dev, h->captured_blocks.back().get(), bytes, hipMemcpyHostToDevice, h->stream));
    h->params_shadow_bytes = 0;
    h->shadow_retired = true;
    return TOA_OK;
  }
  if (h->shadow_retired) {
    HIP_TRY(hipMemcpyAsync(h->params_dev, blk, bytes, hipMemcpyHostToDevice, h->stream));
    return TOA_OK;
  }
  if (bytes == h->params_shadow_bytes && std::memcmp(h->params_shadow, blk, bytes) == 0) return TOA_OK;
  HIP_TRY(hipMemcpyAsync(h->params_dev, blk, bytes, hipMemcpyHostToDevice, h->stream));
  std::memcpy(h->params_shadow, blk, bytes);
  h->params_shadow_bytes = bytes;
  return TOA_OK;
}

inline int ensure_pass_ring(toa_handle h) {
  if (h->pass_flags) return TOA_OK;
  constexpr int kSlots = toa_context::kPassRing * toa_context::kLanes;
  for (int i = 0; i < kSlots; ++i) HIP_TRY(hipEventCreateWithFlags(&h->pass_done[i], hipEventDisableTiming));
  for (hipStream_t& s : h->lane_stream) HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  HIP_TRY(hipEventCreateWithFlags(&h->lane_fork, hipEventDisableTiming));
  for (hipEvent_t& e : h->lane_join) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (hipEvent_t& e : h->lane_gram) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->pass_flags), kSlots * 2 * sizeof(int), hipHostMallocDefault));
  return TOA_OK;
}

// waves per workgroup: 4 (256 threads) everywhere but the team form of the fused kernel; LDS per wave decides how many WGs fit per CU.
template <typename T>
inline int lds_fit(toa_handle h, int n, size_t* per_wave, size_t* per_wg, int waves = 4, size_t stage = 0) {
  size_t pw = WaveLds<T>::bytes(n);
  pw = (pw + 15) & ~size_t(15);
  pw += stage_carve_off<T>(stage, n);   // a model's LDS stage (row_model.hpp) overlays the carve's pass-dead head: what sticks out in front
  *per_wave = pw;
  *per_wg = pw * waves;
  if (*per_wg > 160 * 1024) return toa_fail(TOA_E_UNSUPPORTED, "LDS footprint exceeds 160 KiB per workgroup");
  (void)h;
  return TOA_OK;
}

// Chunks per pass of a cooperative model — a property of the SHAPE, never of the batch, the position in it, or of which
// form of the kernel runs (classic / team), so that a problem's bits depend on none of them: ~1024 rows per chunk
// (256 for the 64-row super-batch layouts); toa_tuning::coop_chunks overrides (experiments, and the team form's tests).
template <typename Model>
inline void coop_chunking(toa_handle h, int n, int m, int* K_out, int* cs_out) {
  (void)n;
  const int steps_total = (m + 3) / 4;
  constexpr bool super16 = Model::kCoopPeriod == 16;
  int K = super16 ? std::max(2, std::min(16, (m + 128) / 256)) : std::max(2, std::min(16, (m + 512) / 1024));
  if (h->tune.coop_chunks >= 2 && h->tune.coop_chunks <= 64) K = h->tune.coop_chunks;
  const int period = Model::kCoopPeriod;   // steps per ring turn / super-batch: chunk boundaries fall on it
  int cs = (steps_total + K - 1) / K;
  cs = (cs + period - 1) / period * period;
  *cs_out = cs;
  *K_out = (steps_total + cs - 1) / cs;
}
template <typename Model>
inline int launch_accumulate(toa_handle h, int n, int m, int64_t P, const void* data, const void* x, int want_grad,
                             void* g, void* H, double* cost, int32_t* nres) {
  using T = typename Model::Scalar;
  long long grid = (P + 3) / 4;
  const long long cap = (long long)h->num_cus * 8;
  if (grid > cap) grid = cap;
  size_t pw, pwg;
  if (int rc = lds_fit<T>(h, n, &pw, &pwg, 4, ModelStageBytes<Model>::value)) return rc;
  using RModel = typename RobustOf<Model>::type;
  if (h->loss != TOA_LOSS_L2 && !std::is_same<RModel, Model>::value) {
    if (int rc = ensure_lds_attr(h, (const void*)accumulate_kernel<RModel>, pwg)) return rc;
    hipLaunchKernelGGL((accumulate_kernel<RModel>), dim3((unsigned)grid), dim3(256), pwg, h->stream, data, x, (long long)P, n, m,
                       want_grad, g, H, cost, nres, (int)pw, h->loss, h->loss_th2);
  } else {
    if (int rc = ensure_lds_attr(h, (const void*)accumulate_kernel<Model>, pwg)) return rc;
    hipLaunchKernelGGL((accumulate_kernel<Model>), dim3((unsigned)grid), dim3(256), pwg, h->stream, data, x, (long long)P, n, m,
                       want_grad, g, H, cost, nres, (int)pw, h->loss, h->loss_th2);
  }
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}

template <typename Model>
inline int launch_fused(toa_handle h, const FusedParams& prm_in) {
  using T = typename Model::Scalar;
  constexpr int kW = ModelWaves<Model>::value;
  constexpr int NO = kW;   // (every wave of a workgroup pulls problems)
  FusedParams prm = prm_in;
  size_t pw, pwg;
  if (int rc = lds_fit<T>(h, prm.n, &pw, &pwg, kW)) return rc;
  prm.lds_per_wave = (int)pw;
  prm.queue = h->queue;
  // [0] pop counter, [16] waves that have left: zeroed when the handle is created and by the last wave of every launch
  // (lm_fused_kernel); a launch that failed may have left them dirty, so the next one starts from a memset again
  if (h->queue_dirty) {
    HIP_TRY(hipMemsetAsync(h->queue, 0, 48 * sizeof(int), h->stream));
    h->queue_dirty = false;
  }
  auto kern = lm_fused_kernel<Model>;
  // resident workgroups per CU for a dynamic-LDS size (cached: the two HIP calls cost milliseconds)
  auto occupancy = [&](size_t lds_bytes, int* out) -> int {
    int w = 0;
    size_t max_set = 0;
    for (int i = 0; i < h->ncfg; ++i)
      if (h->cfg[i].fn == (const void*)kern) {
        if (h->cfg[i].lds == lds_bytes && h->cfg[i].wg_per_cu > 0) w = h->cfg[i].wg_per_cu;
        if (h->cfg[i].lds > max_set) max_set = h->cfg[i].lds;
      }
    if (w == 0) {
      if (lds_bytes > max_set)   // the limit only ever grows: a smaller request must not lower it under a cached larger one
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
      HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&w, kern, 64 * kW, lds_bytes));
      if (w < 1) w = 1;
      if (h->ncfg < 256) h->cfg[h->ncfg++] = {(const void*)kern, lds_bytes, w};
    }
    *out = w;
    return TOA_OK;
  };
  prm.stage_off = 0;
  prm.carve_off = 0;
  if constexpr (ModelStageBytes<Model>::value > 0) {
    // The LDS stage of the row-per-lane pass: it begins at the wave's region and ends INSIDE the carve, over the part of it
    // that is dead while a pass runs (WaveLds::pass_dead_bytes: LDL^T workspace, the solve's scratch, the step).  C3 (fp64,
    // n = 12): 8 192 + 6 464 = 14 656 bytes per wave were two workgroups per compute unit (round 3 / 4 ran this kernel at
    // two waves per SIMD without noticing); overlaid 11 104, with the memo's 2 048 behind them 13 152 <= 160 KiB / 12: three.
    const size_t dead = std::min(WaveLds<T>::pass_dead_bytes(prm.n), size_t(ModelStageBytes<Model>::value));
    prm.carve_off = int(ModelStageBytes<Model>::value - dead);
    pw += prm.carve_off;
    pwg = pw * kW;
    prm.lds_per_wave = (int)pw;
  }
  int wg_per_cu = 0;
  if (int rc = occupancy(pwg, &wg_per_cu)) return rc;
  prm.memo = nullptr;
  prm.memo_stride = 0;
  prm.memo_lds_off = 0;
  bool memo_on = false;
  if constexpr (ModelMemo<Model>::value) {
    // One parked linearisation per resident wave (the Gram registers of the last accepted point: ~10 KB at n = 50, 2 KB at
    // n = 12 fp64): the re-accumulation that follows a rejected step reads it back instead of streaming the problem's rows
    // again.  toa_tuning::memo_off switches it off (A/B, and the test that the results do not depend on it).
    // (not with an M-estimator in the fused kernel — row models: the cost of a robust linearisation is the sum of the losses, which the
    //  parked Gram does not carry)
    memo_on = !h->tune.memo_off && prm.loss == TOA_LOSS_L2;
    if (memo_on) {
      // a small Gram is parked in LDS when that costs no resident workgroup (C3: parking in HBM after every accepted step
      // measured 1.5 % of the launch for a workload that never rejects a step)
      const size_t mb = (Model::kMemoBytes + 15) & ~size_t(15);
      if (mb <= 4096 && (pw + mb) * 4 <= 160 * 1024) {
        int w2 = 0;
        if (int rc = occupancy((pw + mb) * 4, &w2)) return rc;
        if (w2 == wg_per_cu) {
          prm.memo_lds_off = (int)pw;
          pw += mb;
          pwg = pw * 4;
          prm.lds_per_wave = (int)pw;
        }
      }
    }
  }
  prm.coop_K = 0;
  prm.coop_cs = 0;
  prm.coop_tot_off = 0;
  if constexpr (ModelCoop<Model>::value) {
    prm.coop_K = 1;                       // one chunk = the classic pass, bit for bit
    prm.coop_cs = (prm.m + 3) / 4;
    pwg += kCoopCtlBytes;                 // the control block (the waves' carves are far below the LDS limit of a smaller grid)
    // Cooperative passes (CoopCtl): on for a SHAPE (never for a batch size or a position in the batch, so that a problem's
    // bits do not depend on them), when a pass has enough rows to be worth sharing.  The chunk total of a pass is summed in
    // the owner's LDL^T workspace when the Gram registers fit it, in an area of its own otherwise (if that costs no
    // resident workgroup).  toa_tuning::coop_off switches it off (A/B).
    const bool coop_on = !h->tune.coop_off;
    constexpr bool super16 = Model::kCoopPeriod == 16;   // fp64 n <= 15: 64-row super-batches, 52 KB problems — share from 256 rows
    bool room = Model::kMemoBytes <= WaveLds<T>::m_elems(prm.n) * sizeof(T);
    if (coop_on && prm.m >= (super16 ? 256 : 1024) && !room) {
      const size_t mb = (Model::kMemoBytes + 15) & ~size_t(15);
      int w2 = 0;
      if ((pw + mb) * kW + kCoopCtlBytes <= 160 * 1024) {
        if (int rc = occupancy((pw + mb) * kW + kCoopCtlBytes, &w2)) return rc;
        if (w2 == wg_per_cu) {
          prm.coop_tot_off = (int)pw;
          pw += mb;
          pwg = pw * kW + kCoopCtlBytes;
          prm.lds_per_wave = (int)pw;
          room = true;
        }
      }
    }
    if (coop_on && prm.m >= (super16 ? 256 : 1024) && room) {
      // chunks per pass: ~1024 rows each (256 for the super-batch form).  Same box, C4 (m = 2000), three interleaved rounds
      // (profiles/r03_ab_log.md): K = 2: 12.83 M it/s, K = 3: 12.69, K = 4: 12.71, K = 8: 12.41, off: 12.52 — every chunk pays
      // its own ramp of the load ring, so the coarsest split that still lets a sibling help wins.  (toa_tuning::coop_chunks: experiments)
      coop_chunking<Model>(h, prm.n, prm.m, &prm.coop_K, &prm.coop_cs);
    }
  }
  long long grid = (long long)h->num_cus * wg_per_cu;
  const long long need = (prm.P + NO - 1) / NO;
  if (grid > need) grid = need;
  if (grid < 1) grid = 1;
  // (Sizing the grid to P / rounds waves so that every round is full was tried: at the BASELINE shard size 625 workgroups
  // instead of 768 are ~2 % slower, tools/grid_ab.sh — more resident waves hide more latency than full rounds save.)
  if (h->tune.max_workgroups > 0 && grid > h->tune.max_workgroups) grid = h->tune.max_workgroups;   // experiments only
  if constexpr (ModelMemo<Model>::value) {
    if (memo_on && prm.memo_lds_off == 0) {
      const size_t stride = (Model::kMemoBytes + 255) & ~size_t(255);
      const size_t need_b = stride * size_t(grid) * NO;
      if (need_b > h->memo_bytes) {
        if (int rc = grow_sync(h, "memo of the last accepted linearisation")) return rc;
        toa_release_workspace(h, h->memo);
        h->memo = nullptr;
        h->memo_bytes = 0;
        HIP_TRY(hipMalloc(&h->memo, need_b));
        h->memo_bytes = need_b;
      }
      prm.memo = h->memo;
      prm.memo_stride = stride;
    }
  }
  static_assert(sizeof(FusedParams) <= 1024, "parameter block too large");
  // stream-ordered upload of the parameter block (kept out of the kernarg segment so that its ~60
  // scalars are loaded on demand instead of being pinned in SGPRs across the hot loop)
  if (int rc = upload_params(h, &prm, sizeof(prm))) return rc;
  const char* tl_path = h->timeline_path.empty() ? nullptr : h->timeline_path.c_str();
  unsigned long long* tl_dev = nullptr;
  if (tl_path) {  // debug: per-problem start / end stamps of this launch, appended to the file as text
    HIP_TRY(hipMalloc(&tl_dev, size_t(prm.P) * 16));
    HIP_TRY(hipMemsetAsync(tl_dev, 0, size_t(prm.P) * 16, h->stream));
    prm.timeline = tl_dev;
    if (int rc = upload_params(h, &prm, sizeof(prm))) return rc;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * kW), pwg, h->stream, (const FusedParams*)h->params_dev);
  if (hipError_t e_ = hipGetLastError(); e_ != hipSuccess) {
    h->queue_dirty = true;
    return toa_fail(TOA_E_HIP, std::string("lm_fused_kernel launch: ") + hipGetErrorString(e_));
  }
  if (tl_path) {
    std::vector<unsigned long long> tl(size_t(prm.P) * 2);
    HIP_TRY(hipMemcpy(tl.data(), tl_dev, tl.size() * 8, hipMemcpyDeviceToHost));
    (void)hipFree(tl_dev);
    if (FILE* f = std::fopen(tl_path, "a")) {
      std::fprintf(f, "# launch P=%lld grid=%lld\n", prm.P, grid);
      for (long long q = 0; q < prm.P; ++q) std::fprintf(f, "%llu %llu\n", tl[2 * q], tl[2 * q + 1]);
      std::fclose(f);
    }
  }
  return TOA_OK;
}

// Stepping form (`lm::Optimizer<H_t> optimizer(options)`; `optimizer.Step(x, acc, out)`, optimizer.h:199,331-539) on the
// launch-per-iteration kernels above with ONE chunk per problem: begin = wide_init_kernel, a step = the data pass
// (wide_partial_kernel<Model>: H, g, cost of the current x into the caller's state block) + wide_step_kernel (one
// lm_iteration).  The H of the last build stays in the state block, which is what eval-only iterations keep solving
// with while x sits at a trial point (optimizer.h:281-299, lm.h:96-117).
// State block layout: [ WideState<T>[P] | partial (H, g, cost, inliers)[P] | folded H [P][n*n] ], each 256-byte aligned.
template <typename T>
inline size_t stepping_state_bytes(int n, long long P, size_t* o_part = nullptr, size_t* o_hsum = nullptr) {
  const size_t stride = size_t(n) * n + n + 2;
  const size_t b_state = (size_t(P) * sizeof(WideState<T>) + 255) & ~size_t(255);
  const size_t b_part = (size_t(P) * stride * sizeof(T) + 255) & ~size_t(255);
  const size_t b_hsum = (size_t(P) * n * n * sizeof(T) + 255) & ~size_t(255);
  if (o_part) *o_part = b_state;
  if (o_hsum) *o_hsum = b_state + b_part;
  return b_state + b_part + b_hsum;
}

template <typename Model, int NPAD, typename Manifold>
inline int launch_stepping(toa_handle h, const FusedParams& fp) {
  using T = typename Model::Scalar;
  const int n = fp.n, m = fp.m;
  const long long P = fp.P;
  size_t pw, pwg, o_part, o_hsum;
  if (int rc = lds_fit<T>(h, n, &pw, &pwg, 4, ModelStageBytes<Model>::value)) return rc;
  (void)stepping_state_bytes<T>(n, P, &o_part, &o_hsum);
  WideParams wp;
  std::memset(&wp, 0, sizeof(wp));
  wp.data = fp.data; wp.x = fp.x; wp.P = P; wp.n = n; wp.m = m;
  wp.splits = 1;
  wp.chunk_rows = (((m + 3) & ~3) + 15) & ~15;
  wp.opt = fp.opt; wp.res = fp.res; wp.counters = fp.counters;
  wp.state = fp.state;
  wp.partials = static_cast<char*>(fp.state) + o_part;
  wp.hsum = static_cast<char*>(fp.state) + o_hsum;
  wp.step_mode = 1;
  wp.active = fp.active;
  wp.stop_request = fp.stop_request;
  wp.loss = fp.loss; wp.loss_th2 = fp.loss_th2;
  wp.lds_per_wave = int(pw);
  if (int rc = upload_params(h, &wp, sizeof(wp))) return rc;
  const WideParams* dp = static_cast<const WideParams*>(h->params_dev);
  const unsigned g_p = unsigned((P + 3) / 4);
  if (fp.mode == 1) {
    auto k_init = wide_init_kernel<T, Manifold::kXdim>;
    if (int rc = ensure_lds_attr(h, (const void*)k_init, pwg)) return rc;
    hipLaunchKernelGGL(k_init, dim3(g_p), dim3(256), pwg, h->stream, dp);
  } else if (fp.mode == 3) {
    auto k_stop = wide_stop_kernel<T, NPAD, Manifold>;
    if (int rc = ensure_lds_attr(h, (const void*)k_stop, pwg)) return rc;
    hipLaunchKernelGGL(k_stop, dim3(g_p), dim3(256), pwg, h->stream, dp);
  } else {
    using RModel = typename RobustOf<Model>::type;
    const bool robust = fp.loss != TOA_LOSS_L2 && !std::is_same<RModel, Model>::value;
    void (*k_part)(const WideParams*) = robust ? wide_partial_kernel<RModel> : wide_partial_kernel<Model>;
    auto k_step = wide_step_kernel<T, NPAD, Manifold>;
    if (int rc = ensure_lds_attr(h, (const void*)k_part, pwg)) return rc;
    if (int rc = ensure_lds_attr(h, (const void*)k_step, pwg)) return rc;
    hipLaunchKernelGGL(k_part, dim3(g_p), dim3(256), pwg, h->stream, dp);
    hipLaunchKernelGGL(k_step, dim3(g_p), dim3(256), pwg, h->stream, dp);
  }
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}

// Row-split driver.  Model = the chunk-capable residual model, NPAD / Manifold as for the step kernel.
template <typename Model, int NPAD, typename Manifold>
inline int launch_wide(toa_handle h, const FusedParams& fp, int splits_req) {
  using T = typename Model::Scalar;
  if (fp.mode != 0) return launch_stepping<Model, NPAD, Manifold>(h, fp);
  const int n = fp.n, m = fp.m;
  const long long P = fp.P;
  const int m4 = (m + 3) & ~3;
  // chunking (automatic): short chunks keep the per-iteration latency down — one wave streams its chunk at HBM
  // round-trip pace (~2 us per 16-row batch) — while the fold of the S partials costs ~3 us per 64 chunks for n <= 6
  // and S * n^2 / 64 serial loads per lane beyond; and P * S <= #CUs keeps the one-launch persistent form available.
  long long S = splits_req > 0 ? splits_req : m4 / (n <= 6 ? 32 : 256);
  if (splits_req <= 0) {
    if (S > 64) S = 64;
    const long long cap = (long long)h->num_cus / (P > 0 ? P : 1);
    if (cap >= 1 && S > cap) S = cap;
  }
  if (S > m4 / 16) S = m4 / 16;
  if (S < 1) S = 1;
  int chunk = int((m4 + S - 1) / S);
  chunk = (chunk + 15) & ~15;
  S = (m4 + chunk - 1) / chunk;
  size_t pw, pwg;
  if (int rc = lds_fit<T>(h, n, &pw, &pwg, 4, ModelStageBytes<Model>::value)) return rc;
  const size_t stride = size_t(n) * n + n + 2;
  const size_t b_state = (size_t(P) * sizeof(WideState<T>) + 255) & ~size_t(255);
  const size_t b_part = (size_t(P) * S * stride * sizeof(T) + 255) & ~size_t(255);
  const size_t b_hsum = (size_t(P) * n * n * sizeof(T) + 255) & ~size_t(255);
  const size_t b_sync = (size_t(2 * P + 1) * sizeof(unsigned) + 255) & ~size_t(255);
  const size_t need = b_state + b_part + b_hsum + b_sync;
  if (need > h->scratch_bytes) {
    if (int rc = grow_sync(h, "device workspace")) return rc;
    toa_release_workspace(h, h->scratch);
    h->scratch = nullptr;
    h->scratch_bytes = 0;
    HIP_TRY(hipMalloc(&h->scratch, need));
    h->scratch_bytes = need;
  }
  WideParams wp;
  std::memset(&wp, 0, sizeof(wp));
  wp.data = fp.data; wp.x = fp.x; wp.P = P; wp.n = n; wp.m = m;
  wp.splits = int(S); wp.chunk_rows = chunk;
  wp.opt = fp.opt; wp.res = fp.res; wp.counters = fp.counters;
  wp.state = h->scratch;
  wp.partials = static_cast<char*>(h->scratch) + b_state;
  wp.hsum = static_cast<char*>(h->scratch) + b_state + b_part;
  wp.sync = reinterpret_cast<unsigned*>(static_cast<char*>(h->scratch) + b_state + b_part + b_hsum);
  wp.lds_per_wave = int(pw);
  wp.loss = fp.loss; wp.loss_th2 = fp.loss_th2;
  static_assert(sizeof(WideParams) <= 1024, "parameter block too large");
  if (int rc = upload_params(h, &wp, sizeof(wp))) return rc;
  const WideParams* dp = static_cast<const WideParams*>(h->params_dev);
  // With an M-estimator on the handle (toa_set_loss) the data pass is the ROBUST variant of the model, which exists in the
  // pass-only kernels: the solve then runs in the launch-per-iteration form (no team / persistent kernel).
  using RModel = typename RobustOf<Model>::type;
  const bool robust = fp.loss != TOA_LOSS_L2 && !std::is_same<RModel, Model>::value;
  auto k_init = wide_init_kernel<T, Manifold::kXdim>;
  void (*k_part)(const WideParams*) = robust ? wide_partial_kernel<RModel> : wide_partial_kernel<Model>;
  auto k_step = wide_step_kernel<T, NPAD, Manifold>;
  if (int rc = ensure_lds_attr(h, (const void*)k_init, pwg)) return rc;
  if (int rc = ensure_lds_attr(h, (const void*)k_part, pwg)) return rc;
  if (int rc = ensure_lds_attr(h, (const void*)k_step, pwg)) return rc;
  const unsigned g_p = unsigned((P + 3) / 4), g_u = unsigned((P * S + 3) / 4);
  const int iters = fp.opt.max_iters + 1 + (fp.opt.check_final_cost ? 1 : 0);  // optimizer.h:248-250
  // Direct launches by default: an A/B on MI355X (tests/tools/latency_probe.py) shows graph replay and eager launches
  // of this 23..103-kernel sequence within 1 % of each other (C2 71 us, C5 93-99 us device time per solve), as
  // MI355X_MICROARCH.md's "boundary" row predicts (eager == hipGraph).  toa_tuning::wide_graph selects the graph path.
  // Persistent form (one launch for the whole solve) whenever every workgroup is certainly co-resident: one
  // 64-thread workgroup per chunk, at most one per CU.  toa_tuning::wide_multilaunch forces the launch-per-iteration form.
  // Instantiated for the small systems only (n <= 15: BASELINE configs C2 / C5 are n = 6): the kernel carries a whole
  // lm_iteration with the NPAD-unrolled register LDL^T per residual-model layout, and 40 copies of it tripled the build.
  const bool multilaunch_env = h->tune.wide_multilaunch != 0;
  const bool multilaunch = multilaunch_env || robust;
  const bool noteam = h->tune.wide_no_team != 0;
  if constexpr (NPAD <= 16) {
    // Team form: a small problem (<= 4096 rows) is cheaper on ONE compute unit with barrier hand-overs than on 16-64
    // of them with HBM hand-overs.  Up to 8 waves (512 threads: two waves per SIMD keep the whole register file usable).
    if (!multilaunch && !noteam && splits_req <= 0 && m4 <= 4096 && m4 >= 32) {
      long long St = std::min<long long>(8, m4 / 16);
      int chunk_t = int((m4 + St - 1) / St);
      chunk_t = (chunk_t + 15) & ~15;
      St = (m4 + chunk_t - 1) / chunk_t;
      const size_t lds_team = size_t(St) * pw + (size_t(St) * stride + 64) * sizeof(T) + 16;
      if (lds_team <= size_t(h->max_lds)) {
        wp.splits = int(St);
        wp.chunk_rows = chunk_t;
        if (int rc = upload_params(h, &wp, sizeof(wp))) return rc;
        auto k_team = wide_team_kernel<Model, NPAD, Manifold>;
        if (int rc = ensure_lds_attr(h, (const void*)k_team, lds_team)) return rc;
        hipLaunchKernelGGL(k_team, dim3(unsigned(P)), dim3(unsigned(64 * St)), lds_team, h->stream, dp);
        HIP_TRY(hipGetLastError());
        return TOA_OK;
      }
    }
    if (!multilaunch && P * S <= (long long)h->num_cus && pw <= 64 * 1024) {
      auto k_pers = wide_persistent_kernel<Model, NPAD, Manifold>;
      if (int rc = ensure_lds_attr(h, (const void*)k_pers, pw)) return rc;
      HIP_TRY(hipMemsetAsync(wp.sync, 0, size_t(2 * P + 1) * sizeof(unsigned), h->stream));
      hipLaunchKernelGGL(k_pers, dim3(unsigned(P * S)), dim3(64), pw, h->stream, dp);
      HIP_TRY(hipGetLastError());
      return TOA_OK;
    }
  }
  const bool use_graph = h->tune.wide_graph != 0;
  if (!use_graph) {
    hipLaunchKernelGGL(k_init, dim3(g_p), dim3(256), pwg, h->stream, dp);
    for (int it = 0; it < iters; ++it) {
      hipLaunchKernelGGL(k_part, dim3(g_u), dim3(256), pwg, h->stream, dp);
      hipLaunchKernelGGL(k_step, dim3(g_p), dim3(256), pwg, h->stream, dp);
    }
    HIP_TRY(hipGetLastError());
    return TOA_OK;
  }
  hipGraphExec_t exec = nullptr;
  for (int i = 0; i < h->nwgraphs; ++i) {
    const auto& w = h->wgraphs[i];
    if (w.k_init == (const void*)k_init && w.k_part == (const void*)k_part && w.k_step == (const void*)k_step && w.g_p == g_p &&
        w.g_u == g_u && w.lds == pwg && w.iters == iters)
      exec = w.exec;
  }
  if (!exec) {
    hipGraph_t graph;
    HIP_TRY(hipGraphCreate(&graph, 0));
    void* args[1] = {(void*)&dp};
    hipGraphNode_t prev = nullptr;
    auto add = [&](const void* fn, unsigned grid) -> hipError_t {
      hipKernelNodeParams kp;
      std::memset(&kp, 0, sizeof(kp));
      kp.func = const_cast<void*>(fn);
      kp.gridDim = dim3(grid);
      kp.blockDim = dim3(256);
      kp.sharedMemBytes = (unsigned)pwg;
      kp.kernelParams = args;
      hipGraphNode_t node;
      const hipError_t e = hipGraphAddKernelNode(&node, graph, prev ? &prev : nullptr, prev ? 1 : 0, &kp);
      prev = node;
      return e;
    };
    HIP_TRY(add((const void*)k_init, g_p));
    for (int it = 0; it < iters; ++it) {
      HIP_TRY(add((const void*)k_part, g_u));
      HIP_TRY(add((const void*)k_step, g_p));
    }
    HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    HIP_TRY(hipGraphDestroy(graph));
    if (h->nwgraphs < 16) h->wgraphs[h->nwgraphs++] = {(const void*)k_init, (const void*)k_part, (const void*)k_step, g_p, g_u, pwg, iters, exec};
  }
  // the parameter block was uploaded above with hipMemcpyAsync from pageable memory (staged before returning), so
  // back-to-back calls cannot race on it; the graph itself holds kernels only
  HIP_TRY(hipGraphLaunch(exec, h->stream));
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}

template <typename T, int NPAD>
inline int launch_solve(toa_handle h, int n, int64_t P, const void* H, const void* g, double scale, void* dx, int32_t* ok) {
  long long grid = (P + 3) / 4;
  const long long cap = (long long)h->num_cus * 8;
  if (grid > cap) grid = cap;
  size_t pw, pwg;
  if (int rc = lds_fit<T>(h, n, &pw, &pwg)) return rc;
  if (int rc = ensure_lds_attr(h, (const void*)solve_damped_kernel<T, NPAD>, pwg)) return rc;
  hipLaunchKernelGGL((solve_damped_kernel<T, NPAD>), dim3((unsigned)grid), dim3(256), pwg, h->stream, H, g, (long long)P, n,
                     scale, dx, ok, (int)pw);
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}
template <typename T, int NPAD>
inline int launch_inv_cov(toa_handle h, int n, int64_t P, const void* H, void* C, int32_t* ok) {
  long long grid = (P + 3) / 4;
  const long long cap = (long long)h->num_cus * 8;
  if (grid > cap) grid = cap;
  size_t pw, pwg;
  if (int rc = lds_fit<T>(h, n, &pw, &pwg)) return rc;
  if (int rc = ensure_lds_attr(h, (const void*)inv_cov_kernel<T, NPAD>, pwg)) return rc;
  hipLaunchKernelGGL((inv_cov_kernel<T, NPAD>), dim3((unsigned)grid), dim3(256), pwg, h->stream, H, (long long)P, n, C, ok, (int)pw);
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}
}  // namespace toa

// ---- per-(dtype, NBM) entry points defined in inst.hip (dtag: 0 = f32, 1 = f64) ----
int toa_inst_fused(int dtag, int nbm, int thin, toa_handle h, const toa::FusedParams& prm);
int toa_inst_accumulate(int dtag, int nbm, int thin, toa_handle h, int n, int m, int64_t P, const void* data,
                        const void* x, int want_grad, void* g, void* H, double* cost, int32_t* nres);
// Gaussian-prior / sqrt2 models (inst.hip -DTOA_INST_MISC)
int toa_inst_misc_fused(int dtag, int model, int npad, toa_handle h, const toa::FusedParams& prm);
int toa_inst_misc_accumulate(int dtag, int model, int npad, toa_handle h, int n, int m, int64_t P, const void* data,
                             const void* x, int want_grad, void* g, void* H, double* cost, int32_t* nres);
int toa_inst_wide(int dtag, int model, int nbm, int thin, toa_handle h, const toa::FusedParams& prm, int splits);
int toa_inst_inv_cov(int dtag, int npad, toa_handle h, int n, int64_t P, const void* H, void* C, int32_t* ok);
int toa_inst_solve(int dtag, int npad, toa_handle h, int n, int64_t P, const void* H, const void* g, double scale,
                   void* dx, int32_t* ok);
