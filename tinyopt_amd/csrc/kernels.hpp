// Kernels of the LM hot path + their host launchers (templates).  Included by capi.hip (C-ABI,
// dispatch) and by inst.hip, which is compiled once per (dtype, block count) so that the ~40
// instantiations of the fused kernel build in parallel (see __graft_entry__.build()).
//
// Kernel inventory:
//   lm_fused_kernel        K1+K2+K3+K4: whole LM solves, one wavefront per problem at a time,
//                          dynamic problem queue, no inter-wave or host synchronisation.
//   accumulate_kernel      K1/K2 seam: the Accumulate callback for a batch (g, H, cost out).
//   solve_damped_kernel    K3 seam: damping + LDL^T solve for a batch.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#include "../../include/tinyopt_amd.h"
#include "dense_row.hpp"
#include "ldlt_lds.hpp"
#include "ldlt_regs.hpp"
#include "lm_device.hpp"
#include "wave_utils.hpp"

namespace toa {

// ------------------------------------------------------------------------------------------------
// DenseRow model adaptor for the LM state machine.
// ------------------------------------------------------------------------------------------------
template <typename T, int NBM, int THIN>
struct DenseRowModel {
  static constexpr int kNpad = 16 * (NBM + (THIN > 0 ? 1 : 0));  // n <= kNpad - 1 ... see DenseRowLayout
  DenseRowGram<T, NBM, THIN> gram;
  const T* prob;
  DenseRowLayout lay;
  int m;
  __device__ __forceinline__ void accumulate(WaveLds<T>& L, int n, int lane, T& cost, int& nres) {
    gram.template pass<true>(prob, lay, n, L.xs, lane);
    cost = gram.extract_g_diag_cost(L.g, L.hd, lay, n, lane, L.tmp);
    nres = m;
  }
  __device__ __forceinline__ void evaluate(WaveLds<T>& L, int n, int lane, T& cost, int& nres) {
    cost = gram.template pass<false>(prob, lay, n, L.xs, lane);
    nres = m;
  }
  template <typename O>
  __device__ __forceinline__ void write_sym(O* M, int LD, int n, int lane) const {
    gram.write_sym(M, LD, lay, n, lane);
  }
};

struct FusedParams {
  const void* data;
  void* x;
  long long P;
  int n, m;
  toa_options opt;
  toa_results res;
  unsigned long long* counters;  // [4] or null
  int* queue;                    // work-queue head
  int lds_per_wave;
};

// Minimum resident waves per SIMD the register allocator must honour (2nd __launch_bounds__ argument
// is waves per SIMD on CDNA).  The fused kernel alternates an MFMA-paced accumulate phase with a
// latency-bound LDL^T phase, so >= 3 co-resident waves per SIMD are needed to keep the matrix pipe
// and the HBM queue busy; wide fp64 Gram tiles (NB >= 3: 48-80 accumulator registers) cannot afford it.
template <typename T, int NB>
constexpr int fused_min_waves() {
  return sizeof(T) == 4 ? (NB <= 2 ? 4 : 3) : (NB == 1 ? 4 : (NB == 2 ? 2 : 1));
}

template <typename T, int NBM, int THIN>
__global__ void __launch_bounds__(256) lm_fused_kernel(const FusedParams* __restrict__ prm_g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int n = prm_g->n;
  WaveLds<T> L = WaveLds<T>::carve(smem + size_t(wave) * prm_g->lds_per_wave, n);
  // private per-wave copies of the option / result PODs (no inter-wave synchronisation anywhere)
  {
    const int* src_o = reinterpret_cast<const int*>(&prm_g->opt);
    int* dst_o = reinterpret_cast<int*>(L.opt);
    for (int i = lane; i < int(sizeof(toa_options) / 4); i += 64) dst_o[i] = src_o[i];
    const int* src_r = reinterpret_cast<const int*>(&prm_g->res);
    int* dst_r = reinterpret_cast<int*>(L.res);
    for (int i = lane; i < int(sizeof(toa_results) / 4); i += 64) dst_r[i] = src_r[i];
    L.st->acc_passes = 0; L.st->eval_passes = 0; L.st->solves = 0; L.st->problems = 0;
  }
  wave_sync();
  const int m = prm_g->m;
  const long long P = prm_g->P;
  const DenseRowLayout lay = DenseRowLayout::make(n, m);
  DenseRowModel<T, NBM, THIN> model;
  model.m = m;
  model.lay = lay;
  const T* data = static_cast<const T*>(prm_g->data);
  T* X = static_cast<T*>(prm_g->x);
  int* queue = prm_g->queue;
  for (;;) {
    int p = 0;
    if (lane == 0) p = atomicAdd(queue, 1);
    p = __builtin_amdgcn_readfirstlane(p);
    if (p >= P) break;
    model.prob = data + size_t(p) * lay.elems_per_problem();
    wave_sync();
    L.xs[lane] = lane < n ? X[size_t(p) * n + lane] : T(0);
    wave_sync();
    lm_solve_problem<T>(model, L, n, lane, (long long)p);
    if (lane < n) X[size_t(p) * n + lane] = L.xs[lane];
  }
  unsigned long long* counters = prm_g->counters;
  if (counters && lane == 0) {
    atomicAdd(&counters[0], L.st->acc_passes);
    atomicAdd(&counters[1], L.st->eval_passes);
    atomicAdd(&counters[2], L.st->solves);
    atomicAdd(&counters[3], L.st->problems);
  }
}

// K1/K2 seam: one wave per problem (grid-stride), writes g [P][n], H [P][n*n], cost, nres.
template <typename T, int NBM, int THIN>
#ifndef TOA_ACC_WAVES
#define TOA_ACC_WAVES 1
#endif
__global__ void __launch_bounds__(256, TOA_ACC_WAVES) accumulate_kernel(const void* data_, const void* x_, long long P, int n, int m,
                                                         int want_grad, void* g_, void* H_, double* cost, int* nres) {
  __shared__ T xs_all[4][64];
  __shared__ T tmp_all[4][64 * 2 + 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T* xs = xs_all[wave];
  T* gl = tmp_all[wave];
  T* hd = gl + 64;
  T* slot = hd + 64;
  const DenseRowLayout lay = DenseRowLayout::make(n, m);
  const T* data = static_cast<const T*>(data_);
  const T* X = static_cast<const T*>(x_);
  DenseRowGram<T, NBM, THIN> gram;
  for (long long p = (long long)blockIdx.x * 4 + wave; p < P; p += (long long)gridDim.x * 4) {
    wave_sync();
    xs[lane] = lane < n ? X[size_t(p) * n + lane] : T(0);
    wave_sync();
    const T* prob = data + size_t(p) * lay.elems_per_problem();
    if (want_grad) {
      gram.template pass<true>(prob, lay, n, xs, lane);
      const T c = gram.extract_g_diag_cost(gl, hd, lay, n, lane, slot);
      T* G = static_cast<T*>(g_) + size_t(p) * n;
      T* H = static_cast<T*>(H_) + size_t(p) * n * n;
      if (lane < n) G[lane] = gl[lane];
      gram.write_sym(H, n, lay, n, lane);
      wave_sync();
      if (lane < n) H[lane * n + lane] = hd[lane];  // thin-tail diagonal entries come from hd
      if (lane == 0) { cost[p] = double(c); if (nres) nres[p] = m; }
    } else {
      const T c = gram.template pass<false>(prob, lay, n, xs, lane);
      if (lane == 0) { cost[p] = double(c); if (nres) nres[p] = m; }
    }
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
    for (int e = lane; e < n * n; e += 64) {
      const int i = e / n, j = e % n;
      const int a = i < j ? i : j, b = i < j ? j : i;
      T v = H[size_t(b) * n + a];
      if (i == j) v = T(double(v) * scale);
      L.M[i * L.LD + j] = v;
    }
    wave_sync();
    const T gl = lane < n ? gg[size_t(p) * n + lane] : T(0);
    bool ok;
    T dx = 0;
    {
      LdltRegs<T, NPAD> F;
      F.load(L.M, L.LD, n, lane);
      ok = F.factor(n, lane);
      if (ok) dx = F.solve(n, lane, -gl);
    }
    if (!ok) {
      ok = ldlt_factor_wave<T>(L.M, L.LD, L.perm, L.tmp, n, lane);
      if (ok) dx = ldlt_solve_wave<T>(L.M, L.LD, L.perm, L.vec, n, lane, -gl);
    }
    if (lane < n) dxg[size_t(p) * n + lane] = dx;
    if (lane == 0) ok_[p] = ok ? 1 : 0;
  }
}

}  // namespace toa

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
  void* params_dev = nullptr;  // device copy of the fused kernel's parameter block
  // launch-configuration cache: (kernel, dynamic LDS bytes) -> resident workgroups per CU.
  // hipFuncSetAttribute / hipOccupancy* cost milliseconds per call; pay them once per variant.
  struct Cfg { const void* fn; size_t lds; int wg_per_cu; };
  Cfg cfg[32];
  int ncfg = 0;
};

// error reporting lives in capi.hip (one thread_local message for the whole library)
int toa_fail(int code, const std::string& msg);
#define HIP_TRY(expr)                                                                           \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return toa_fail(e_ == hipErrorOutOfMemory ? TOA_E_NOMEM : TOA_E_HIP,                      \
                      std::string(#expr) + ": " + hipGetErrorString(e_));                       \
  } while (0)

namespace toa {
template <typename T, int NBM, int THIN>
inline int launch_accumulate(toa_handle h, int n, int m, int64_t P, const void* data, const void* x, int want_grad,
                             void* g, void* H, double* cost, int32_t* nres) {
  long long grid = (P + 3) / 4;
  const long long cap = (long long)h->num_cus * 8;
  if (grid > cap) grid = cap;
  hipLaunchKernelGGL((accumulate_kernel<T, NBM, THIN>), dim3((unsigned)grid), dim3(256), 0, h->stream, data, x, (long long)P, n, m,
                     want_grad, g, H, cost, nres);
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}

// waves per workgroup is fixed at 4 (256 threads); LDS per wave decides how many WGs fit per CU.
template <typename T>
inline int lds_fit(toa_handle h, int n, size_t* per_wave, size_t* per_wg) {
  size_t pw = WaveLds<T>::bytes(n);
  pw = (pw + 15) & ~size_t(15);
  *per_wave = pw;
  *per_wg = pw * 4;
  if (*per_wg > 160 * 1024) return toa_fail(TOA_E_UNSUPPORTED, "LDS footprint exceeds 160 KiB per workgroup");
  (void)h;
  return TOA_OK;
}

template <typename T, int NBM, int THIN>
inline int launch_fused(toa_handle h, const FusedParams& prm_in) {
  FusedParams prm = prm_in;
  size_t pw, pwg;
  if (int rc = lds_fit<T>(h, prm.n, &pw, &pwg)) return rc;
  prm.lds_per_wave = (int)pw;
  prm.queue = h->queue;
  HIP_TRY(hipMemsetAsync(h->queue, 0, sizeof(int), h->stream));
  auto kern = lm_fused_kernel<T, NBM, THIN>;
  int wg_per_cu = 0;
  for (int i = 0; i < h->ncfg; ++i)
    if (h->cfg[i].fn == (const void*)kern && h->cfg[i].lds == pwg) wg_per_cu = h->cfg[i].wg_per_cu;
  if (wg_per_cu == 0) {
    HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pwg));
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&wg_per_cu, kern, 256, pwg));
    if (wg_per_cu < 1) wg_per_cu = 1;
    if (h->ncfg < 32) h->cfg[h->ncfg++] = {(const void*)kern, pwg, wg_per_cu};
  }
  long long grid = (long long)h->num_cus * wg_per_cu;
  const long long need = (prm.P + 3) / 4;
  if (grid > need) grid = need;
  if (grid < 1) grid = 1;
  static_assert(sizeof(FusedParams) <= 1024, "parameter block too large");
  // stream-ordered upload of the parameter block (kept out of the kernarg segment so that its ~60
  // scalars are loaded on demand instead of being pinned in SGPRs across the hot loop)
  HIP_TRY(hipMemcpyAsync(h->params_dev, &prm, sizeof(prm), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), pwg, h->stream, (const FusedParams*)h->params_dev);
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
  HIP_TRY(hipFuncSetAttribute((const void*)solve_damped_kernel<T, NPAD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pwg));
  hipLaunchKernelGGL((solve_damped_kernel<T, NPAD>), dim3((unsigned)grid), dim3(256), pwg, h->stream, H, g, (long long)P, n,
                     scale, dx, ok, (int)pw);
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}
}  // namespace toa

// ---- per-(dtype, NBM) entry points defined in inst.hip (dtag: 0 = f32, 1 = f64) ----
int toa_inst_fused(int dtag, int nbm, int thin, toa_handle h, const toa::FusedParams& prm);
int toa_inst_accumulate(int dtag, int nbm, int thin, toa_handle h, int n, int m, int64_t P, const void* data,
                        const void* x, int want_grad, void* g, void* H, double* cost, int32_t* nres);
int toa_inst_solve(int dtag, int npad, toa_handle h, int n, int64_t P, const void* H, const void* g, double scale,
                   void* dx, int32_t* ok);
