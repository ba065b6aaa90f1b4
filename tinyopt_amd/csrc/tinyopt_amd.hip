// libtinyopt_amd.so — kernels + the C-ABI of include/tinyopt_amd.h.  gfx950 (MI355X, CDNA4) only.
//
// Kernel inventory:
//   lm_fused_kernel        K1+K2+K3+K4: whole LM solves, one wavefront per problem at a time,
//                          dynamic problem queue, no inter-wave or host synchronisation.
//   accumulate_kernel      K1/K2 seam: the Accumulate callback for a batch (g, H, cost out).
//   solve_damped_kernel    K3 seam: damping + LDL^T solve for a batch.
//   dense_row_pack_kernel / dense_row_synth_kernel   data-format callers either side of the path.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#include "../../include/tinyopt_amd.h"
#include "dense_row.hpp"
#include "ldlt_lds.hpp"
#include "lm_device.hpp"
#include "wave_utils.hpp"

namespace toa {

// ------------------------------------------------------------------------------------------------
// DenseRow model adaptor for the LM state machine.
// ------------------------------------------------------------------------------------------------
template <typename T, int NB>
struct DenseRowModel {
  DenseRowGram<T, NB> gram;
  const T* prob;
  int m, m4, RS;
  __device__ __forceinline__ void accumulate(WaveLds<T>& L, int n, int lane, T& cost, int& nres) {
    gram.template pass<true>(prob, m4, RS, n, L.xs, lane);
    cost = gram.extract_g_diag_cost(L.g, L.hd, n, lane, L.tmp);
    nres = m;
  }
  __device__ __forceinline__ void evaluate(WaveLds<T>& L, int n, int lane, T& cost, int& nres) {
    cost = gram.template pass<false>(prob, m4, RS, n, L.xs, lane);
    nres = m;
  }
  template <typename O>
  __device__ __forceinline__ void write_sym(O* M, int LD, int n, int lane) const {
    gram.write_sym(M, LD, n, lane);
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

template <typename T, int NB>
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
  DenseRowModel<T, NB> model;
  model.m = m;
  model.m4 = lay.m4;
  model.RS = lay.rs;
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
template <typename T, int NB>
__global__ void __launch_bounds__(256) accumulate_kernel(const void* data_, const void* x_, long long P, int n, int m,
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
  DenseRowGram<T, NB> gram;
  for (long long p = (long long)blockIdx.x * 4 + wave; p < P; p += (long long)gridDim.x * 4) {
    wave_sync();
    xs[lane] = lane < n ? X[size_t(p) * n + lane] : T(0);
    wave_sync();
    const T* prob = data + size_t(p) * lay.elems_per_problem();
    if (want_grad) {
      gram.template pass<true>(prob, lay.m4, lay.rs, n, xs, lane);
      const T c = gram.extract_g_diag_cost(gl, hd, n, lane, slot);
      T* G = static_cast<T*>(g_) + size_t(p) * n;
      T* H = static_cast<T*>(H_) + size_t(p) * n * n;
      if (lane < n) G[lane] = gl[lane];
      gram.write_sym(H, n, n, lane);
      if (lane == 0) { cost[p] = double(c); if (nres) nres[p] = m; }
    } else {
      const T c = gram.template pass<false>(prob, lay.m4, lay.rs, n, xs, lane);
      if (lane == 0) { cost[p] = double(c); if (nres) nres[p] = m; }
    }
  }
}

// K3 seam: H_ii *= scale (double), dx = -H^-1 g with Eigen's acceptance rule.
template <typename T>
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
    const T gl = lane < n ? gg[size_t(p) * n + lane] : T(0);
    const bool ok = ldlt_factor_wave<T>(L.M, L.LD, L.perm, L.tmp, n, lane);
    T dx = 0;
    if (ok) dx = ldlt_solve_wave<T>(L.M, L.LD, L.perm, L.vec, n, lane, -gl);
    if (lane < n) dxg[size_t(p) * n + lane] = dx;
    if (lane == 0) ok_[p] = ok ? 1 : 0;
  }
}

// Natural (A [P][m][n], b [P][m]) -> packed [P][m4][RS].
template <typename T>
__global__ void dense_row_pack_kernel(const T* __restrict__ A, const T* __restrict__ b, T* __restrict__ out,
                                      long long P, int n, int m, int m4, int RS) {
  const long long total = P * (long long)m4 * RS;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int q = int(e % RS);
    const long long rowg = e / RS;
    const int i = int(rowg % m4);
    const long long p = rowg / m4;
    T v = 0;
    if (i < m) {
      if (q < n) v = A[(p * m + i) * n + q];
      else if (q == n) v = b[p * m + i];
    }
    out[e] = v;
  }
}

// ---- synthetic inputs (same recipe as oracle/synth.hpp) ----
__host__ __device__ inline unsigned long long sm64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__host__ __device__ inline unsigned long long skey(unsigned long long seed, unsigned long long p, unsigned long long s) {
  return sm64(sm64(seed + p) ^ (s * 0xD6E8FEB86659FD93ull));
}
__host__ __device__ inline double u11(unsigned long long k, unsigned long long idx) {
  return double(sm64(k + idx) >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
}

template <typename T>
__global__ void dense_row_synth_kernel(T* __restrict__ out, T* __restrict__ x0, T* __restrict__ xstar,
                                       long long P, int n, int m, int m4, int RS, unsigned long long seed,
                                       long long problem0) {
  const long long rows = P * (long long)m4;
  for (long long rg = (long long)blockIdx.x * blockDim.x + threadIdx.x; rg < rows; rg += (long long)gridDim.x * blockDim.x) {
    const long long p = rg / m4;
    const int i = int(rg % m4);
    const unsigned long long pid = (unsigned long long)(problem0 + p);
    T* row = out + rg * RS;
    if (i >= m) {
      for (int q = 0; q < RS; ++q) row[q] = T(0);
    } else {
      const unsigned long long kA = skey(seed, pid, 0), kx = skey(seed, pid, 1), kn = skey(seed, pid, 2);
      double t = 0;
      for (int j = 0; j < n; ++j) {
        const T a = T(u11(kA, (unsigned long long)i * n + j));
        row[j] = a;
        t += double(a) * u11(kx, j);
      }
      row[n] = T(t + 0.1 * sin(t) + 1e-3 * u11(kn, i));
      for (int q = n + 1; q < RS; ++q) row[q] = T(0);
    }
    if (i == 0) {
      const unsigned long long kx = skey(seed, pid, 1), k0 = skey(seed, pid, 3);
      for (int j = 0; j < n; ++j) {
        const double xs = u11(kx, j);
        if (xstar) xstar[p * n + j] = T(xs);
        if (x0) x0[p * n + j] = T(xs + 0.5 * u11(k0, j));
      }
    }
  }
}

}  // namespace toa

// ================================================================================================
// C-ABI
// ================================================================================================
using namespace toa;

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

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIP_TRY(expr)                                                                           \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return fail(e_ == hipErrorOutOfMemory ? TOA_E_NOMEM : TOA_E_HIP,                          \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                           \
  } while (0)

template <typename T, int NB>
static int launch_accumulate(toa_handle h, int n, int m, int64_t P, const void* data, const void* x, int want_grad,
                             void* g, void* H, double* cost, int32_t* nres) {
  long long grid = (P + 3) / 4;
  const long long cap = (long long)h->num_cus * 8;
  if (grid > cap) grid = cap;
  hipLaunchKernelGGL((accumulate_kernel<T, NB>), dim3((unsigned)grid), dim3(256), 0, h->stream, data, x, (long long)P, n, m,
                     want_grad, g, H, cost, nres);
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}

// waves per workgroup is fixed at 4 (256 threads); LDS per wave decides how many WGs fit per CU.
template <typename T>
static int lds_fit(toa_handle h, int n, size_t* per_wave, size_t* per_wg) {
  size_t pw = WaveLds<T>::bytes(n);
  pw = (pw + 15) & ~size_t(15);
  *per_wave = pw;
  *per_wg = pw * 4;
  if (*per_wg > 160 * 1024) return fail(TOA_E_UNSUPPORTED, "LDS footprint exceeds 160 KiB per workgroup");
  (void)h;
  return TOA_OK;
}

template <typename T, int NB>
static int launch_fused(toa_handle h, const FusedParams& prm_in) {
  FusedParams prm = prm_in;
  size_t pw, pwg;
  if (int rc = lds_fit<T>(h, prm.n, &pw, &pwg)) return rc;
  prm.lds_per_wave = (int)pw;
  prm.queue = h->queue;
  HIP_TRY(hipMemsetAsync(h->queue, 0, sizeof(int), h->stream));
  auto kern = lm_fused_kernel<T, NB>;
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

extern "C" {

const char* toa_last_error(void) { return g_err.c_str(); }

void toa_options_default(toa_options* o) {
  std::memset(o, 0, sizeof(*o));
  o->solver_type = 0;
  o->max_iters = 50;
  o->min_error = 1e-12f;
  o->min_rerr_dec = 1e-10f;
  o->min_step_norm2 = 1e-14f;
  o->min_grad_norm2 = 1e-18f;
  o->max_total_failures = 0;
  o->max_consec_failures = 5;
  o->damping_init = 1e-4f;
  o->damping_min = 1e-9f;
  o->damping_max = 1e9f;
  o->good_factor = 1.0f / 3.0f;
  o->bad_factor = 2.0f;
  o->grad_clipping = 0;
  o->check_min_H_diag = 0;
  o->check_final_cost = 0;
  o->use_step_quality_approx = 0;
  o->use_ldlt = 1;
  o->H_is_full = 1;
  o->save_last = 1;
  o->use_squared_norm = 1;
  o->downscale_by_2 = 0;
  o->normalize = 0;
}

void toa_options_benchmark(toa_options* o) {
  toa_options_default(o);
  o->max_iters = 10;
  o->min_error = 0;
  o->min_rerr_dec = 1e-12f;
  o->min_step_norm2 = 1e-16f;
  o->max_consec_failures = 3;
  o->save_last = 0;
}

int toa_create(toa_handle* out, int device, void* stream) {
  if (!out) return fail(TOA_E_ARG, "toa_create: out is null");
  int count = 0;
  HIP_TRY(hipGetDeviceCount(&count));
  if (device < 0 || device >= count) return fail(TOA_E_ARG, "toa_create: no such device");
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(TOA_E_UNSUPPORTED, std::string("toa_create: this library is built for gfx950 only, device is ") + prop.gcnArchName);
  toa_context* c = new (std::nothrow) toa_context();
  if (!c) return fail(TOA_E_NOMEM, "toa_create: host allocation failed");
  c->device = device;
  c->stream = static_cast<hipStream_t>(stream);
  c->num_cus = prop.multiProcessorCount;
  c->clock_khz = prop.clockRate;
  c->max_lds = int(prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor : prop.sharedMemPerBlock);
  std::strncpy(c->name, prop.name, sizeof(c->name) - 1);
  hipError_t e = hipMalloc(&c->queue, 256);
  if (e != hipSuccess) { delete c; return fail(TOA_E_NOMEM, "toa_create: hipMalloc(queue) failed"); }
  e = hipMalloc(&c->params_dev, 1024);
  if (e != hipSuccess) { (void)hipFree(c->queue); delete c; return fail(TOA_E_NOMEM, "toa_create: hipMalloc(params) failed"); }
  *out = c;
  return TOA_OK;
}

int toa_destroy(toa_handle h) {
  if (!h) return TOA_OK;
  (void)hipSetDevice(h->device);
  if (h->queue) (void)hipFree(h->queue);
  if (h->params_dev) (void)hipFree(h->params_dev);
  delete h;
  return TOA_OK;
}

int toa_device_info(toa_handle h, int* num_cus, int* clock_khz, char* name, size_t name_len) {
  if (!h) return fail(TOA_E_ARG, "null handle");
  if (num_cus) *num_cus = h->num_cus;
  if (clock_khz) *clock_khz = h->clock_khz;
  if (name && name_len) { std::strncpy(name, h->name, name_len - 1); name[name_len - 1] = 0; }
  return TOA_OK;
}

int toa_malloc(toa_handle h, void** dev_ptr, size_t bytes) {
  if (!h || !dev_ptr) return fail(TOA_E_ARG, "toa_malloc: null argument");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipMalloc(dev_ptr, bytes ? bytes : 1));
  return TOA_OK;
}
int toa_free(toa_handle h, void* dev_ptr) {
  if (!h) return fail(TOA_E_ARG, "null handle");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipFree(dev_ptr));
  return TOA_OK;
}
int toa_memcpy_h2d(toa_handle h, void* dst, const void* src, size_t bytes) {
  if (!h) return fail(TOA_E_ARG, "null handle");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return TOA_OK;
}
int toa_memcpy_d2h(toa_handle h, void* dst, const void* src, size_t bytes) {
  if (!h) return fail(TOA_E_ARG, "null handle");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return TOA_OK;
}
int toa_memset(toa_handle h, void* dst, int value, size_t bytes) {
  if (!h) return fail(TOA_E_ARG, "null handle");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipMemsetAsync(dst, value, bytes, h->stream));
  return TOA_OK;
}
int toa_synchronize(toa_handle h) {
  if (!h) return fail(TOA_E_ARG, "null handle");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return TOA_OK;
}

static int check_shape(int dtype, int n, int m, int64_t P) {
  if (dtype != TOA_F32 && dtype != TOA_F64) return fail(TOA_E_ARG, "dtype must be TOA_F32 or TOA_F64");
  if (n < 1 || n > 63) return fail(TOA_E_ARG, "n must be in [1, 63] on the LDS-resident path");
  if (m < 1) return fail(TOA_E_ARG, "m must be >= 1");
  if (P < 0 || P > 0x7fffffff) return fail(TOA_E_ARG, "P out of range");
  return TOA_OK;
}

int toa_dense_row_layout(int dtype, int n, int m, int* nb, int* row_stride, int* rows_padded, size_t* bytes_per_problem) {
  if (int rc = check_shape(dtype, n, m, 0)) return rc;
  const DenseRowLayout L = DenseRowLayout::make(n, m);
  if (nb) *nb = L.nb;
  if (row_stride) *row_stride = L.rs;
  if (rows_padded) *rows_padded = L.m4;
  if (bytes_per_problem) *bytes_per_problem = L.elems_per_problem() * (dtype == TOA_F32 ? 4 : 8);
  return TOA_OK;
}

int toa_dense_row_pack(toa_handle h, int dtype, int n, int m, int64_t P, const void* A, const void* b, void* packed) {
  if (!h) return fail(TOA_E_ARG, "null handle");
  if (int rc = check_shape(dtype, n, m, P)) return rc;
  if (P == 0) return TOA_OK;
  HIP_TRY(hipSetDevice(h->device));
  const DenseRowLayout L = DenseRowLayout::make(n, m);
  const int grid = h->num_cus * 8;
  if (dtype == TOA_F32)
    hipLaunchKernelGGL(dense_row_pack_kernel<float>, dim3(grid), dim3(256), 0, h->stream, (const float*)A, (const float*)b,
                       (float*)packed, (long long)P, n, m, L.m4, L.rs);
  else
    hipLaunchKernelGGL(dense_row_pack_kernel<double>, dim3(grid), dim3(256), 0, h->stream, (const double*)A, (const double*)b,
                       (double*)packed, (long long)P, n, m, L.m4, L.rs);
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}

int toa_dense_row_synth(toa_handle h, int dtype, int n, int m, int64_t P, uint64_t seed, int64_t problem0,
                        void* packed, void* x0, void* xstar) {
  if (!h) return fail(TOA_E_ARG, "null handle");
  if (int rc = check_shape(dtype, n, m, P)) return rc;
  if (P == 0) return TOA_OK;
  HIP_TRY(hipSetDevice(h->device));
  const DenseRowLayout L = DenseRowLayout::make(n, m);
  const int grid = h->num_cus * 16;
  if (dtype == TOA_F32)
    hipLaunchKernelGGL(dense_row_synth_kernel<float>, dim3(grid), dim3(256), 0, h->stream, (float*)packed, (float*)x0,
                       (float*)xstar, (long long)P, n, m, L.m4, L.rs, (unsigned long long)seed, (long long)problem0);
  else
    hipLaunchKernelGGL(dense_row_synth_kernel<double>, dim3(grid), dim3(256), 0, h->stream, (double*)packed, (double*)x0,
                       (double*)xstar, (long long)P, n, m, L.m4, L.rs, (unsigned long long)seed, (long long)problem0);
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}

int toa_accumulate(toa_handle h, int model, int dtype, int n, int m, int64_t P, const void* data, const void* x,
                   int want_grad, void* g, void* H, double* cost, int32_t* nres) {
  if (!h) return fail(TOA_E_ARG, "null handle");
  if (model != TOA_MODEL_DENSE_ROW) return fail(TOA_E_UNSUPPORTED, "toa_accumulate: model not available on this path");
  if (int rc = check_shape(dtype, n, m, P)) return rc;
  if (!data || !x || !cost || (want_grad && (!g || !H))) return fail(TOA_E_ARG, "toa_accumulate: null pointer");
  if (P == 0) return TOA_OK;
  HIP_TRY(hipSetDevice(h->device));
  const int nb = DenseRowLayout::make(n, m).nb;
#define TOA_ACC(T, NB) return launch_accumulate<T, NB>(h, n, m, P, data, x, want_grad, g, H, cost, nres)
  if (dtype == TOA_F32) {
    switch (nb) { case 1: TOA_ACC(float, 1); case 2: TOA_ACC(float, 2); case 3: TOA_ACC(float, 3); case 4: TOA_ACC(float, 4); }
  } else {
    switch (nb) { case 1: TOA_ACC(double, 1); case 2: TOA_ACC(double, 2); case 3: TOA_ACC(double, 3); case 4: TOA_ACC(double, 4); }
  }
#undef TOA_ACC
  return fail(TOA_E_ARG, "toa_accumulate: bad block count");
}

int toa_solve_damped(toa_handle h, int dtype, int n, int64_t P, const void* H, const void* g, double scale, void* dx,
                     int32_t* ok) {
  if (!h) return fail(TOA_E_ARG, "null handle");
  if (int rc = check_shape(dtype, n, 1, P)) return rc;
  if (!H || !g || !dx || !ok) return fail(TOA_E_ARG, "toa_solve_damped: null pointer");
  if (P == 0) return TOA_OK;
  HIP_TRY(hipSetDevice(h->device));
  long long grid = (P + 3) / 4;
  const long long cap = (long long)h->num_cus * 8;
  if (grid > cap) grid = cap;
  size_t pw, pwg;
  if (dtype == TOA_F32) {
    if (int rc = lds_fit<float>(h, n, &pw, &pwg)) return rc;
    HIP_TRY(hipFuncSetAttribute((const void*)solve_damped_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pwg));
    hipLaunchKernelGGL(solve_damped_kernel<float>, dim3((unsigned)grid), dim3(256), pwg, h->stream, H, g, (long long)P, n, scale,
                       dx, ok, (int)pw);
  } else {
    if (int rc = lds_fit<double>(h, n, &pw, &pwg)) return rc;
    HIP_TRY(hipFuncSetAttribute((const void*)solve_damped_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pwg));
    hipLaunchKernelGGL(solve_damped_kernel<double>, dim3((unsigned)grid), dim3(256), pwg, h->stream, H, g, (long long)P, n,
                       scale, dx, ok, (int)pw);
  }
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}

int toa_lm_run(toa_handle h, int model, int dtype, int n, int m, int64_t P, const void* data, void* x,
               const toa_options* options, const toa_results* results, uint64_t* counters) {
  if (!h) return fail(TOA_E_ARG, "null handle");
  if (model != TOA_MODEL_DENSE_ROW) return fail(TOA_E_UNSUPPORTED, "toa_lm_run: model not available on this path");
  if (int rc = check_shape(dtype, n, m, P)) return rc;
  if (!data || !x || !options || !results) return fail(TOA_E_ARG, "toa_lm_run: null pointer");
  if (!results->stop_reason || !results->num_iters || !results->final_cost)
    return fail(TOA_E_ARG, "toa_lm_run: stop_reason, num_iters and final_cost outputs are required");
  if (options->solver_type != 0 && options->solver_type != 1)
    return fail(TOA_E_ARG, "toa_lm_run: solver_type must be 0 (LM) or 1 (GN) on this path");  // optimize.h:75
  if (!options->use_ldlt && n > 1)
    return fail(TOA_E_UNSUPPORTED, "toa_lm_run: use_ldlt=false is only implemented for n == 1 (gn.h:157-162)");
  if ((results->errs || results->deltas2 || results->successes) && results->hist_stride < options->max_iters + 2)
    return fail(TOA_E_ARG, "toa_lm_run: hist_stride must be >= max_iters + 2");
  if (options->max_iters < 0 || options->max_iters > 65535) return fail(TOA_E_ARG, "max_iters out of range");
  if (P == 0) return TOA_OK;
  HIP_TRY(hipSetDevice(h->device));
  FusedParams prm;
  std::memset(&prm, 0, sizeof(prm));
  prm.data = data;
  prm.x = x;
  prm.P = P;
  prm.n = n;
  prm.m = m;
  prm.opt = *options;
  prm.res = *results;
  prm.counters = reinterpret_cast<unsigned long long*>(counters);
  const int nb = DenseRowLayout::make(n, m).nb;
#define TOA_RUN(T, NB) return launch_fused<T, NB>(h, prm)
  if (dtype == TOA_F32) {
    switch (nb) { case 1: TOA_RUN(float, 1); case 2: TOA_RUN(float, 2); case 3: TOA_RUN(float, 3); case 4: TOA_RUN(float, 4); }
  } else {
    switch (nb) { case 1: TOA_RUN(double, 1); case 2: TOA_RUN(double, 2); case 3: TOA_RUN(double, 3); case 4: TOA_RUN(double, 4); }
  }
#undef TOA_RUN
  return fail(TOA_E_ARG, "toa_lm_run: bad block count");
}

}  // extern "C"
