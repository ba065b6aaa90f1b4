// K3 for large n (n > 63): the damped normal-equations solve through rocSOLVER's batched Cholesky.
//
// Replaces the same reference code as the register / LDS path (SolverLM::Build's damping lm.h:108-117 +
// SolverGN::Solve gn.h:150-171 -> SolveLDLT math.h:232-240) for matrices that no longer fit one wavefront:
//   H_ii <- H_ii * scale ; potrf_strided_batched ; potrs_strided_batched ; dx = -solution.
// Acceptance differs from Eigen's LDLT only for singular positive SEMI-definite matrices (Eigen: Success &&
// isPositive() accepts them, Cholesky reports a zero pivot); the damped matrices of an LM run are definite.
//
// rocSOLVER / rocBLAS are opened with dlopen on first use: the 900 MB library is not a link-time dependency of the
// n <= 63 product path and costs nothing until a large-n solve is requested.  No CPU fallback: if the libraries
// are missing the call fails with TOA_E_UNSUPPORTED.
#include <type_traits>
#include <dlfcn.h>

#include <mutex>
#include <string>

#include "kernels.hpp"
#include "ldlt_wg.hpp"
#include "chol_diag.hpp"

namespace toa {
namespace {

using rb_handle = void*;
constexpr int kFillUpper = 121;  // rocblas_fill_upper (rocblas-types.h)
constexpr int kOpN = 111, kOpT = 112;  // rocblas_operation_none / _transpose

struct RocApi {
  int (*create)(rb_handle*) = nullptr;
  int (*destroy)(rb_handle) = nullptr;
  int (*set_stream)(rb_handle, hipStream_t) = nullptr;
  int (*spotrf)(rb_handle, int, int, float*, int, int64_t, int*, int) = nullptr;
  int (*dpotrf)(rb_handle, int, int, double*, int, int64_t, int*, int) = nullptr;
  int (*spotrs)(rb_handle, int, int, int, float*, int, int64_t, float*, int, int64_t, int) = nullptr;
  // general LU for use_ldlt = false (gn.h:157-162: dx = -H.inverse() * g, unchecked)
  int (*sgetrf)(rb_handle, int, int, float*, int, int64_t, int*, int64_t, int*, int) = nullptr;
  int (*dgetrf)(rb_handle, int, int, double*, int, int64_t, int*, int64_t, int*, int) = nullptr;
  int (*sgetrs)(rb_handle, int, int, int, float*, int, int64_t, const int*, int64_t, float*, int, int64_t, int) = nullptr;
  int (*dgetrs)(rb_handle, int, int, int, double*, int, int64_t, const int*, int64_t, double*, int, int64_t, int) = nullptr;
  int (*dpotrs)(rb_handle, int, int, int, double*, int, int64_t, double*, int, int64_t, int) = nullptr;
  int (*sgemm)(rb_handle, int, int, int, int, int, const float*, const float*, int, int64_t, const float*, int, int64_t,
               const float*, float*, int, int64_t, int) = nullptr;
  int (*dgemm)(rb_handle, int, int, int, int, int, const double*, const double*, int, int64_t, const double*, int, int64_t,
               const double*, double*, int, int64_t, int) = nullptr;
  int (*sgemm_b)(rb_handle, int, int, int, int, int, const float*, const float* const*, int, const float* const*, int, const float*,
                 float* const*, int, int) = nullptr;
  int (*dgemm_b)(rb_handle, int, int, int, int, int, const double*, const double* const*, int, const double* const*, int,
                 const double*, double* const*, int, int) = nullptr;
  int (*sgemv)(rb_handle, int, int, int, const float*, const float*, int, int64_t, const float*, int, int64_t, const float*,
               float*, int, int64_t, int) = nullptr;
  int (*dgemv)(rb_handle, int, int, int, const double*, const double*, int, int64_t, const double*, int, int64_t,
               const double*, double*, int, int64_t, int) = nullptr;
  std::string err;
  bool ok = false;
};

RocApi& roc_api() {
  static RocApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    void* blas = dlopen("librocblas.so", RTLD_NOW | RTLD_GLOBAL);
    if (!blas) blas = dlopen("/opt/rocm/lib/librocblas.so", RTLD_NOW | RTLD_GLOBAL);
    void* sol = dlopen("librocsolver.so", RTLD_NOW | RTLD_GLOBAL);
    if (!sol) sol = dlopen("/opt/rocm/lib/librocsolver.so", RTLD_NOW | RTLD_GLOBAL);
    if (!blas || !sol) {
      api.err = std::string("cannot open rocBLAS / rocSOLVER: ") + (dlerror() ? dlerror() : "?");
      return;
    }
    auto sym = [&](void* lib, const char* name) -> void* {
      void* p = dlsym(lib, name);
      if (!p && api.err.empty()) api.err = std::string("missing symbol ") + name;
      return p;
    };
    api.create = reinterpret_cast<decltype(api.create)>(sym(blas, "rocblas_create_handle"));
    api.destroy = reinterpret_cast<decltype(api.destroy)>(sym(blas, "rocblas_destroy_handle"));
    api.set_stream = reinterpret_cast<decltype(api.set_stream)>(sym(blas, "rocblas_set_stream"));
    api.spotrf = reinterpret_cast<decltype(api.spotrf)>(sym(sol, "rocsolver_spotrf_strided_batched"));
    api.dpotrf = reinterpret_cast<decltype(api.dpotrf)>(sym(sol, "rocsolver_dpotrf_strided_batched"));
    api.spotrs = reinterpret_cast<decltype(api.spotrs)>(sym(sol, "rocsolver_spotrs_strided_batched"));
    api.dpotrs = reinterpret_cast<decltype(api.dpotrs)>(sym(sol, "rocsolver_dpotrs_strided_batched"));
    api.sgetrf = reinterpret_cast<decltype(api.sgetrf)>(sym(sol, "rocsolver_sgetrf_strided_batched"));
    api.dgetrf = reinterpret_cast<decltype(api.dgetrf)>(sym(sol, "rocsolver_dgetrf_strided_batched"));
    api.sgetrs = reinterpret_cast<decltype(api.sgetrs)>(sym(sol, "rocsolver_sgetrs_strided_batched"));
    api.dgetrs = reinterpret_cast<decltype(api.dgetrs)>(sym(sol, "rocsolver_dgetrs_strided_batched"));
    api.sgemm = reinterpret_cast<decltype(api.sgemm)>(sym(blas, "rocblas_sgemm_strided_batched"));
    api.dgemm = reinterpret_cast<decltype(api.dgemm)>(sym(blas, "rocblas_dgemm_strided_batched"));
    api.sgemm_b = reinterpret_cast<decltype(api.sgemm_b)>(sym(blas, "rocblas_sgemm_batched"));
    api.dgemm_b = reinterpret_cast<decltype(api.dgemm_b)>(sym(blas, "rocblas_dgemm_batched"));
    api.sgemv = reinterpret_cast<decltype(api.sgemv)>(sym(blas, "rocblas_sgemv_strided_batched"));
    api.dgemv = reinterpret_cast<decltype(api.dgemv)>(sym(blas, "rocblas_dgemv_strided_batched"));
    api.ok = api.err.empty();
  });
  return api;
}

// Grow the context's scratch block (shared with the row-split path; contents are per-call) to at least `need` bytes.
int ensure_scratch(toa_handle h, size_t need, const char* what) {
  // toa_tuning::fail_workspace_alloc: a test hook — the request fails as on an exhausted device
  if (h->tune.fail_workspace_alloc)
    return toa_fail(TOA_E_NOMEM, std::string(what) + ": cannot allocate " + std::to_string(need >> 20) + " MiB of device workspace (toa_tuning::fail_workspace_alloc)");
  if (need <= h->scratch_bytes) return TOA_OK;
  if (int rc = grow_sync(h, what)) return rc;
  toa_release_workspace(h, h->scratch);
  h->scratch = nullptr;
  h->scratch_bytes = 0;
  if (hipMalloc(&h->scratch, need) != hipSuccess) {
    (void)hipGetLastError();
    return toa_fail(TOA_E_NOMEM, std::string(what) + ": cannot allocate " + std::to_string(need >> 20) + " MiB of device workspace");
  }
  h->scratch_bytes = need;
  return TOA_OK;
}

// The context's rocBLAS handle (created on first use), bound to the context's stream.
int ensure_blas(toa_handle h, RocApi& api) {
  if (!h->blas) {
    if (api.create(&h->blas) != 0) return toa_fail(TOA_E_HIP, "rocblas_create_handle failed");
    h->blas_destroy = api.destroy;
  }
  if (api.set_stream(h->blas, h->stream) != 0) return toa_fail(TOA_E_HIP, "rocblas_set_stream failed");
  return TOA_OK;
}

// work[p] = H[p] with the diagonal scaled (in double, like lm.h:108-117); rhs[p] = g[p]
template <typename T>
__global__ void __launch_bounds__(256) large_damp_kernel(const T* __restrict__ H, const T* __restrict__ g, T* __restrict__ work,
                                                         T* __restrict__ rhs, const int n, const double scale) {
  const size_t p = blockIdx.y;
  const size_t nn = size_t(n) * n;
  const T* Hp = H + p * nn;
  T* Wp = work + p * nn;
  for (size_t e = size_t(blockIdx.x) * blockDim.x + threadIdx.x; e < nn; e += size_t(gridDim.x) * blockDim.x) {
    const int i = int(e / n), j = int(e % n);
    const T v = Hp[e];
    Wp[e] = (i == j) ? T(double(v) * scale) : v;
  }
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < n; i += blockDim.x) rhs[p * n + i] = g[p * n + i];
}

// dx = -solution where the factorisation succeeded and the solution is finite; ok flags as on the small-n path
template <typename T>
__global__ void __launch_bounds__(256) large_finish_kernel(const T* __restrict__ sol, const int* __restrict__ info,
                                                           T* __restrict__ dx, int32_t* __restrict__ ok, const int n,
                                                           const int* __restrict__ active = nullptr) {
  const size_t p = blockIdx.x;
  if (active && !active[p]) {   // a masked-out matrix was not factorised: no step, no verdict
    for (int i = threadIdx.x; i < n; i += blockDim.x) dx[p * n + i] = T(0);
    if (threadIdx.x == 0) ok[p] = 0;
    return;
  }
  __shared__ int bad;
  if (threadIdx.x == 0) bad = info[p] != 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const T v = sol[p * n + i];
    if (!(fabs(v) <= NumLimits<T>::max())) bad = 1;  // benign race: every writer stores 1
  }
  __syncthreads();
  const bool good = !bad;
  for (int i = threadIdx.x; i < n; i += blockDim.x) dx[p * n + i] = good ? -sol[p * n + i] : T(0);
  if (threadIdx.x == 0) ok[p] = good ? 1 : 0;
}

template <typename T>
int large_solve_t(toa_handle h, RocApi& api, int n, int64_t P, const T* H, const T* g, double scale, T* dx, int32_t* ok) {
  const size_t nn = size_t(n) * n;
  const size_t b_work = (size_t(P) * nn * sizeof(T) + 255) & ~size_t(255);
  const size_t b_rhs = (size_t(P) * n * sizeof(T) + 255) & ~size_t(255);
  const size_t b_info = (size_t(P) * sizeof(int) + 255) & ~size_t(255);
  const size_t need = b_work + b_rhs + b_info;
  if (int rc = ensure_scratch(h, need, "large-n solve")) return rc;
  char* base = static_cast<char*>(h->scratch);
  T* work = reinterpret_cast<T*>(base);
  T* rhs = reinterpret_cast<T*>(base + b_work);
  int* info = reinterpret_cast<int*>(base + b_work + b_rhs);
  if (int rc = ensure_blas(h, api)) return rc;
  const unsigned gx = unsigned(std::min<size_t>((nn + 255) / 256, 64));
  hipLaunchKernelGGL(large_damp_kernel<T>, dim3(gx, unsigned(P)), dim3(256), 0, h->stream, H, g, work, rhs, n, scale);
  HIP_TRY(hipGetLastError());
  int rc;
  if constexpr (sizeof(T) == 4) {
    rc = api.spotrf(h->blas, kFillUpper, n, work, n, int64_t(nn), info, int(P));
    if (rc == 0) rc = api.spotrs(h->blas, kFillUpper, n, 1, work, n, int64_t(nn), rhs, n, int64_t(n), int(P));
  } else {
    rc = api.dpotrf(h->blas, kFillUpper, n, work, n, int64_t(nn), info, int(P));
    if (rc == 0) rc = api.dpotrs(h->blas, kFillUpper, n, 1, work, n, int64_t(nn), rhs, n, int64_t(n), int(P));
  }
  if (rc != 0) return toa_fail(TOA_E_HIP, "rocSOLVER potrf/potrs returned status " + std::to_string(rc));
  hipLaunchKernelGGL(large_finish_kernel<T>, dim3(unsigned(P)), dim3(256), 0, h->stream, rhs, info, dx, ok, n);
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}

// dx = -H^-1 g "without any checks on invertibility" (gn.h:157-162, options.h:59: use_ldlt = false): a general LU with partial
// pivoting from the library, its verdict ignored (a solution that is not finite still ends as ok = 0: the step is refused as
// everywhere else).  ONE matrix per call: a matrix solved alone gives the bits of its row in a batch.
template <typename T>
int large_solve_lu_t(toa_handle h, RocApi& api, int n, int64_t P, const T* H, const T* g, double scale, T* dx, int32_t* ok) {
  const size_t nn = size_t(n) * n;
  const size_t b_work = (size_t(P) * nn * sizeof(T) + 255) & ~size_t(255);
  const size_t b_rhs = (size_t(P) * n * sizeof(T) + 255) & ~size_t(255);
  const size_t b_info = (size_t(P) * sizeof(int) + 255) & ~size_t(255);
  const size_t b_piv = (size_t(P) * n * sizeof(int) + 255) & ~size_t(255);
  if (int rc = ensure_scratch(h, b_work + b_rhs + b_info + b_piv, "large-n solve")) return rc;
  char* base = static_cast<char*>(h->scratch);
  T* work = reinterpret_cast<T*>(base);
  T* rhs = reinterpret_cast<T*>(base + b_work);
  int* info = reinterpret_cast<int*>(base + b_work + b_rhs);
  int* ipiv = reinterpret_cast<int*>(base + b_work + b_rhs + b_info);
  if (int rc = ensure_blas(h, api)) return rc;
  const unsigned gx = unsigned(std::min<size_t>((nn + 255) / 256, 64));
  hipLaunchKernelGGL(large_damp_kernel<T>, dim3(gx, unsigned(P)), dim3(256), 0, h->stream, H, g, work, rhs, n, scale);
  HIP_TRY(hipGetLastError());
  for (int64_t p = 0; p < P; ++p) {
    int rc;
    if constexpr (sizeof(T) == 4) {
      rc = api.sgetrf(h->blas, n, n, work + p * nn, n, int64_t(nn), ipiv + p * n, int64_t(n), info + p, 1);
      if (rc == 0) rc = api.sgetrs(h->blas, kOpN, n, 1, work + p * nn, n, int64_t(nn), ipiv + p * n, int64_t(n), rhs + p * n, n, int64_t(n), 1);
    } else {
      rc = api.dgetrf(h->blas, n, n, work + p * nn, n, int64_t(nn), ipiv + p * n, int64_t(n), info + p, 1);
      if (rc == 0) rc = api.dgetrs(h->blas, kOpN, n, 1, work + p * nn, n, int64_t(nn), ipiv + p * n, int64_t(n), rhs + p * n, n, int64_t(n), 1);
    }
    if (rc != 0) return toa_fail(TOA_E_HIP, "rocSOLVER getrf/getrs returned status " + std::to_string(rc));
  }
  HIP_TRY(hipMemsetAsync(info, 0, size_t(P) * sizeof(int), h->stream));   // unchecked: a singular pivot is not a failure by itself
  hipLaunchKernelGGL(large_finish_kernel<T>, dim3(unsigned(P)), dim3(256), 0, h->stream, rhs, info, dx, ok, n);
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}

// The same solve with ONE matrix per library call — rocSOLVER's batched Cholesky picks its blocking from the batch size, and
// a caller that promises "a problem solved alone gives the bits of its row in a batch" (bundle adjustment with visibility
// lists) cannot use it — but the calls spread over up to eight side streams, each with a rocBLAS handle (and so a device
// workspace) of its own: the P independent factorisations, one small workgroup-count each, overlap instead of queueing
// behind one another on the context's stream.  Fork / join by events; everything else stays on the context's stream.
template <typename T>
int large_solve_each_t(toa_handle h, RocApi& api, int n, int64_t P, const T* H, const T* g, double scale, T* dx, int32_t* ok) {
  const size_t nn = size_t(n) * n;
  const size_t b_work = (size_t(P) * nn * sizeof(T) + 255) & ~size_t(255);
  const size_t b_rhs = (size_t(P) * n * sizeof(T) + 255) & ~size_t(255);
  const size_t b_info = (size_t(P) * sizeof(int) + 255) & ~size_t(255);
  if (int rc = ensure_scratch(h, b_work + b_rhs + b_info, "large-n solve")) return rc;
  char* base = static_cast<char*>(h->scratch);
  T* work = reinterpret_cast<T*>(base);
  T* rhs = reinterpret_cast<T*>(base + b_work);
  int* info = reinterpret_cast<int*>(base + b_work + b_rhs);
  if (int rc = ensure_blas(h, api)) return rc;   // (also records how to destroy a handle)
  const int K = int(std::min<int64_t>(toa_context::kSide, P));
  while (h->nside < K) {
    const int i = h->nside;
    HIP_TRY(hipStreamCreateWithFlags(&h->side_stream[i], hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&h->side_done[i], hipEventDisableTiming));
    if (api.create(&h->side_blas[i]) != 0) return toa_fail(TOA_E_HIP, "rocblas_create_handle failed");
    if (api.set_stream(h->side_blas[i], h->side_stream[i]) != 0) return toa_fail(TOA_E_HIP, "rocblas_set_stream failed");
    h->nside = i + 1;
  }
  if (!h->side_fork) HIP_TRY(hipEventCreateWithFlags(&h->side_fork, hipEventDisableTiming));
  const unsigned gx = unsigned(std::min<size_t>((nn + 255) / 256, 64));
  hipLaunchKernelGGL(large_damp_kernel<T>, dim3(gx, unsigned(P)), dim3(256), 0, h->stream, H, g, work, rhs, n, scale);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(h->side_fork, h->stream));
  for (int i = 0; i < K; ++i) HIP_TRY(hipStreamWaitEvent(h->side_stream[i], h->side_fork, 0));
  int rc = 0;
  for (int64_t q = 0; q < P && rc == 0; ++q) {
    void* hb = h->side_blas[q % K];
    if constexpr (sizeof(T) == 4) {
      rc = api.spotrf(hb, kFillUpper, n, work + q * nn, n, int64_t(nn), info + q, 1);
      if (rc == 0) rc = api.spotrs(hb, kFillUpper, n, 1, work + q * nn, n, int64_t(nn), rhs + q * n, n, int64_t(n), 1);
    } else {
      rc = api.dpotrf(hb, kFillUpper, n, work + q * nn, n, int64_t(nn), info + q, 1);
      if (rc == 0) rc = api.dpotrs(hb, kFillUpper, n, 1, work + q * nn, n, int64_t(nn), rhs + q * n, n, int64_t(n), 1);
    }
  }
  for (int i = 0; i < K; ++i) {   // join (also after a failed call: nothing may still be running on the side when we return)
    HIP_TRY(hipEventRecord(h->side_done[i], h->side_stream[i]));
    HIP_TRY(hipStreamWaitEvent(h->stream, h->side_done[i], 0));
  }
  if (rc != 0) return toa_fail(TOA_E_HIP, "rocSOLVER potrf/potrs returned status " + std::to_string(rc));
  hipLaunchKernelGGL(large_finish_kernel<T>, dim3(unsigned(P)), dim3(256), 0, h->stream, rhs, info, dx, ok, n);
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}

// C = H^-1 for n > 63 (tinyopt::InvCov, math.h:41-91): Cholesky against the identity.
template <typename T>
__global__ void __launch_bounds__(256) large_identity_kernel(T* __restrict__ C, const int n) {
  const size_t p = blockIdx.y, nn = size_t(n) * n;
  for (size_t e = size_t(blockIdx.x) * 256 + threadIdx.x; e < nn; e += size_t(gridDim.x) * 256) C[p * nn + e] = (e / n == e % n) ? T(1) : T(0);
}
template <typename T>
__global__ void __launch_bounds__(256) large_inv_finish_kernel(T* __restrict__ C, const int* __restrict__ info, int32_t* __restrict__ ok, const int n) {
  __shared__ int bad;
  const size_t p = blockIdx.x, nn = size_t(n) * n;
  if (threadIdx.x == 0) bad = info[p] != 0;
  __syncthreads();
  for (size_t e = threadIdx.x; e < nn; e += 256)
    if (!(fabs(C[p * nn + e]) <= NumLimits<T>::max())) bad = 1;
  __syncthreads();
  if (bad) for (size_t e = threadIdx.x; e < nn; e += 256) C[p * nn + e] = T(0);
  if (threadIdx.x == 0) ok[p] = bad ? 0 : 1;
}

template <typename T>
int large_inv_t(toa_handle h, RocApi& api, int n, int64_t P, const T* H, T* Cm, int32_t* ok) {
  const size_t nn = size_t(n) * n;
  const size_t b_work = (size_t(P) * nn * sizeof(T) + 255) & ~size_t(255);
  const size_t b_info = (size_t(P) * sizeof(int) + 255) & ~size_t(255);
  const size_t need = b_work + b_info;
  if (int rc = ensure_scratch(h, need, "large-n solve")) return rc;
  T* work = reinterpret_cast<T*>(h->scratch);
  int* info = reinterpret_cast<int*>(static_cast<char*>(h->scratch) + b_work);
  if (int rc = ensure_blas(h, api)) return rc;
  HIP_TRY(hipMemcpyAsync(work, H, size_t(P) * nn * sizeof(T), hipMemcpyDeviceToDevice, h->stream));
  const unsigned gx = unsigned(std::min<size_t>((nn + 255) / 256, 64));
  hipLaunchKernelGGL(large_identity_kernel<T>, dim3(gx, unsigned(P)), dim3(256), 0, h->stream, Cm, n);
  int rc;
  if constexpr (sizeof(T) == 4) {
    rc = api.spotrf(h->blas, kFillUpper, n, work, n, int64_t(nn), info, int(P));
    if (rc == 0) rc = api.spotrs(h->blas, kFillUpper, n, n, work, n, int64_t(nn), Cm, n, int64_t(nn), int(P));
  } else {
    rc = api.dpotrf(h->blas, kFillUpper, n, work, n, int64_t(nn), info, int(P));
    if (rc == 0) rc = api.dpotrs(h->blas, kFillUpper, n, n, work, n, int64_t(nn), Cm, n, int64_t(nn), int(P));
  }
  if (rc != 0) return toa_fail(TOA_E_HIP, "rocSOLVER potrf/potrs returned status " + std::to_string(rc));
  hipLaunchKernelGGL(large_inv_finish_kernel<T>, dim3(unsigned(P)), dim3(256), 0, h->stream, Cm, info, ok, n);
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}

// =====================================================================================================================
// The LM / GN loop for n > 63 (TOA_MODEL_DENSE_ROW_NATURAL): same state machine as lm_device.hpp (its scalar pieces are
// shared: LmState, lm_good_step / lm_bad_step, lm_judge_core), but the vectors of a problem live in HBM, one
// WORKGROUP owns a problem inside the small kernels, and the two heavy operations are library calls:
//   rows kernel   r_i, and J = diag(s) A written once (HBM-bound: reads m (n + 1), writes m n)
//   rocBLAS       H = J^T J (gemm_strided_batched: at n >= 64 the residual block IS a real dense contraction),
//                 g = J^T r (gemv_strided_batched)
//   pre kernel    cost, Build (lm.h:59-120): clip, diagonal check, damping; forms the damped copy + right-hand side
//   rocSOLVER     potrf + potrs strided batched
//   post kernel   the rest of Step (optimizer.h:354-539) and of OptimizeAcc's loop body (:266-310), finalisation
// One host loop pass = one Build + Solve attempt of every active problem; a failed solve re-damps and retries in the
// next pass without advancing the iteration, exactly like the `while` at optimizer.h:358.  The host reads two
// integers per pass (active problems, problems that want a Jacobian) to stop early and to skip the GEMM.
template <typename T>
struct LargeArgs {
  const T* data;  // per problem: A row-major [m][n], then b [m]   (SURVEY §8d layout)
  T* x;           // [P][n] in / out
  int n, m;
  long long P;
  toa_options opt;
  toa_results res;
  LmState<T>* st;
  int* active;   // [P]
  int* built;    // [P]
  int* summary;  // [2 * passes]: (active, want Jacobian) after each pass
  T *g, *hd, *dx, *ldx, *H, *Hnew, *gnew, *work, *rhs, *J, *r;
  const T** jptr;  // [P] J of the problems that want a Jacobian this pass, in index order (large_compact_kernel)
  T** hptr;        // [P] their Hnew
  T* gpart;    // [P][gslots][n] per-wave partial J^T r of the vectorised rows kernel (0 slots: g comes from the GEMV)
  int gslots;
  int* info;
  unsigned long long* counters;
  // hand-written Gram (large_gram_kernel): the rows kernel leaves the row scales s_i = 1 + 0.1 cos(a_i.x) here instead of
  // writing J = diag(s) A; the Gram kernel re-reads A and forms J^T J on the matrix cores, partial Grams per row chunk
  T* sc;        // [P][m]
  T* gram_part; // [P][gram_R][tiles of the lower triangle][32 * 32]
  int own_gram, gram_R, gram_rows;   // row chunks per problem, rows per chunk (a multiple of 4)
  // stepping form (toa_lm_begin / toa_lm_step / toa_lm_stop): one ITERATION per call.  A solver failure that is retried with a
  // larger damping stays inside its iteration (optimizer.h:370-390), so a step is one pass plus, rarely, retry passes that
  // only the problems still owing their iteration take part in: stepped [p] = 1 once problem p has had its iteration.
  int* stepped;                   // [P], NULL outside the stepping form
  int32_t* active_out;            // optional: += 1 per problem still running after its iteration
  const int32_t* stop_request;    // toa_lm_stop: [P] StopReason to impose (0 = none)
  // Memo of linearisations (round 5; lm_device.hpp lm_build_and_solve / lm_iteration, here per problem): H has TWO slots per
  // problem — the current linearisation (cur [p]) and, while S.memo_valid, the one of the last ACCEPTED point (midx [p]); a
  // fresh accumulation never lands on the parked slot and a memo hit makes it the current one again, so nothing n x n is ever
  // copied for the memo.  skip [p], set by the post kernel for the NEXT pass: 0 = stream the rows, 1 = the linearisation at
  // hand is the one of x (a failed solve re-entering Build), 2 = the parked one is (the roll-back restored x bit for bit).
  // The rows / Gram kernels leave a skipped problem alone.  NULL skip = off (stepping form, toa_tuning::memo_off).
  int *skip, *cur, *midx;
  int *lin_ninl, *memo_ninl;       // [P]: inlier residuals of the current / parked linearisation (with a loss)
  int hslots;                      // H slots per problem (2 with the memo, 1 without)
  // M-estimator of the handle (toa_set_loss; robust_norms.h:20-26, round 5): the rows kernels turn every residual r_i into
  // sqrt(s_i) r_i and its Jacobian row scale into sqrt(s_i) (1 + 0.1 cos), s_i = dl / dn2 at n2 = r_i^2 — so that the Gram, J^T r
  // and everything behind them are the weighted ones — and leave the loss l_i per row (the cost is their sum, in the order of the
  // plain cost's sum) and the inlier count (n2 <= th2; integer atomics: order-free) behind.  loss == TOA_LOSS_L2: nothing of this.
  int loss;
  T th2;
  T* lossv;      // [P][m]
  int* ninl;     // [P], zeroed by the pre kernel once read
  T *hdu, *g_m, *hdu_m, *xs_m;     // [P][n]: undamped diagonal of the current linearisation; memo: J^T r, diagonal, x
  double *lin_cost, *memo_cost;    // [P]: normalised cost of the current / parked linearisation
  // the blocked Cholesky called DIRECTLY on a caller's matrices (toa_large_solve_inplace, round 5): factorised in place in `work`, the
  // right-hand side read from `rhs`, the step -x (or 0 and ok = 0) written to ddx / dok by the kernel itself, matrix p skipped where
  // dmask[p * dmask_stride] == 0 — the three launches around the factorisation (mask -> flags, copy + damping, finish) are gone
  T* ddx = nullptr;
  int32_t* dok = nullptr;
  const int32_t* dmask = nullptr;
  long long dmask_stride = 0;
  __device__ __forceinline__ bool on(const long long p) const { return active[p] != 0 && !(stepped && stepped[p] != 0); }
  __device__ __forceinline__ bool streams(const long long p) const { return on(p) && !(skip && skip[p] != 0); }   // takes part in the data pass
  __device__ __forceinline__ T* Hcur(const long long p) const {
    return H + (hslots > 1 ? size_t(p) * 2 + size_t(cur[p]) : size_t(p)) * size_t(n) * size_t(n);
  }
};

template <typename T>
__device__ __forceinline__ double block_sum(const double v, double* red) {  // fixed-order tree: deterministic
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if (int(threadIdx.x) < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const double out = red[0];
  __syncthreads();
  return out;
}

template <typename T>
__global__ void __launch_bounds__(256) large_init_kernel(const LargeArgs<T> a) {
  const long long p = blockIdx.x * 256ll + threadIdx.x;
  if (p >= a.P) return;
  LmState<T>& S = a.st[p];
  const toa_options& opt = a.opt;
  S.lambda = opt.damping_init; S.prev_lambda = 0; S.bad_factor = opt.bad_factor; S.rebuild = 1;  // lm.h:46-52
  S.final_cost = kDblMax; S.final_nres = 0; S.final_ninl = 0; S.cost_ninl = 0; S.final_rerr = kDblMax;  // output.h:104-117
  S.stop = TOA_STOP_NONE; S.num_iters = 0; S.num_failures = 0; S.num_consec = 0;
  S.cost_val = 0; S.cost_nres = 0;
  S.max_iters = opt.max_iters + 1 + (opt.check_final_cost ? 1 : 0);  // optimizer.h:248-250
  S.has_last_dx = 0; S.last_was_success = 1; S.iter = 0;
  S.acc_passes = S.eval_passes = S.solves = S.problems = 0;
  S.reused_passes = 0;
  S.acc_at_x = 0; S.memo_valid = 0; S.memo_hit = 0;
  a.active[p] = 1;
  a.built[p] = 0;
  if (a.skip) { a.skip[p] = 0; a.cur[p] = 0; a.midx[p] = 0; }
  if (a.ninl) a.ninl[p] = 0;
}

// r_i = a_i.x + 0.1 sin(a_i.x) - b_i ; J_i = (1 + 0.1 cos(a_i.x)) a_i   (SURVEY §8d DenseRow family).
// General form (any n): a wave per row, four rows per trip; J^T r is left to a library GEMV.
template <typename T>
__global__ void __launch_bounds__(256) large_rows_kernel(const LargeArgs<T> a) {
  extern __shared__ char lds_raw[];
  T* xs = reinterpret_cast<T*>(lds_raw);
  const long long p = blockIdx.y;
  if (!a.streams(p)) return;
  const LmState<T>& S = a.st[p];
  const bool want_j = a.opt.solver_type != 0 || S.rebuild;
  const int n = a.n, m = a.m;
  for (int j = threadIdx.x; j < n; j += 256) xs[j] = a.x[p * n + j];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T* A = a.data + size_t(p) * m * (n + 1);
  const T* bv = A + size_t(m) * n;
  T* Jp = a.J + size_t(p) * m * n;
  T* rp = a.r + size_t(p) * m;
  constexpr int R = 4;
  for (int i0 = (blockIdx.x * 4 + wave) * R; i0 < m; i0 += gridDim.x * 4 * R) {
    T t[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      t[k] = 0;
      if (i0 + k < m) {
        const T* row = A + size_t(i0 + k) * n;
        for (int j = lane; j < n; j += 64) t[k] = fma(row[j], xs[j], t[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < R; ++k) t[k] = wave_allreduce_sum(t[k]);
#pragma unroll
    for (int k = 0; k < R; ++k) {
      if (i0 + k >= m) break;
      const T* row = A + size_t(i0 + k) * n;
      T sn, cs;
      sincos_t(t[k], &sn, &cs);
      T ri = t[k] + T(0.1) * sn - bv[i0 + k];
      T sq = T(1);
      if (a.loss != TOA_LOSS_L2) {   // (wave-uniform)
        const T n2 = ri * ri;
        T l, sw;
        robust_norm(a.loss, n2, a.th2, l, sw);
        sq = r_sqrt(sw);
        ri *= sq;
        if (lane == 0) {
          a.lossv[p * m + i0 + k] = l;
          if (n2 <= a.th2) atomicAdd(&a.ninl[p], 1);
        }
      }
      if (lane == 0) rp[i0 + k] = ri;
      if (want_j) {
        const T sc = (T(1) + T(0.1) * cs) * sq;
        if (a.own_gram) { if (lane == 0) a.sc[p * m + i0 + k] = sc; }
        else for (int j = lane; j < n; j += 64) Jp[size_t(i0 + k) * n + j] = sc * row[j];
      }
    }
  }
}

// Vectorised form (n a multiple of the 16-byte vector, problem base 16-byte aligned): LPR = 2^k lanes share a row with
// 16-byte loads / stores, 64 / LPR rows per wave and trip, and each lane keeps the partial J^T r of its own columns in
// registers across all its rows — written once per wave and summed in fixed order by the pre kernel, so the GEMV and
// its second pass over J disappear.  KV = 16-byte vectors per lane and row.
template <typename T, int KV>
__global__ void __launch_bounds__(256) large_rows_vec_kernel(const LargeArgs<T> a, const int LPR) {
  constexpr int VEC = 16 / sizeof(T);
  using V = T __attribute__((ext_vector_type(VEC)));
  extern __shared__ char lds_raw[];
  T* xs = reinterpret_cast<T*>(lds_raw);
  const long long p = blockIdx.y;
  if (!a.streams(p)) return;
  const LmState<T>& S = a.st[p];
  const bool want_j = a.opt.solver_type != 0 || S.rebuild;
  const int n = a.n, m = a.m, nv = n / VEC;
  for (int j = threadIdx.x; j < n; j += 256) xs[j] = a.x[p * n + j];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rl = lane & (LPR - 1), rg = lane / LPR, RPW = 64 / LPR;  // lane in row, row group, rows per wave
  const T* A = a.data + size_t(p) * m * (n + 1);
  const T* bv = A + size_t(m) * n;
  T* Jp = a.J + size_t(p) * m * n;
  T* rp = a.r + size_t(p) * m;
  V xv[KV], gacc[KV];
#pragma unroll
  for (int k = 0; k < KV; ++k) {
    const int c = rl + k * LPR;
    gacc[k] = V(0);
    xv[k] = V(0);
    if (c < nv) xv[k] = *reinterpret_cast<const V*>(xs + c * VEC);
  }
  const int slot = blockIdx.x * 4 + wave;
  // Round 5: R of the wave's trips at a time.  A trip was: one 16-byte load per lane and vector, wait, dot, butterfly, and then the
  // row's SCALAR work — sincos, residual, loss, scale — executed by all 64 lanes for the one or two rows of the trip: ~250 vector
  // instructions per row, and the kernel ran at 1.8 TB/s bound by issue (23 % of the n = 256 pipeline).  Now the R trips' loads go
  // out together, their butterflies interleave, the rows' totals are collected one per lane (lane u of a row group takes trip u's)
  // so that the scalar work runs ONCE per R trips, and scale and residual come back to the row's lanes for J^T r.  Every row sees
  // the arithmetic it saw before and a wave's rows enter J^T r in the same order.
  constexpr int R = KV >= 5 ? 1 : (KV >= 3 ? 2 : (KV == 2 ? 4 : 8));
  const int stride = gridDim.x * 4 * RPW, gbase = lane & ~(LPR - 1);
  for (int ib = slot * RPW; ib < m; ib += R * stride) {
    V av[R][KV];
    T t[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const int i = ib + u * stride + rg;
#pragma unroll
      for (int k = 0; k < KV; ++k) {
        const int c = rl + k * LPR;
        av[u][k] = V(0);
        if (i < m && c < nv) av[u][k] = __builtin_nontemporal_load(reinterpret_cast<const V*>(A + size_t(i) * n) + c);
      }
    }
#pragma unroll
    for (int u = 0; u < R; ++u) {
      t[u] = 0;
#pragma unroll
      for (int k = 0; k < KV; ++k)
#pragma unroll
        for (int e = 0; e < VEC; ++e) t[u] = fma(av[u][k][e], xv[k][e], t[u]);
    }
    for (int off = 1; off < LPR; off <<= 1) {
#pragma unroll
      for (int u = 0; u < R; ++u) t[u] += __shfl_xor(t[u], off);
    }
    T tt = 0;
#pragma unroll
    for (int u = 0; u < R; ++u) tt = rl == u ? t[u] : tt;
    const int i = ib + rl * stride + rg;             // lane rl < R of a row group: the row of trip rl
    const bool live = rl < R && i < m;
    T sn, cs;
    sincos_t(tt, &sn, &cs);
    T ri = live ? tt + T(0.1) * sn - bv[live ? i : 0] : T(0);
    T sq = T(1);
    if (a.loss != TOA_LOSS_L2) {   // (wave-uniform)
      const T n2 = ri * ri;
      T l, sw;
      robust_norm(a.loss, n2, a.th2, l, sw);
      sq = r_sqrt(sw);
      ri *= sq;
      if (live) {
        a.lossv[p * m + i] = l;
        if (n2 <= a.th2) atomicAdd(&a.ninl[p], 1);
      }
    }
    if (live) rp[i] = ri;
    if (want_j) {
      const T sc = (T(1) + T(0.1) * cs) * sq;
      if (a.own_gram && live) a.sc[p * m + i] = sc;
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const T sc_u = __shfl(sc, gbase | u), ri_u = __shfl(ri, gbase | u);
        const int iu = ib + u * stride + rg;
#pragma unroll
        for (int k = 0; k < KV; ++k) {
          const int c = rl + k * LPR;
          const V jv = av[u][k] * sc_u;
          gacc[k] += jv * ri_u;
          if (!a.own_gram && iu < m && c < nv) __builtin_nontemporal_store(jv, reinterpret_cast<V*>(Jp + size_t(iu) * n) + c);
        }
      }
    }
  }
  if (want_j) {  // fold the row groups of the wave (fixed order), then one partial vector per wave
#pragma unroll
    for (int k = 0; k < KV; ++k)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        T v = gacc[k][e];
        for (int off = LPR; off < 64; off <<= 1) v += __shfl_xor(v, off);
        gacc[k][e] = v;
      }
    if (rg == 0) {
      T* gp = a.gpart + (size_t(p) * a.gslots + slot) * n;
#pragma unroll
      for (int k = 0; k < KV; ++k) {
        const int c = rl + k * LPR;
        if (c < nv) *reinterpret_cast<V*>(gp + c * VEC) = gacc[k];
      }
    }
  }
}

// ---- H = J^T J = A^T diag(s^2) A for n > 128 on the matrix cores, hand-written (round 3).  The library GEMM computes the
// full square and needs J = diag(s) A written to and read back from HBM; here the rows kernel leaves only the row scales
// s_i, and the Gram kernel reads A ONCE per row chunk:
//   * the Gram is cut into 32 x 32 tiles (ti >= tj: the lower tile triangle, T = nt (nt + 1) / 2 of them); the rows into
//     gram_R chunks;
//   * one WORKGROUP owns (row chunk, up to 64 tiles): it stages K rows at a time — every column, scaled by s_i — in LDS (two
//     buffers: the next stage's global loads are in flight during this stage's MFMAs and are written to the other buffer
//     afterwards; one barrier per stage).  Its W waves (W = 4, 8, 12 or 16: a multiple of the four SIMDs, so that every SIMD
//     carries the same number of waves) own TPW <= 4 tiles each, dealt round-robin: one v_mfma_f32_32x32x2_f32 per tile and
//     two-row step, operands read from LDS straight into MFMA order (lane l reads column 32 t + (l & 31) of row l >> 5:
//     conflict-free ds_read_b32), the next step's operands in flight during this step's MFMAs;
//   * partial tiles go to HBM and are summed over the row chunks in fixed order by large_gram_reduce_kernel, which also
//     mirrors the upper triangle.
// (A first version gave each wave a 64 x 64 block: 10 waves at n = 256 sit 3 + 3 + 2 + 2 on the SIMDs and the barrier of
// every stage waits for the fullest — 77 TFLOP/s; profiles/r03_ab_log.md.)
// Problems that are not running or do not rebuild their Hessian this pass return at once: the host does not need their
// count.  fp32 only (fp64 keeps the library).
struct SyrkGeom {
  int nt, T;      // 32-column tiles per side, tiles in the lower triangle
  int groups;     // workgroups per (problem, row chunk)
  int per_group;  // tiles per workgroup (the last one may hold fewer)
  int waves, tpw; // waves per workgroup, tiles per wave
  int K;          // rows per LDS stage (a multiple of 8)
  int ns;         // LDS row stride in elements: 32 nt (the columns beyond n hold zeros)
};
static inline SyrkGeom syrk_geom(const int n) {
  SyrkGeom g;
  g.nt = (n + 31) / 32;
  g.T = g.nt * (g.nt + 1) / 2;
  g.groups = (g.T + 63) / 64;
  g.per_group = (g.T + g.groups - 1) / g.groups;
  g.waves = 16; g.tpw = (g.per_group + 15) / 16;
  for (int w = 12; w >= 4; w -= 4) {           // fewer slots wasted wins; on a tie the larger workgroup stays
    const int t = (g.per_group + w - 1) / w;
    if (t <= 4 && w * t < g.waves * g.tpw) { g.waves = w; g.tpw = t; }
  }
  g.ns = g.nt * 32;
  g.K = 8;                                      // rows per stage: the largest of 32 / 16 / 8 with at most three 16-byte loads per thread
  for (int K = 32; K > 8; K >>= 1)              // and 128 KB for the two buffers
    if (K * (g.ns / 4) <= 3 * g.waves * 64 && 2 * K * g.ns * 4 <= 131072) { g.K = K; break; }
  return g;
}
// LDS image of a stage: one K x 32 panel per 32-column tile (panel t at byte t K 128, row r of it at r 128): the operand of
// two-row step q of tile t is at  t K 128 + (2 q + (lane >> 5)) 128 + (lane & 31) 4  — the step enters as an IMMEDIATE
// offset of the ds_read, so the inner loop is matrix instructions and LDS reads only.  (On this chip VALU work does not
// overlap the f32 MFMAs of the same SIMD: with [row][column] staging the six address adds and six register moves per
// step cost 36 % on top of the MFMA time — profiles/r03_ab_log.md.)
template <typename T, int TPW, int NV>   // NV <= TPW: the tiles this wave really owns
__device__ __forceinline__ void syrk_stage(float __attribute__((ext_vector_type(16))) (&acc)[TPW], const unsigned char* lds, const int base,
                                           const int (&offA)[TPW], const int (&offB)[TPW], const int K) {
  if constexpr (NV > 0) {
    int pa[NV], pb[NV];
    T ra[NV], rb[NV];
#pragma unroll
    for (int s = 0; s < NV; ++s) {
      pa[s] = base + offA[s];
      pb[s] = base + offB[s];
      ra[s] = *reinterpret_cast<const T*>(lds + pa[s]);
      rb[s] = *reinterpret_cast<const T*>(lds + pb[s]);
    }
    for (int k8 = 0; k8 < K; k8 += 8) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        T na[NV], nb[NV];                       // the next step's operands are in flight during this step's MFMAs
#pragma unroll
        for (int s = 0; s < NV; ++s) {          // (q == 3 reads the first step of the next group: one panel row past the end at the
          na[s] = *reinterpret_cast<const T*>(lds + pa[s] + (q + 1) * 256);   //  very end of a stage — inside the LDS allocation, never used)
          nb[s] = *reinterpret_cast<const T*>(lds + pb[s] + (q + 1) * 256);
        }
#pragma unroll
        for (int s = 0; s < NV; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[s], rb[s], acc[s], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < NV; ++s) { ra[s] = na[s]; rb[s] = nb[s]; }
      }
#pragma unroll
      for (int s = 0; s < NV; ++s) { pa[s] += 1024; pb[s] += 1024; }
    }
  }
}
// The 36 tiles of an 8 x 8 block triangle (224 < n <= 256) dealt to 12 waves so that a wave's three tiles SHARE their operands (round 4,
// late): the kernel above reads two operands from LDS per MFMA — 96 of the LDS' 128 bytes per cycle at three waves per SIMD, which is
// what holds it at 77 % of its MFMA bound.  K8 minus the perfect matching (0,1)(2,3)(4,5)(6,7) splits into eight triangles: a wave
// with blocks x > y > z owns tiles (x,y), (x,z), (y,z) — three operands for three MFMAs; the four matched pairs a > b take (a,a), (b,b),
// (a,b) — two operands.  Still three MFMAs per wave and step on every SIMD.  Which wave computes a tile never mattered to its bits.
__device__ static constexpr unsigned kTriBlocks[12] = {0x420, 0x630, 0x750, 0x721, 0x531, 0x641, 0x652, 0x743, 0x010, 0x032, 0x054, 0x076};
template <typename T, bool PAIR, typename Mid>
__device__ __forceinline__ void syrk_stage_tri(float __attribute__((ext_vector_type(16))) (&acc)[3], const unsigned char* lds, const int base,
                                               const int o0, const int o1, const int o2, const int K, const int mid_at, Mid&& mid) {
  // `mid` (the staging of the stage after next, round 5) runs in front of the k8 == mid_at group of this wave: the three waves of a SIMD
  // take it at three different points of the stage, so that one wave's loads / scales / LDS writes fall under the others' MFMAs —
  // with every wave staging at the head of the stage, in step behind the barrier, the matrix pipe idled a third of the time
  int p0 = base + o0, p1 = base + o1, p2 = base + o2;
  if constexpr (!PAIR) {
    T r0 = *reinterpret_cast<const T*>(lds + p0), r1 = *reinterpret_cast<const T*>(lds + p1), r2 = *reinterpret_cast<const T*>(lds + p2);
    for (int k8 = 0; k8 < K; k8 += 8) {
      if (k8 == mid_at) mid();
#pragma unroll
      for (int q = 0; q < 4; ++q) {             // the next step's operands are in flight during this step's MFMAs (q == 3: see syrk_stage)
        const T n0 = *reinterpret_cast<const T*>(lds + p0 + (q + 1) * 256), n1 = *reinterpret_cast<const T*>(lds + p1 + (q + 1) * 256),
                n2 = *reinterpret_cast<const T*>(lds + p2 + (q + 1) * 256);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(r0, r1, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(r0, r2, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(r1, r2, acc[2], 0, 0, 0);
        r0 = n0; r1 = n1; r2 = n2;
      }
      p0 += 1024; p1 += 1024; p2 += 1024;
    }
  } else {
    T r0 = *reinterpret_cast<const T*>(lds + p0), r1 = *reinterpret_cast<const T*>(lds + p1);
    for (int k8 = 0; k8 < K; k8 += 8) {
      if (k8 == mid_at) mid();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const T n0 = *reinterpret_cast<const T*>(lds + p0 + (q + 1) * 256), n1 = *reinterpret_cast<const T*>(lds + p1 + (q + 1) * 256);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(r0, r0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(r1, r1, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(r0, r1, acc[2], 0, 0, 0);
        r0 = n0; r1 = n1;
      }
      p0 += 1024; p1 += 1024;
    }
  }
}
template <typename T, int TPW, bool TRI = false>
__global__ void __launch_bounds__(TRI ? 768 : 1024) large_gram_kernel(const LargeArgs<T> a, const SyrkGeom geo) {
  static_assert(!TRI || TPW == 3, "the operand-sharing deal: three tiles per wave");
  static_assert(sizeof(T) == 4, "fp32 only");
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) unsigned char syrk_lds[];
  const long long p = blockIdx.y;
  if (!a.streams(p)) return;
  if (!(a.opt.solver_type != 0 || a.st[p].rebuild)) return;
  const int n = a.n, m = a.m, ns = geo.ns, K = geo.K;
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int group = blockIdx.x % geo.groups, chunk = blockIdx.x / geo.groups;
  const int t_lo = group * geo.per_group, t_hi = min(geo.T, t_lo + geo.per_group);
  const int l31 = lane & 31, l5 = lane >> 5;
  // this wave's tiles: t_lo + wave + s W, s < TPW.  The valid ones are a prefix.  A surplus slot recomputes another tile and
  // is not stored: the barrier of every stage waits for the waves with TPW tiles anyway, so skipping it would buy nothing,
  // and a second code path costs accumulator copies and registers.
  int tile[TPW], offA[TPW], offB[TPW];
  int cnt = 0;
  bool tri_pair = false;
  if constexpr (TRI) {                          // blocks x > y > z of this wave: tiles (x,y), (x,z), (y,z); a matched pair x > y: (x,x), (y,y), (x,y)
    const unsigned code = kTriBlocks[wave];
    const int bx = int(code >> 8), by = int((code >> 4) & 15), bz = int(code & 15);
    tri_pair = wave >= 8;
    const int lo = l5 * 128 + l31 * 4;
    if (!tri_pair) {
      tile[0] = bx * (bx + 1) / 2 + by; tile[1] = bx * (bx + 1) / 2 + bz; tile[2] = by * (by + 1) / 2 + bz;
      offA[0] = bx * K * 128 + lo; offA[1] = by * K * 128 + lo; offA[2] = bz * K * 128 + lo;
    } else {
      tile[0] = by * (by + 1) / 2 + by; tile[1] = bz * (bz + 1) / 2 + bz; tile[2] = by * (by + 1) / 2 + bz;
      offA[0] = by * K * 128 + lo; offA[1] = bz * K * 128 + lo; offA[2] = offA[1];
    }
    offB[0] = offB[1] = offB[2] = 0;
    cnt = 3;
  } else {
#pragma unroll
  for (int s = 0; s < TPW; ++s) {
    int t = t_lo + wave + s * geo.waves;
    if (t < t_hi) cnt = s + 1; else t = t_lo + wave < t_hi ? t_lo + wave : t_lo;
    tile[s] = t;
    int ti = 0, rem = t;                        // t -> (ti, tj), ti >= tj, row-major over the lower triangle
    while (rem > ti) { rem -= ti + 1; ++ti; }
    offA[s] = ti * K * 128 + l5 * 128 + l31 * 4;
    offB[s] = rem * K * 128 + l5 * 128 + l31 * 4;
  }
  }
  const T* A = a.data + size_t(p) * m * (n + 1);
  const T* scp = a.sc + size_t(p) * m;
  const int row0 = chunk * a.gram_rows, row1 = min(m, row0 + a.gram_rows);
  const int buf_bytes = K * ns * 4;
  const int ns4 = ns >> 2, total4 = K * ns4;    // 16-byte elements per stage
  constexpr int NLD = 3;
  f32x4 st[NLD];
  T ssc[NLD];
  int goff[NLD], rr[NLD], loff[NLD];            // element e = tid + i nthr of a stage: row rr, global offset (from the stage's first row), LDS byte
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int e = tid + i * nthr;
    rr[i] = e / ns4;
    const int c4 = e - rr[i] * ns4;
    goff[i] = rr[i] * n + 4 * c4;
    loff[i] = e < total4 ? (c4 >> 3) * K * 128 + rr[i] * 128 + (c4 & 7) * 16 : -1;
    if (e >= total4 || 4 * c4 >= n) rr[i] = 1 << 28;   // never inside the chunk: the element stays zero
  }
  auto fetch = [&](const int r0) __attribute__((always_inline)) {   // issue only: nothing here waits for the loads
    const T* Ar = A + size_t(r0) * n;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const bool live = r0 + rr[i] < row1;
      st[i] = live ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Ar + goff[i])) : f32x4{0, 0, 0, 0};
      ssc[i] = live ? scp[r0 + rr[i]] : T(0);
    }
  };
  auto put = [&](const int base) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NLD; ++i)               // J = diag(s) A, rounded exactly as the rows kernel rounds it for the library path
      if (loff[i] >= 0) *reinterpret_cast<f32x4*>(syrk_lds + base + loff[i]) = st[i] * ssc[i];
  };
  f32x16 acc[TPW];
#pragma unroll
  for (int s = 0; s < TPW; ++s) acc[s] = f32x16{0};
  // Stage s computes on buffer s & 1.  At its START the rows of stage s + 1 (fetched during stage s - 1) are written to the
  // other buffer — free since the barrier that ended stage s - 1 — and the loads of stage s + 2 are issued: the LDS writes
  // and the global latency hide behind this stage's MFMAs, and the barrier at the end has nothing queued in front of it.
  fetch(row0);
  put(0);
  fetch(row0 + K);                              // (rows past the chunk fetch nothing and stage zeros)
  __syncthreads();
  // the stage loop and the stores of the tiles; STAGE: 0 the plain deal, 1 a triangle of blocks, 2 a matched pair.  (The two operand-sharing
  // forms are two INSTANCES of the whole loop, chosen once per wave: as two arms inside one loop hipcc keeps an accumulator set per arm.)
  auto run = [&](auto stage_c) __attribute__((always_inline)) {
    constexpr int STAGE = decltype(stage_c)::value;
    int cur = 0;
#ifdef TOA_GRAM_TIMING
    unsigned long long tg[3] = {0, 0, 0}, tgp = wall_clock64(); const unsigned long long tg0 = tgp; int nst = 0;
#define GR_TICK(i) { const unsigned long long now_ = wall_clock64(); tg[i] += now_ - tgp; tgp = now_; }
#else
#define GR_TICK(i)
#endif
    // (the waves w, w + 4, w + 8 share a SIMD: their staging goes in front of the 1st, 2nd, 3rd group of eight rows — TOA_GRAM_STAGGER 0: all at the head)
#ifndef TOA_GRAM_STAGGER
#define TOA_GRAM_STAGGER 1
#endif
    const int mid_at = (TRI && STAGE != 0 && TOA_GRAM_STAGGER && K >= 32) ? 8 * (wave >> 2) : 0;
    for (int r0 = row0; r0 < row1; r0 += K) {
      auto staging = [&]() __attribute__((always_inline)) {
        if (r0 + K < row1) {
          put(cur ? 0 : buf_bytes);
          fetch(r0 + 2 * K);
        }
      };
      GR_TICK(0)
      if constexpr (STAGE == 0) { staging(); syrk_stage<T, TPW, TPW>(acc, syrk_lds, cur ? buf_bytes : 0, offA, offB, K); }
      else if constexpr (TRI) syrk_stage_tri<T, STAGE == 2>(acc, syrk_lds, cur ? buf_bytes : 0, offA[0], offA[1], offA[2], K, mid_at, staging);
#ifdef TOA_GRAM_TIMING
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); ++nst;
#endif
      GR_TICK(1)
      __syncthreads();
      GR_TICK(2)
      cur ^= 1;
    }
#ifdef TOA_GRAM_TIMING
    if (lane == 0 && blockIdx.x == 1 && p == 3 && (wave == 0 || wave == 5 || wave == 11)) printf("gram wave %d: %d stages, %.1f us: put+fetch %.2f stage %.2f barrier %.2f us per stage\n", wave, nst, (wall_clock64() - tg0) * 0.01, tg[0] * 0.01 / nst, tg[1] * 0.01 / nst, tg[2] * 0.01 / nst);
#endif
    // C/D map of the 32 x 32 forms: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int s = 0; s < TPW; ++s) {
      if (s >= cnt) continue;
      T* blk = a.gram_part + ((size_t(p) * a.gram_R + chunk) * geo.T + tile[s]) * 1024;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) blk[((reg & 3) + 8 * (reg >> 2) + 4 * l5) * 32 + l31] = acc[s][reg];
    }
  };
  if constexpr (!TRI) run(std::integral_constant<int, 0>{});
  else if (tri_pair) run(std::integral_constant<int, 2>{});
  else run(std::integral_constant<int, 1>{});
}
template <typename T>
__global__ void __launch_bounds__(256) large_gram_reduce_kernel(const LargeArgs<T> a, const SyrkGeom geo) {
  const long long p = blockIdx.y;
  if (!a.streams(p)) return;
  if (!(a.opt.solver_type != 0 || a.st[p].rebuild)) return;
  const int n = a.n;
  const int t = blockIdx.x;
  int ti = 0, rem = t;
  while (rem > ti) { rem -= ti + 1; ++ti; }
  const int tj = rem;
  T* H = a.Hnew + size_t(p) * n * n;
  for (int e = threadIdx.x; e < 1024; e += 256) {
    const int li = e >> 5, lj = e & 31, gi = 32 * ti + li, gj = 32 * tj + lj;
    if (gi >= n || gj >= n) continue;
    T s = 0;
    // fixed order; eight loads in flight per trip (a dependent load per partial was 0.7 us each: 181 us for the 256 row chunks
    // of a single huge problem, nine times the Gram itself)
    const T* gp = a.gram_part + (size_t(p) * a.gram_R * geo.T + t) * 1024 + e;
    const size_t gstride = size_t(geo.T) * 1024;
    int r = 0;
    for (; r + 8 <= a.gram_R; r += 8) {
      T v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = gp[size_t(r + u) * gstride];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; r < a.gram_R; ++r) s += gp[size_t(r) * gstride];
    H[size_t(gi) * n + gj] = s;
    if (ti != tj) H[size_t(gj) * n + gi] = s;
  }
}

// Pointer lists for the batched GEMM: only the problems that are still running AND rebuild their Hessian this pass
// (index order: deterministic).  The host knows their count from the previous pass's read-back.
template <typename T>
__global__ void __launch_bounds__(256) large_compact_kernel(const LargeArgs<T> a, int* __restrict__ count_out = nullptr) {
  __shared__ int cnt[256];
  const int tid = threadIdx.x;
  const long long per = (a.P + 255) / 256, lo = tid * per, hi = lo + per < a.P ? lo + per : a.P;
  auto want = [&](long long p) { return a.streams(p) && (a.opt.solver_type != 0 || a.st[p].rebuild); };
  int c = 0;
  for (long long p = lo; p < hi; ++p) c += want(p) ? 1 : 0;
  cnt[tid] = c;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int t = 0; t < 256; ++t) { const int v = cnt[t]; cnt[t] = run; run += v; }
    if (count_out) *count_out = run;
  }
  __syncthreads();
  int off = cnt[tid];
  const size_t sj = size_t(a.m) * a.n, sh = size_t(a.n) * a.n;
  for (long long p = lo; p < hi; ++p)
    if (want(p)) { a.jptr[off] = a.J + p * sj; a.hptr[off] = a.Hnew + p * sh; ++off; }
}

// cost + Build (lm.h:59-120) up to the damped matrix and right-hand side
template <typename T>
__global__ void __launch_bounds__(256) large_pre_kernel(const LargeArgs<T> a) {
  __shared__ double red[256];
  __shared__ int sh_built;
  const long long p = blockIdx.x;
  if (!a.on(p)) return;
  LmState<T>& S = a.st[p];
  const toa_options& opt = a.opt;
  const int n = a.n, m = a.m, tid = threadIdx.x;
  const bool is_lm = opt.solver_type == 0;
  const bool do_acc = !is_lm || S.rebuild;
  const int skip = a.skip ? a.skip[p] : 0;   // != 0: the rows were not streamed this pass, the linearisation is read back (memo)
  double c = 0;
  const bool robust = a.loss != TOA_LOSS_L2;
  if (!skip && !robust) {
    int i = tid;
    for (; i + 7 * 256 < m; i += 8 * 256) {   // (eight loads in flight per trip, the same order of additions)
      T v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = a.r[p * m + i + u * 256];
#pragma unroll
      for (int u = 0; u < 8; ++u) c += double(v[u] * v[u]);
    }
    for (; i < m; i += 256) { const T r = a.r[p * m + i]; c += double(r * r); }
  } else if (!skip) {   // Cost += l per residual (cost.h:84-95): the rows kernel left the losses
    int i = tid;
    for (; i + 7 * 256 < m; i += 8 * 256) {
      T v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = a.lossv[p * m + i + u * 256];
#pragma unroll
      for (int u = 0; u < 8; ++u) c += double(v[u]);
    }
    for (; i < m; i += 256) c += double(a.lossv[p * m + i]);
  }
  c = block_sum<T>(c, red);
  const double cost_val = skip ? (skip == 2 ? a.memo_cost[p] : a.lin_cost[p]) : normalize_cost(double(T(c)), m, opt);
  bool built = m > 0 && cost_val != kDblMax;  // cost.h:83 isValid
  bool assigned = false;                      // H, g took the fresh accumulation (also when the diagonal check then fails the Build)
  T* g = a.g + p * n;
  T* hd = a.hd + p * n;
  if (built && do_acc && skip) {   // the linearisation of THIS x, bit for bit, is at hand (1) or parked (2: cur [p] already names its H slot)
    double low = 0;
    for (int i = tid; i < n; i += 256) {
      if (skip == 2) { g[i] = a.g_m[p * n + i]; a.hdu[p * n + i] = a.hdu_m[p * n + i]; }
      const T d = a.hdu[p * n + i];
      hd[i] = d;
      if (opt.check_min_H_diag > 0 && fabs(d) < T(opt.check_min_H_diag)) low = 1;  // lm.h:82-86
    }
    if (block_sum<T>(low, red) > 0) built = false;
  } else if (built && do_acc) {  // H, g assigned from the fresh accumulation (gn.h:77-81,109-113)
    const T* Hn = a.Hnew + size_t(p) * n * n;   // (H = Hnew itself: large_stage_kernel, all CUs instead of this one workgroup)
    double low = 0;
    for (int i = tid; i < n; i += 256) {
      T gi;
      if (a.gslots > 0) {  // fixed-order sum of the per-wave partials of the vectorised rows kernel
        gi = 0;
        const T* gq = a.gpart + size_t(p) * a.gslots * n + i;   // (eight loads in flight per trip, added in slot order)
        int sl = 0;
        for (; sl + 8 <= a.gslots; sl += 8) {
          T v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = gq[size_t(sl + u) * n];
#pragma unroll
          for (int u = 0; u < 8; ++u) gi += v[u];
        }
        for (; sl < a.gslots; ++sl) gi += gq[size_t(sl) * n];
      } else {
        gi = a.gnew[p * n + i];
      }
      if (opt.grad_clipping != 0) { const T mm = opt.grad_clipping; gi = fmin(fmax(gi, -mm), mm); }  // base.h:29-38
      g[i] = gi;
      const T d = Hn[size_t(i) * n + i];
      hd[i] = d;
      if (a.skip) a.hdu[p * n + i] = d;   // the undamped diagonal outlives the damping below
      if (opt.check_min_H_diag > 0 && fabs(d) < T(opt.check_min_H_diag)) low = 1;  // lm.h:82-86
    }
    assigned = true;
    if (block_sum<T>(low, red) > 0) built = false;
  }
  __syncthreads();
  if (built && is_lm && S.lambda > T(0)) {  // lm.h:108-117, s in double
    const double s = S.rebuild ? 1.0 + double(S.lambda) : (1.0 + double(S.lambda)) / (1.0 + double(S.prev_lambda));
    for (int i = tid; i < n; i += 256) hd[i] = T(double(hd[i]) * s);
  }
  __syncthreads();
  if (built)
    for (int i = tid; i < n; i += 256) a.rhs[p * n + i] = g[i];
  if (tid == 0) {
    if (skip) S.reused_passes++; else if (do_acc) S.acc_passes++; else S.eval_passes++;
    S.cost_val = cost_val;
    S.cost_nres = m;
    int ninl = m;
    if (robust) {
      if (skip) ninl = skip == 2 ? a.memo_ninl[p] : a.lin_ninl[p];
      else { ninl = a.ninl[p]; a.ninl[p] = 0; }   // (the next pass's rows kernel counts from zero)
    }
    S.cost_ninl = ninl;
    if (a.skip && do_acc) { a.lin_cost[p] = cost_val; a.lin_ninl[p] = ninl; S.acc_at_x = 1; S.memo_hit = 0; }
    a.built[p] = (built ? 1 : 0) | (assigned ? 2 : 0);   // bit 0: Build succeeded; bit 1: H = Hnew is due (large_stage_kernel)
  }
}
// ---- the K1 / K2 seam beyond 128 unknowns (toa_accumulate on the natural layout; SolverGN::Accumulate / Evaluate, gn.h:97-113, any
// Dims): ONE data pass of the pipeline — the rows kernel, the Gram (ours or the library's) — and this kernel instead of Build: the cost
// summed in large_pre_kernel's order, J^T r from the rows kernel's partials in slot order, H copied out.  want_grad == 0: the cost only
// (the rows kernels skip the Jacobian side when the state says "no rebuild").
template <typename T>
struct LargeSeamOut {
  T* g;           // [P][n]    (want_grad)
  T* H;           // [P][n][n] (want_grad)
  double* cost;   // [P] raw: sum of r_i^2, or of the losses l_i with toa_set_loss
  int32_t* nres;  // [P] optional
  int want_grad;
};
template <typename T>
__global__ void __launch_bounds__(256) large_seam_prep_kernel(const LargeArgs<T> a, const int want_grad) {
  const long long p = blockIdx.x * 256ll + threadIdx.x;
  if (p < a.P) a.st[p].rebuild = want_grad ? 1 : 0;
}
template <typename T>
__global__ void __launch_bounds__(256) large_seam_out_kernel(const LargeArgs<T> a, const LargeSeamOut<T> o) {
  __shared__ double red[256];
  const long long p = blockIdx.x;
  const int n = a.n, m = a.m, tid = threadIdx.x;
  const T* v = (a.loss != TOA_LOSS_L2 ? a.lossv : a.r) + size_t(p) * m;
  const bool sq = a.loss == TOA_LOSS_L2;
  double c = 0;
  for (int i = tid; i < m; i += 256) { const T r = v[i]; c += sq ? double(r * r) : double(r); }   // (large_pre_kernel's order of additions)
  c = block_sum<T>(c, red);
  if (tid == 0) {
    o.cost[p] = double(T(c));
    if (o.nres) o.nres[p] = m;
  }
  if (!o.want_grad) return;
  for (int i = tid; i < n; i += 256) {
    T gi;
    if (a.gslots > 0) {
      gi = 0;
      const T* gq = a.gpart + size_t(p) * a.gslots * n + i;
      for (int sl = 0; sl < a.gslots; ++sl) gi += gq[size_t(sl) * n];
    } else {
      gi = a.gnew[p * n + i];
    }
    o.g[p * n + i] = gi;
  }
  const T* Hn = a.Hnew + size_t(p) * n * n;
  T* Ho = o.H + size_t(p) * n * n;
  for (size_t e = tid; e < size_t(n) * n; e += 256) Ho[e] = Hn[e];
}

// The two n x n copies of Build — H = the fresh accumulation (gn.h:77-81), work = H with the damped diagonal (lm.h:108-117)
// — spread over (row slices, problems) instead of riding in large_pre_kernel's one workgroup per problem (128 workgroups on
// 256 CUs: 0.2 ms of a 2 ms pass at n = 256).  16-byte accesses when a row is a multiple of four elements.
template <typename T>
__global__ void __launch_bounds__(256) large_stage_kernel(const LargeArgs<T> a, const int rows_per_slice) {
  const long long p = blockIdx.y;
  if (!a.on(p) || !a.built[p]) return;
  const int n = a.n, tid = threadIdx.x;
  const bool do_acc = (a.built[p] & 2) != 0, built = (a.built[p] & 1) != 0;
  T* H = a.Hcur(p);
  const T* src = do_acc ? a.Hnew + size_t(p) * n * n : H;
  T* W = a.work + size_t(p) * n * n;
  const T* hd = a.hd + p * n;
  const int r0 = blockIdx.x * rows_per_slice, r1 = min(n, r0 + rows_per_slice);
  constexpr int V = 16 / int(sizeof(T));
  if (n % V == 0) {
    typedef T Vec __attribute__((ext_vector_type(V)));
    const int nv = n / V, total = (r1 - r0) * nv;
    for (int idx = tid; idx < total; idx += 256) {
      const int i = r0 + idx / nv, c = (idx % nv) * V;
      Vec v = *reinterpret_cast<const Vec*>(src + size_t(i) * n + c);
      if (do_acc) *reinterpret_cast<Vec*>(H + size_t(i) * n + c) = v;
      if (i >= c && i < c + V) v[i - c] = hd[i];
      if (built) *reinterpret_cast<Vec*>(W + size_t(i) * n + c) = v;
    }
  } else {
    const int total = (r1 - r0) * n;
    for (int idx = tid; idx < total; idx += 256) {
      const int i = r0 + idx / n, j = idx % n;
      const T v = src[size_t(i) * n + j];
      if (do_acc) H[size_t(i) * n + j] = v;
      if (built) W[size_t(i) * n + j] = (i == j) ? hd[i] : v;
    }
  }
}

// (H + lambda diag H) sol = g for 64 <= n <= 128 (the measured crossover with rocSOLVER) without the library: one workgroup
// per problem, blocked LDL^T of the LDS image by its four waves with the trailing updates on the matrix cores, substitutions
// in wave 0 (ldlt_wg.hpp; the column-by-column register Cholesky this replaces took ~100 us per n = 128 matrix against
// ~45 us).  Skips problems that have stopped or whose Build failed, which the library calls cannot.
template <typename T, int NB>
__global__ void __launch_bounds__(256) large_ldlt_solve_kernel(const LargeArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  T* M = reinterpret_cast<T*>(lds_raw);  // n x (n | 1) image, factored in place
  __shared__ T dinv[16 * NB], sol[16 * NB];
  const size_t p = blockIdx.x;
  if (!a.on(p) || !(a.built[p] & 1)) return;
  const int n = a.n, LD = n | 1, tid = threadIdx.x;
  const T* Wp = a.work + p * size_t(n) * n;
  for (int e = tid; e < n * n; e += 256) {
    const int i = e / n, j = e - i * n;
    M[i * LD + j] = Wp[e];
  }
  for (int i = tid; i < 16 * NB; i += 256) sol[i] = i < n ? a.rhs[p * n + i] : T(0);
  __syncthreads();
  const bool ok = WgLdlt<T, NB>::factor(M, LD, n, dinv, tid);
  if (!ok) {
    if (tid == 0) a.info[p] = 1;
    return;
  }
  if (tid < 64) WgLdlt<T, NB>::solve(M, LD, n, dinv, sol, tid);
  __syncthreads();
  for (int i = tid; i < n; i += 256) a.rhs[p * n + i] = sol[i];
  if (tid == 0) a.info[p] = 0;
}
// ---- (H + lambda diag H) sol = g for 128 < n <= 1024 (fp64: 512) without the library (round 3) ---------------------------------
// rocSOLVER's batched potrf + potrs is ~30 launches per solve (two potf2 kernels of 0.2 ms, ten forward and four backward
// substitution launches, syr2k, small GEMMs): 0.75 ms of a 1.5 ms pass at n = 256 and 65 % of a bundle-adjustment pass with 384
// camera unknowns — latency, not flops (n^3 / 3 = 5.6 Mflop).  Here ONE workgroup factors its matrix in place (row-major lower
// triangle of `work`, L2-resident) by panels of 32 columns and solves, in one launch for the whole batch:
//   diagonal block   wave 0, a row per lane in registers (chol_diag.hpp): 32 unrolled Cholesky columns, pivots and column entries by
//                    lane broadcast; LDS in, LDS out
//   panel below      rows through LDS with lane = column, then a thread per row: x L_kk^T = a (32-step substitution against the block
//                    in LDS); the rows stay in LDS for the update and go back to the matrix with lane = column
//   trailing update  16 x 16 tiles of the lower triangle on the matrix cores, operands from the panel in LDS, read-modify-write of the
//                    tile in the L2-resident matrix
//   substitutions    forward inside the factorisation loop; backward block by block: wave 0 the chain of a block, the others the
//                    unknowns above, operands one block ahead
// (rounds 3-5: profiles/r03_ab_log.md section 6-7, r04 section 8a, r05 section 5 — the last one with the per-wave timers of -DTOA_CHOL_TIMING)
// Cholesky without pivoting, as the library path: a pivot that is not positive (or not finite) fails the solve (info != 0).
// Every sum has a fixed order: a matrix solved alone gives the bits of its row in a batch.
// (round 4, late: v_readlane into scalar registers instead of ds_bpermute — the lane index is a compile-time constant at every call site
//  after unrolling, and a dependent 32-step chain pays the LDS crossbar's latency per step with the shuffle)
template <typename T>
__device__ __forceinline__ T chol_bcast(const T v, const int src) { return wave_bcast(v, src); }
constexpr int kCholThreads = 512;
// LOOK (round 4): look-ahead — in the trailing-update phase of block k, wave 0 updates the four tiles that hold the NEXT diagonal
// block first and factors it at once, while the other seven waves update the rest; the next step then starts at its panel.  The
// diagonal blocks were 0.17 of an n = 384 fp64 solve's 0.68 ms with seven of eight waves waiting; every tile and every pivot
// sees the arithmetic it saw before (which wave computes a tile never mattered): the same bits, checked against LOOK = false.
template <typename T, bool LOOK = true>
__global__ void __launch_bounds__(kCholThreads) large_chol_solve_kernel(const LargeArgs<T> a) {
  constexpr int B = 32, LS = B + 4, LSP = B + 4, NT = kCholThreads, NW = NT / 64;   // (LS: rows of the block 16-byte aligned — the panel reads them four / two entries per ds_read_b128)
  extern __shared__ __attribute__((aligned(16))) char chol_lds[];
  T* Ld = reinterpret_cast<T*>(chol_lds);          // [B][LS]   the factored diagonal block
  T* Lp = Ld + B * LS;                             // [n][LSP]  the panel's rows below it (row i of the matrix at Lp[i - k1]); row stride 36:
                                                   //           the MFMA operand reads (16 rows x 4 columns per instruction) hit every bank once (fp32) / twice (fp64)
  T* ys = Lp + size_t(a.n) * LSP;                  // [n]       right-hand side / solution (+ 64 scratch entries, + 32 reciprocals of the block's diagonal)
  __shared__ int fail;
  const size_t p = blockIdx.x;
  const bool direct = a.ddx != nullptr;
  if (direct) {
    if (a.dmask && a.dmask[p * a.dmask_stride] == 0) {   // a masked-out matrix is not factorised: no step, no verdict
      for (int i = threadIdx.x; i < a.n; i += NT) a.ddx[p * a.n + i] = T(0);
      if (threadIdx.x == 0) a.dok[p] = 0;
      return;
    }
  } else if (!a.on(p) || !(a.built[p] & 1)) return;
  const int n = a.n, wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);   // (uniform for hipcc: tile indices, trip counts and the wave-0 branches live in scalar registers)
  int tid = threadIdx.x, lane = tid & 63;   // (not const: see TOA_CHOL_FRESH_LANE)
  // hipcc hoists every per-lane invariant of the block loop's phases (LDS addresses, masks like c == lane, row offsets: hundreds of
  // values) in front of the loop and then spills them — scratch reloads with an s_waitcnt in front of loads that should be in flight
  // together.  The lane index is re-declared opaque at the head of each phase: what depends on it is computed where it is used.
#define TOA_CHOL_FRESH_LANE asm volatile("" : "+v"(lane), "+v"(tid));
  T* A = a.work + p * size_t(n) * n;
  for (int i = tid; i < n; i += NT) ys[i] = a.rhs[p * n + i];
  if (tid == 0) fail = 0;
  __syncthreads();
#ifdef TOA_CHOL_TIMING
  __shared__ unsigned long long wt[8], wt0[8];
  if (tid < 8) wt[tid] = wt0[tid] = 0;
  const unsigned long long clk0 = clock64(), wall0 = wall_clock64();
  unsigned long long tkc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = wall_clock64(), tsub = tprev;
#define CH_TICK(i) { const unsigned long long now_ = wall_clock64(); tkc[i] += now_ - tprev; tprev = now_; tsub = now_; }
#define CH_SUB(i) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = wall_clock64(); tkc[i] += now_ - tsub; tsub = now_; }
#else
#define CH_TICK(i)
#define CH_SUB(i)
#endif
  // the diagonal block at k0 (wave 0): chol_diag_block on the copy in Ld.  `staged`: the look-ahead left the updated entries there —
  // no store / fence / reload through the matrix (round 5); otherwise they come from the matrix first, lane = column.  The factored
  // block reaches the matrix from the last wave during the panel phase.
  auto diag_block = [&](const int k0, const int bs, const bool staged) __attribute__((always_inline)) {
    if (!staged) {
      const int col = lane & 31;
      T in[B / 2];
#pragma unroll
      for (int q = 0; q < B / 2; ++q) {
        const int row = 2 * q + (lane >> 5);
        in[q] = (row < bs && col <= row) ? ld_at(A, unsigned((k0 + row) * n + k0 + col)) : T(0);
      }
#pragma unroll
      for (int q = 0; q < B / 2; ++q) {
        const int row = 2 * q + (lane >> 5);
        if (row < bs && col <= row) Ld[row * LS + col] = in[q];
      }
      __builtin_amdgcn_wave_barrier();
    }
#ifdef TOA_CHOL_TIMING
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long td0 = wall_clock64();
#endif
    using L3 = __attribute__((address_space(3))) T*;
    chol_diag_block<T, LS>((L3)Ld, (L3)(ys + n + 64), (__attribute__((address_space(3))) int*)&fail, bs);
#ifdef TOA_CHOL_TIMING
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long td2 = wall_clock64();
    tkc[5] += td0 - tsub; tkc[6] += td2 - td0; tsub = td2;
#endif
  };
  bool have_diag = false;   // LOOK: the block was factored during the previous step's trailing update
  for (int k0 = 0; k0 < n; k0 += B) {
    const int bs = min(B, n - k0), k1 = k0 + bs;
    TOA_CHOL_FRESH_LANE
    if (!have_diag) {
      if (wave == 0) diag_block(k0, bs, false);
      __syncthreads();
    }
    CH_TICK(0)
    if (fail) break;
    // LOOK: the forward substitution L y = b rides along — block k's unknowns by the LAST wave here (the arithmetic of the
    // separate pass below, with the block read from LDS instead of the matrix), the rows below at the start of the update
    // phase from the panel in LDS: the pass over L in the L2-resident matrix (0.09 of an n = 384 fp64 solve's 0.68 ms) is gone
    if constexpr (LOOK) {
      if (wave == NW - 1) {
        T y = lane < bs ? ys[k0 + lane] : T(0);
        T lr[B];
#pragma unroll
        for (int c = 0; c < B; ++c) lr[c] = (lane < bs && c <= lane) ? Ld[lane * LS + c] : T(1);
        T rinv = T(1);
#pragma unroll
        for (int c = 0; c < B; ++c) rinv = (c == lane && lane < bs) ? T(1) / lr[c] : rinv;
#pragma unroll
        for (int c = 0; c < B; ++c) {
          const T yc = chol_bcast(y * rinv, c);
          if (lane == c) y = yc;
          else if (lane > c) y = fma(-lr[c], yc, y);
        }
        if (lane < bs) ys[k0 + lane] = y;
      }
    }
    // ---- the rows below: x L_kk^T = a, a thread per row; the result to the matrix (L) and to LDS (for the update).  Round 5:
    //  * a thread per row read its 32 entries with 32 loads that touch 64 cache lines each and stored them the same way; now each
    //    wave moves its 64 rows between the matrix and the panel in LDS with lane = column (two rows, four lines per instruction)
    //  * the block's rows sit 16-byte aligned in LDS (stride 36): row c's entries come four (fp32) / two (fp64) per ds_read_b128 — the
    //    phase was bound by the NUMBER of broadcast reads (496 per row and wave), not by their bytes; the right-looking order
    //    (independent FMAs, strided reads of one entry each) measured 50 % slower for that reason
    //  * every difference / product is in a register before the first store of a group: hipcc put an s_waitcnt vmcnt(0) in front
    //    of each predicated store whose value was loaded — each store waiting for the previous one's acknowledgement
    // (rows below exist only under a full block: bs == B here)
    if (wave == NW - 1) {   // the factored block goes to the matrix from here, off wave 0's path (the substitutions read it there)
      const int col = lane & 31;
#pragma unroll
      for (int q = 0; q < B / 2; ++q) {
        const int row = 2 * q + (lane >> 5);
        if (row < bs && col <= row) st_at(A, unsigned((k0 + row) * n + k0 + col), Ld[row * LS + col]);
      }
    }
    for (int rb = k1 + 64 * wave; rb < n; rb += NT) {
      {
        T in[B];
        const int col = lane & 31;
#pragma unroll
        for (int q = 0; q < B; ++q) {
          const int row = min(rb + 2 * q + (lane >> 5), n - 1);
          in[q] = ld_at(A, unsigned(row * n + k0 + col));
        }
#pragma unroll
        for (int q = 0; q < B; ++q) {
          const int row = rb + 2 * q + (lane >> 5);
          if (row < n) Lp[(row - k1) * LSP + col] = in[q];
        }
      }
      __builtin_amdgcn_wave_barrier();
      CH_SUB(8)
      const int i = rb + lane;
      if (i < n) {
        T x[B];
#pragma unroll
        for (int c = 0; c < B; ++c) x[c] = Lp[(i - k1) * LSP + c];
#pragma unroll
        for (int c = 0; c < B; ++c) {
          asm volatile("" ::: "memory");   // (the block's entries are read where they are used: hoisted out of the row loop they are 528 registers)
          T v = x[c];
#pragma unroll
          for (int t = 0; t < c; ++t) v = fma(-x[t], Ld[c * LS + t], v);
          x[c] = v * ys[n + 64 + c];
        }
#pragma unroll
        for (int c = 0; c < B; ++c) Lp[(i - k1) * LSP + c] = x[c];
      }
      __builtin_amdgcn_wave_barrier();
      CH_SUB(9)
      {
        const int col = lane & 31;
        T out[B];
#pragma unroll
        for (int q = 0; q < B; ++q) out[q] = Lp[(min(rb + 2 * q + (lane >> 5), n - 1) - k1) * LSP + col];
#pragma unroll
        for (int q = 0; q < B; ++q) {
          const int row = rb + 2 * q + (lane >> 5);
          if (row < n) st_at(A, unsigned(row * n + k0 + col), out[q]);
        }
      }
      CH_SUB(10)
    }
    // the trailing block at k1: r rows, nt x nt tiles of 16 x 16, the lower triangle's ntile tiles numbered row-major.  A tile's four
    // values per lane: element (out_row(lane, reg), lane & 15) of tile (ti, tj) = Ak[16 ti n + 16 tj + lane_off + reg * reg_step]
    TOA_CHOL_FRESH_LANE
    using Acc = typename Mfma<T>::Acc;
    constexpr int U = sizeof(T) == 8 ? 2 : 4;          // tiles per trip (fp64, four: 211 -> 227 us in round 3; 341.7 -> 352.0 us per call on the finished kernel of round 5)
    const int r = n - k1, nt = (r + 15) >> 4, ntile = nt * (nt + 1) / 2;
    const int l15 = lane & 15, kq = lane >> 4;
    T* const Ak = A + size_t(k1) * n + k1;
    const int lane_off = Mfma<T>::out_row(lane, 0) * n + l15;
    const int reg_step = (Mfma<T>::out_row(0, 1) - Mfma<T>::out_row(0, 0)) * n;
    auto tile_full = [&](const int ti, const int tj) __attribute__((always_inline)) { return ti != tj && 16 * ti + 16 <= r; };
    const bool look = LOOK && k1 < n;   // there is a next diagonal block: tiles 0 .. 2 of the trailing triangle (3 rides along)
    T look_old[4][4];                   // wave 0: their old values (final since the last barrier), asked for at the end of the panel phase
    // (on their way across the barrier, the y update and the products; held across the whole panel phase they cost 32 registers
    //  beside a row's 64 and the kernel spilled)
    if (look && wave == 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int gi = 16 * (u == 0 ? 0 : u == 3 ? 2 : 1) + Mfma<T>::out_row(lane, reg), gj = 16 * (u == 2 ? 1 : 0) + l15;
          look_old[u][reg] = (gi < r && gj <= gi) ? ld_at(Ak, unsigned(gi * n + gj)) : T(0);
        }
    }
    __syncthreads();
    CH_TICK(1)
    // ---- trailing update A22 -= L21 L21^T on the matrix cores: 16 x 16 tiles of the lower triangle dealt to the waves, both
    // operands straight from the panel in LDS (row stride 36: conflict-free), eight v_mfma_*_16x16x4 per tile, then a
    // read-modify-write of the tile in the L2-resident matrix.  (A lane per column with the rows in batches of eight — VALU, one
    // dependent global round trip per batch — took 122 us of a 376 us n = 256 solve.)
    {
      if constexpr (LOOK) {   // y_i -= L_i,k . y_k for the rows below, from the panel in LDS (same order of FMAs as the separate pass)
        for (int i = k1 + tid; i < n; i += NT) {
          T sy = ys[i];
          const T* Li = Lp + (i - k1) * LSP;
#pragma unroll
          for (int c = 0; c < B; ++c) sy = fma(-Li[c], ys[k0 + c], sy);
          ys[i] = sy;
        }
      }
      // the products of N tiles: the tiles take turns on the matrix pipe, the operands of the next step on their way from LDS during
      // the MFMAs of this one (round 5; tile after tile the 8 MFMAs of a tile were one dependent chain with an exposed LDS round
      // trip per pair, and an fp64 16x16x4 occupies the pipe for 64 cycles on this part).  Each tile still sums q = 0 .. 7 in order.
      auto products = [&](const auto& ti, const auto& tj, auto& acc) __attribute__((always_inline)) {
        constexpr int N = int(sizeof(ti) / sizeof(ti[0]));
#pragma unroll
        for (int u = 0; u < N; ++u) acc[u] = Acc{0, 0, 0, 0};
        const T* pa[N];
        const T* pb[N];
#pragma unroll
        for (int u = 0; u < N; ++u) {
          pa[u] = Lp + (16 * ti[u] + l15) * LSP + kq;   // (rows past the trailing block hold stale panel rows: their products are not stored)
          pb[u] = Lp + (16 * tj[u] + l15) * LSP + kq;
        }
#pragma unroll
        for (int q = 0; q < B / 4; ++q)
#pragma unroll
          for (int u = 0; u < N; ++u) acc[u] = Mfma<T>::fma(pa[u][4 * q], pb[u][4 * q], acc[u]);
        // (hipcc's own wait states for the builtin are enough for the hardware; tools/isa_lint.py asks for the 16-pass margin)
        if constexpr (N == 4) asm volatile("s_nop 9" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]));
        else asm volatile("s_nop 9" : "+a"(acc[0]), "+a"(acc[1]));
      };
      auto store_res = [&](const int t, const int step, const int (&ti)[U], const int (&tj)[U], const T (&res)[U][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (t + u * step < ntile) {
            T* dst = Ak + (16 * ti[u] * n + 16 * tj[u]);
            if (tile_full(ti[u], tj[u])) {
#pragma unroll
              for (int reg = 0; reg < 4; ++reg) st_at(dst, unsigned(lane_off + reg * reg_step), res[u][reg]);
            } else {
#pragma unroll
              for (int reg = 0; reg < 4; ++reg) {
                const int gi = 16 * ti[u] + Mfma<T>::out_row(lane, reg), gj = 16 * tj[u] + l15;
                if (gi < r && gj <= gi) st_at(dst, unsigned(lane_off + reg * reg_step), res[u][reg]);
              }
            }
          }
        }
      };
      // trips t, t + U step, ... of this wave.  What round 5 found in a trip (n = 384 fp64, 1.9 us per two tiles, 0.4 of them MFMAs):
      //  * `A = old - acc` inside each predicated store: hipcc put an s_waitcnt vmcnt(0) in front of every store (a loaded register
      //    first read after a branch join), each store waiting for the previous one's acknowledgement — every difference is in
      //    a register before the first store now
      //  * t -> (ti, tj) as a loop from zero, in VECTOR registers with the wave index = threadIdx >> 6 (0.7 us per trip): the wave
      //    index is uniform by readfirstlane, the pair is carried from trip to trip
      //  * per-lane masks and 64-bit index arithmetic per access: tiles off the diagonal that end inside the matrix go without
      // (the trip's old values loaded one trip ahead, two register sets: built, measured, no change — dropped)
      auto sweep = [&](int t, const int step) __attribute__((always_inline)) {
        const int stride = U * step;
        int ti[U], tj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {                    // t -> (ti, tj), ti >= tj, row-major over the lower triangle
          int row = 0, rem = t + u * step;
          while (rem > row) { rem -= row + 1; ++row; }
          ti[u] = row; tj[u] = rem;
        }
        for (; t < ntile; t += stride) {
          T old[U][4];
#pragma unroll
          for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) old[u][reg] = T(0);
            if (t + u * step < ntile) {
              const T* src = Ak + (16 * ti[u] * n + 16 * tj[u]);
              if (tile_full(ti[u], tj[u])) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) old[u][reg] = ld_at(src, unsigned(lane_off + reg * reg_step));
              } else {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                  const int gi = 16 * ti[u] + Mfma<T>::out_row(lane, reg), gj = 16 * tj[u] + l15;
                  if (gi < r && gj <= gi) old[u][reg] = ld_at(src, unsigned(lane_off + reg * reg_step));
                }
              }
            }
          }
          Acc acc[U];
          products(ti, tj, acc);
          T res[U][4];
#pragma unroll
          for (int u = 0; u < U; ++u)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) { res[u][reg] = old[u][reg] - acc[u][reg]; asm volatile("" : "+v"(res[u][reg])); }
          store_res(t, step, ti, tj, res);
#pragma unroll
          for (int u = 0; u < U; ++u) {                  // the same tiles of the next trip
            tj[u] += stride;
            while (tj[u] > ti[u]) { tj[u] -= ti[u] + 1; ++ti[u]; }
          }
        }
      };
#ifdef TOA_CHOL_TIMING
      const unsigned long long tw0 = wall_clock64();
#endif
      if (!look) {
        sweep(wave, NW);
      } else if (wave == 0) {
        // look-ahead: the four tiles that hold the next diagonal block (their old values came in before the panel phase), the block
        // staged in Ld — no store / fence / reload through the matrix — and factored at once; tile 3 = (2, 0) rides along
        const int lti[4] = {0, 1, 1, 2}, ltj[4] = {0, 0, 1, 0};
        Acc acc[4];
        products(lti, ltj, acc);
        T res[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) { res[u][reg] = look_old[u][reg] - acc[u][reg]; asm volatile("" : "+v"(res[u][reg])); }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int gi = 16 * lti[u] + Mfma<T>::out_row(lane, reg), gj = 16 * ltj[u] + l15;
            if (u < 3) { if (gi < r && gj <= gi) Ld[gi * LS + gj] = res[u][reg]; }
            else if (gi < r) st_at(Ak, unsigned(gi * n + gj), res[u][reg]);
          }
        __builtin_amdgcn_wave_barrier();
        diag_block(k1, min(B, n - k1), true);
      } else {
        sweep(4 + (wave - 1), NW - 1);
      }
      have_diag = look;
#ifdef TOA_CHOL_TIMING
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      if (lane == 0) { wt[wave] += wall_clock64() - tw0; if (k0 == 0) wt0[wave] = wall_clock64() - tw0; }
#endif
    }
    __syncthreads();
    CH_TICK(2)
  }
  if (fail) {
    if (direct) {
      for (int i = tid; i < n; i += NT) a.ddx[p * n + i] = T(0);
      if (tid == 0) a.dok[p] = 0;
    } else if (tid == 0) a.info[p] = 1;
    return;
  }
  // ---- L y = b (LOOK: done inside the factorisation loop)
  if constexpr (!LOOK)
  for (int k0 = 0; k0 < n; k0 += B) {
    const int bs = min(B, n - k0), k1 = k0 + bs;
    if (wave == 0) {
      T y = lane < bs ? ys[k0 + lane] : T(0);
      T lr[B];
#pragma unroll
      for (int c = 0; c < B; ++c) lr[c] = (lane < bs && c <= lane) ? A[size_t(k0 + lane) * n + k0 + c] : T(1);
      T rinv = T(1);                                     // 1 / L_rr of this lane's row: the division leaves the dependent chain
#pragma unroll
      for (int c = 0; c < B; ++c) rinv = (c == lane && lane < bs) ? T(1) / lr[c] : rinv;
#pragma unroll
      for (int c = 0; c < B; ++c) {
        const T yc = chol_bcast(y * rinv, c);            // (lane c: its unknown; the value of the others is unused)
        if (lane == c) y = yc;
        else if (lane > c) y = fma(-lr[c], yc, y);
      }
      if (lane < bs) ys[k0 + lane] = y;
    }
    __syncthreads();
    for (int i = k1 + tid; i < n; i += NT) {
      T s = ys[i];
      const T* Li = A + size_t(i) * n + k0;
      T lv[B];
#pragma unroll
      for (int c = 0; c < B; ++c) lv[c] = c < bs ? Li[c] : T(0);   // (all 32 loads in flight)
#pragma unroll
      for (int c = 0; c < B; ++c) s = fma(-lv[c], ys[k0 + (c < bs ? c : 0)], s);
      ys[i] = s;
    }
    __syncthreads();
  }
  CH_TICK(3)
  // ---- L^T x = y
  auto back_block = [&](const int k0, const int bs) __attribute__((always_inline)) {   // wave 0: the 32 unknowns of block k0
    T xv = lane < bs ? ys[k0 + lane] : T(0);
    T lcol[B];                                         // column `lane` of L_kk: entries (c, lane), c >= lane (all loads in flight,
#pragma unroll                                         //  coalesced across lanes; inside the loop below they were 64 dependent round trips)
    for (int c = 0; c < B; ++c) lcol[c] = (lane < bs && c < bs && c >= lane) ? A[size_t(k0 + c) * n + k0 + lane] : T(0);
    T ldiag = T(1);
#pragma unroll
    for (int c = 0; c < B; ++c) ldiag = (c == lane && c < bs) ? lcol[c] : ldiag;
    ldiag = T(1) / ldiag;
#pragma unroll
    for (int c = B - 1; c >= 0; --c) {
      const T xc = chol_bcast(xv * ldiag, c);          // lane c: its unknown (for c >= bs: 0)
      if (lane == c) xv = xc;
      else if (lane < c) xv = fma(-lcol[c], xc, xv);   // L_c,lane: the entry of L^T this lane's equation holds for unknown c
    }
    if (lane < bs) ys[k0 + lane] = xv;
  };
  auto back_update = [&](const int j, const int k0, const int bs) __attribute__((always_inline)) {   // x_j -= sum_c L_(k0+c),j x_(k0+c)
    T sx = ys[j];
    T lv[B];
#pragma unroll
    for (int c = 0; c < B; ++c) lv[c] = c < bs ? A[size_t(k0 + c) * n + j] : T(0);   // (coalesced across lanes)
#pragma unroll
    for (int c = 0; c < B; ++c) sx = fma(-lv[c], ys[k0 + (c < bs ? c : 0)], sx);
    ys[j] = sx;
  };
  if constexpr (LOOK) {
    // look-ahead again: once block k is solved, wave 0 updates the 32 unknowns of the block ABOVE it and solves that block at
    // once, while the other waves update the unknowns further up — one barrier per block instead of two, and the serial
    // 32-step chain of a block runs beside the updates of the previous one.  Every unknown sees the same FMAs in the same order.
    // Round 5: wave 0's two operand blocks (the diagonal block above and the 32 x 32 block between) are in LDS when it gets there —
    // wave 1 fetches them one block ahead into the idle panel buffer (two parities) — instead of two dependent L2 round trips per
    // block in front of the chain: 4.7 -> ~2 us per block.
    const int klast = ((n - 1) / B) * B;
    auto stage_blocks = [&](const int k0, const int bs, const int par) __attribute__((always_inline)) {   // one wave: what takes wave 0 from block k0 to block k0 - B
      const int kn = k0 - B, col = lane & 31;
      T* sd = Lp + 2 * par * B * LSP;
      T* sb = sd + B * LSP;
      T v1[B / 2], v2[B / 2];
#pragma unroll
      for (int q = 0; q < B / 2; ++q) {
        const int row = 2 * q + (lane >> 5);
        v1[q] = row >= col ? ld_at(A, unsigned((kn + row) * n + kn + col)) : T(0);
        v2[q] = row < bs ? ld_at(A, unsigned((k0 + row) * n + kn + col)) : T(0);
      }
#pragma unroll
      for (int q = 0; q < B / 2; ++q) {
        const int row = 2 * q + (lane >> 5);
        sd[row * LSP + col] = v1[q];
        sb[row * LSP + col] = v2[q];
      }
    };
    // the chain of one block: xv = the right-hand side of unknown `lane` with every block below already taken off
    auto solve_block = [&](const int kb, const int bsb, T xv, const T (&lcol)[B]) __attribute__((always_inline)) {
      T ldiag = T(1);
#pragma unroll
      for (int c = 0; c < B; ++c) ldiag = (c == lane && c < bsb) ? lcol[c] : ldiag;
      ldiag = T(1) / ldiag;
#pragma unroll
      for (int c = B - 1; c >= 0; --c) {
        const T xc = chol_bcast(xv * ldiag, c);          // lane c: its unknown (for c >= bs: 0)
        if (lane == c) xv = xc;
        else if (lane < c) xv = fma(-lcol[c], xc, xv);   // L_c,lane: the entry of L^T this lane's equation holds for unknown c
      }
      if (lane < bsb) ys[kb + lane] = xv;
    };
    // the other waves' unknown j0 (first trip): its 32 entries of the NEXT block's rows are loaded one block ahead as well
    const int j0 = tid - 64;
    T lvN[B];
    auto load_rows = [&](const int k0, const int bs, T (&lv)[B]) __attribute__((always_inline)) {
#pragma unroll
      for (int c = 0; c < B; ++c) lv[c] = (c < bs && j0 < k0 - B) ? ld_at(A, unsigned((k0 + c) * n + j0)) : T(0);   // (coalesced across lanes)
    };
    if (wave == 0) back_block(klast, min(B, n - klast));
    else {
      if (wave == 1 && klast > 0) stage_blocks(klast, min(B, n - klast), 0);
      load_rows(klast, min(B, n - klast), lvN);
    }
    __syncthreads();
    int par = 0;
    for (int k0 = klast; k0 > 0; k0 -= B, par ^= 1) {
      const int bs = min(B, n - k0), kn = k0 - B;      // (every block above the last one is full)
      if (wave == 0) {
        const T* sd = Lp + 2 * par * B * LSP;
        const T* sb = sd + B * LSP;
        const int l31 = lane & 31;
        T lcol[B], lv[B];
#pragma unroll
        for (int c = 0; c < B; ++c) { lcol[c] = (lane < B && c >= lane) ? sd[c * LSP + l31] : T(0); lv[c] = lane < B ? sb[c * LSP + l31] : T(0); }
        T sx = lane < B ? ys[kn + lane] : T(0);
#pragma unroll
        for (int c = 0; c < B; ++c) sx = fma(-lv[c], ys[k0 + (c < bs ? c : 0)], sx);
        solve_block(kn, B, sx, lcol);
      } else {
        T lv[B];
#pragma unroll
        for (int c = 0; c < B; ++c) lv[c] = lvN[c];
        if (wave == 1 && kn > 0) stage_blocks(kn, B, par ^ 1);
        if (kn > 0) load_rows(kn, B, lvN);
        if (j0 < kn) {
          T sx = ys[j0];
#pragma unroll
          for (int c = 0; c < B; ++c) sx = fma(-lv[c], ys[k0 + (c < bs ? c : 0)], sx);
          ys[j0] = sx;
        }
        for (int j = j0 + NT - 64; j < kn; j += NT - 64) back_update(j, k0, bs);
      }
      __syncthreads();
    }
  } else {
    for (int k0 = ((n - 1) / B) * B; k0 >= 0; k0 -= B) {
      const int bs = min(B, n - k0);
      if (wave == 0) back_block(k0, bs);
      __syncthreads();
      for (int j = tid; j < k0; j += NT) back_update(j, k0, bs);   // the unknowns above
      __syncthreads();
    }
  }
  CH_TICK(4)
#ifdef TOA_CHOL_TIMING
  if (tid == 0 && p == 0) printf("chol n=%d: diag %.1f us  panel %.1f us  update %.1f us  forward %.1f us  backward %.1f us | wave 0 (look-ahead): tiles + staging %.1f diagonal block %.1f | panel: in %.1f rows %.1f out %.1f\n", n, tkc[0] * 0.01, tkc[1] * 0.01, tkc[2] * 0.01, tkc[3] * 0.01, tkc[4] * 0.01,
                              tkc[5] * 0.01, tkc[6] * 0.01, tkc[8] * 0.01, tkc[9] * 0.01, tkc[10] * 0.01);
  if (tid == 0 && p == 0) printf("chol n=%d: shader clock %.0f MHz over the kernel\n", n, double(clock64() - clk0) / (double(wall_clock64() - wall0) * 0.01));
  if (tid == 0 && p == 0) printf("chol n=%d: update phase by wave, all steps: %.1f %.1f %.1f %.1f %.1f %.1f %.1f %.1f | step 0: %.1f %.1f %.1f %.1f\n", n, wt[0] * 0.01, wt[1] * 0.01, wt[2] * 0.01, wt[3] * 0.01, wt[4] * 0.01, wt[5] * 0.01, wt[6] * 0.01, wt[7] * 0.01,
                              wt0[0] * 0.01, wt0[1] * 0.01, wt0[4] * 0.01, wt0[7] * 0.01);
#endif
  if (direct) {   // large_finish_kernel's part: a solution that is not finite is refused
    for (int i = tid; i < n; i += NT)
      if (!(fabs(ys[i]) <= NumLimits<T>::max())) fail = 1;   // (benign race: every writer stores 1)
    __syncthreads();
    const bool good = fail == 0;
    for (int i = tid; i < n; i += NT) a.ddx[p * n + i] = good ? -ys[i] : T(0);
    if (tid == 0) a.dok[p] = good ? 1 : 0;
    return;
  }
  for (int i = tid; i < n; i += NT) a.rhs[p * n + i] = ys[i];
  if (tid == 0) a.info[p] = 0;
}
template <typename T>
inline size_t chol_solve_lds_bytes(int n) { return (size_t(32) * 36 + size_t(n) * 36 + size_t(n) + 64 + 32) * sizeof(T) + 64; }

template <typename T>
inline size_t ldlt_image_bytes(int n) { return ((size_t(n) * (n | 1) + 16) * sizeof(T) + 15) & ~size_t(15); }
template <typename T>
const void* ldlt_solve_fn(int n) {
  switch ((n + 15) / 16) {
    case 1: case 2: case 3: case 4: return (const void*)large_ldlt_solve_kernel<T, 4>;
    case 5: return (const void*)large_ldlt_solve_kernel<T, 5>;
    case 6: return (const void*)large_ldlt_solve_kernel<T, 6>;
    case 7: return (const void*)large_ldlt_solve_kernel<T, 7>;
    default: return (const void*)large_ldlt_solve_kernel<T, 8>;
  }
}
template <typename T>
void launch_ldlt_solve(int n, unsigned P, size_t lds, hipStream_t st, const LargeArgs<T>& a) {
  switch ((n + 15) / 16) {
    case 1: case 2: case 3: case 4: hipLaunchKernelGGL((large_ldlt_solve_kernel<T, 4>), dim3(P), dim3(256), lds, st, a); break;
    case 5: hipLaunchKernelGGL((large_ldlt_solve_kernel<T, 5>), dim3(P), dim3(256), lds, st, a); break;
    case 6: hipLaunchKernelGGL((large_ldlt_solve_kernel<T, 6>), dim3(P), dim3(256), lds, st, a); break;
    case 7: hipLaunchKernelGGL((large_ldlt_solve_kernel<T, 7>), dim3(P), dim3(256), lds, st, a); break;
    default: hipLaunchKernelGGL((large_ldlt_solve_kernel<T, 8>), dim3(P), dim3(256), lds, st, a); break;
  }
}

// optimizer.h:313-327: problem p is done — the undamped Hessian, the results row, the counters (a whole 256-thread workgroup)
template <typename T>
__device__ void large_finish_problem(const LargeArgs<T>& a, const long long p) {
  LmState<T>& S = a.st[p];
  const toa_options& opt = a.opt;
  const toa_results& res = a.res;
  const int n = a.n, tid = threadIdx.x;
  if (opt.save_last && res.final_hessian) {  // undamped (lm.h:157-171)
    double* Hout = res.final_hessian + size_t(p) * n * n;
    const T* H = a.Hcur(p);
    const T* hd = a.hd + p * n;
    for (size_t e = tid; e < size_t(n) * n; e += 256) {
      const int i = int(e / n), j = int(e % n);
      T v = H[e];
      if (i == j) { v = hd[i]; if (opt.solver_type == 0 && S.prev_lambda > T(0)) v = v / (T(1.0f) + S.prev_lambda); }
      Hout[e] = double(v);
    }
  }
  if (tid == 0) {
    if (S.stop == TOA_STOP_NONE && S.num_iters >= S.max_iters) S.stop = TOA_STOP_MAX_ITERS;  // :320-321
    res.stop_reason[p] = S.stop;
    res.num_iters[p] = S.num_iters;
    res.final_cost[p] = S.final_cost;
    if (res.num_failures) res.num_failures[p] = int(S.num_failures);
    if (res.num_consec_failures) res.num_consec_failures[p] = int(S.num_consec);
    if (res.final_num_residuals) res.final_num_residuals[p] = S.final_nres;
    if (res.final_rerr_dec) res.final_rerr_dec[p] = S.final_rerr;
    if (res.final_inlier_ratio) res.final_inlier_ratio[p] = S.final_nres > 0 ? float(S.final_ninl) / float(S.final_nres) : 1.0f;
    a.active[p] = 0;
    if (a.counters) {
      atomicAdd(&a.counters[0], S.acc_passes);
      atomicAdd(&a.counters[1], S.eval_passes);
      atomicAdd(&a.counters[2], S.solves);
      atomicAdd(&a.counters[3], 1ull);
      if (S.reused_passes) atomicAdd(&a.counters[4], S.reused_passes);
    }
  }
}

// the rest of Step and of the loop body; finalisation of problems that stop
template <typename T>
__global__ void __launch_bounds__(256) large_post_kernel(const LargeArgs<T> a, int* __restrict__ summary) {
  __shared__ double red[256];
  __shared__ int sh_action, sh_cont, sh_retry, sh_park, sh_check;
  const long long p = blockIdx.x;
  if (!a.on(p)) return;
  LmState<T>& S = a.st[p];
  const bool memo = a.skip != nullptr;
  const toa_options& opt = a.opt;
  const toa_results& res = a.res;
  const int n = a.n, tid = threadIdx.x;
  T* x = a.x + p * n;
  T* g = a.g + p * n;
  T* dx = a.dx + p * n;
  T* ldx = a.ldx + p * n;
  const bool built = (a.built[p] & 1) != 0;
  bool solver_failed = true;
  double dx_norm2 = 0, grad_norm2 = 0;
  if (built) {  // gn.h:150-171
    double bad = a.info[p] != 0 ? 1.0 : 0.0, d2 = 0, g2 = 0;
    for (int i = tid; i < n; i += 256) {
      const T v = -a.rhs[p * n + i];
      if (!(fabs(v) <= NumLimits<T>::max())) bad = 1.0;
      dx[i] = v;
      d2 += double(v * v);
      g2 += double(g[i] * g[i]);
    }
    bad = block_sum<T>(bad, red);
    dx_norm2 = double(T(block_sum<T>(d2, red)));
    if (opt.min_grad_norm2 > 0.0f) grad_norm2 = double(T(block_sum<T>(g2, red)));
    solver_failed = bad > 0;
  }
  if (tid == 0) {
    const unsigned max_tries = opt.max_consec_failures > 0 ? (opt.max_consec_failures > 1 ? opt.max_consec_failures : 1) : 255;
    int rc;  // 0 step, 1 solver failed for good, 2 early stop, -1 retry (same iteration, next pass)
    if (built) S.solves++;
    if (!solver_failed) {
      rc = 0;
    } else {  // optimizer.h:370-390
      S.num_consec = (S.num_consec + 1) & 0xff;
      S.num_failures = (S.num_failures + 1) & 0xff;
      if (S.cost_nres == 0) { S.stop = TOA_STOP_SKIPPED; rc = 2; }
      else if (isnan(S.cost_val) || isinf(S.cost_val)) { S.stop = TOA_STOP_NAN_OR_INF; rc = 2; }
      else if (opt.max_consec_failures > 0 && S.num_consec >= unsigned(opt.max_consec_failures)) {
        if (S.final_cost < double(NumLimits<T>::max())) S.stop = TOA_STOP_MAX_CONSEC_NO_DECR;
        rc = 1;
      } else {
        lm_bad_step(S, opt);  // FailedStep == BadStep  lm.h:148
        rc = (S.num_consec <= max_tries) ? -1 : 1;
      }
    }
    int action = 0;  // 1: x += dx, last_dx = dx ; 2: x -= last_dx
    int cont = 1;
    int park = 0, check = 0;
    if (rc >= 0) {
      int status = 0;
      if (rc == 1) S.stop = TOA_STOP_SOLVER_FAILED;  // :396-399
      if (rc == 0) status = lm_judge_core<T>(S, opt, res, p, dx_norm2, grad_norm2, true);
      bool eval_only = false;  // optimizer.h:269-309
      S.memo_hit = 0;           // (only the roll-back below may arm it, for the Build that follows directly)
      if (status & 1) {
        // x is about to leave an ACCEPTED point: park its linearisation (the slot it lives in simply becomes the memo's) — unless
        // it was never formed (an eval-only iteration that succeeded): lm_device.hpp, lm_iteration
        if (memo) {
          if (S.acc_at_x) { park = 1; S.memo_valid = 1; a.midx[p] = a.cur[p]; a.memo_cost[p] = a.lin_cost[p]; a.memo_ninl[p] = a.lin_ninl[p]; }
          else S.memo_valid = 0;
        }
        action = 1;
        S.acc_at_x = 0;
        S.has_last_dx = 1;
        S.last_was_success = 1;
        if (opt.check_final_cost && S.iter + 1 == S.max_iters) eval_only = true;
      } else {
        if (S.has_last_dx) { action = 2; S.has_last_dx = 0; S.acc_at_x = 0; check = (memo && S.memo_valid) ? 1 : 0; }
        else if (status & 2) { action = 1; S.has_last_dx = 1; S.acc_at_x = 0; }
        eval_only = (S.last_was_success == 0);
        S.last_was_success = 0;
      }
      if (opt.solver_type == 0) S.rebuild = eval_only ? 0 : 1;
      S.num_iters = S.num_iters + 1;
      S.iter = S.iter + 1;
      cont = (S.stop == TOA_STOP_NONE && S.iter < S.max_iters) ? 1 : 0;
    }
    sh_action = action;
    sh_cont = cont;
    sh_retry = rc < 0 ? 1 : 0;
    sh_park = park;
    sh_check = check;
  }
  __syncthreads();
  const int action = sh_action;
  if (sh_park) for (int i = tid; i < n; i += 256) { a.g_m[p * n + i] = g[i]; a.hdu_m[p * n + i] = a.hdu[p * n + i]; a.xs_m[p * n + i] = x[i]; }   // (x BEFORE the step)
  if (action == 1) for (int i = tid; i < n; i += 256) { const T d = dx[i]; x[i] += d; ldx[i] = d; }  // traits.h:184-190
  if (action == 2) for (int i = tid; i < n; i += 256) x[i] -= ldx[i];
  if (sh_check) {
    // (x + dx) - dx is x again only when both roundings cancel: the BIT PATTERNS are compared with the parked point's, and only a
    // match in every component lets the next Build read the memo back — never an approximation
    __syncthreads();
    double diff = 0;
    for (int i = tid; i < n; i += 256) diff += bits_equal(x[i], a.xs_m[p * n + i]) ? 0.0 : 1.0;
    diff = block_sum<T>(diff, red);
    if (tid == 0) S.memo_hit = diff == 0 ? 1 : 0;
  }
  const bool retry = sh_retry != 0;
  if (sh_cont) {
    if (tid == 0) {
      // what the NEXT pass does with this problem (the rows / Gram kernels read skip [p], the stage / pre kernels cur [p])
      const bool acc_next = opt.solver_type != 0 || S.rebuild;
      int skip_next = 0;
      if (memo) {
        if (acc_next) skip_next = S.acc_at_x ? 1 : (S.memo_hit ? 2 : 0);
        a.skip[p] = skip_next;
        if (skip_next == 2) a.cur[p] = a.midx[p];
        else if (acc_next && !skip_next && S.memo_valid && a.cur[p] == a.midx[p]) a.cur[p] ^= 1;   // never accumulate over the parked slot
      }
      atomicAdd(&summary[0], 1);
      if (acc_next && !skip_next) atomicAdd(&summary[1], 1);
      if (a.stepped) {   // stepping form: the iteration is over unless the solve is being retried
        if (retry) atomicAdd(&summary[2], 1);
        else {
          a.stepped[p] = 1;   // (every thread read on(p) before the barrier above)
          res.stop_reason[p] = TOA_STOP_NONE;   // running: the results so far (toa_lm_step's contract)
          res.num_iters[p] = S.num_iters;
          res.final_cost[p] = S.final_cost;
          if (res.num_failures) res.num_failures[p] = int(S.num_failures);
          if (res.num_consec_failures) res.num_consec_failures[p] = int(S.num_consec);
          if (a.active_out) atomicAdd(a.active_out, 1);
        }
      }
    }
    return;
  }
  large_finish_problem<T>(a, p);
}


// toa_lm_stop for this pipeline: end the running problems p with stop_request [p] != 0 with that StopReason
// (optimizer.h:302-305,529-534); their rows are finalised exactly as for a problem that stops by itself.
template <typename T>
__global__ void __launch_bounds__(256) large_stop_kernel(const LargeArgs<T> a) {
  const long long p = blockIdx.x;
  if (!a.active[p]) return;
  const int req = a.stop_request[p];
  if (req == TOA_STOP_NONE) return;
  if (threadIdx.x == 0) a.st[p].stop = req;
  __syncthreads();
  large_finish_problem<T>(a, p);
}

// What the stepping form keeps between calls (toa_lm_state_bytes at n >= 64): the scalar state, the running flags, and the
// linearisation an eval-only iteration keeps solving with (g, diag H, H — lm.h:96-117) plus the last steps (roll-back).
template <typename T>
struct LargeStateLayout {
  size_t st, active, built, g, hd, dx, ldx, H, total;
  __host__ __device__ LargeStateLayout(int n, long long P) {
    auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
    const size_t b_vec = al(size_t(P) * n * sizeof(T));
    size_t o = 0;
    st = o; o += al(size_t(P) * sizeof(LmState<T>));
    active = o; o += al(size_t(P) * sizeof(int));
    built = o; o += al(size_t(P) * sizeof(int));
    g = o; o += b_vec; hd = o; o += b_vec; dx = o; o += b_vec; ldx = o; o += b_vec;
    H = o; o += al(size_t(P) * n * n * sizeof(T));
    total = o;
  }
};

// toa_lm_step_info for that state block: cost, |dx|^2, |g|^2 (the arithmetic of large_post_kernel) and the two vectors
template <typename T>
__global__ void __launch_bounds__(256) large_step_info_kernel(const char* __restrict__ state, const long long P, const int n,
                                                              double* __restrict__ err, double* __restrict__ dx2,
                                                              double* __restrict__ g2, T* __restrict__ dx_out, T* __restrict__ g_out) {
  __shared__ double red[256];
  const long long p = blockIdx.x;
  const LargeStateLayout<T> lay(n, P);
  const LmState<T>& S = reinterpret_cast<const LmState<T>*>(state + lay.st)[p];
  const T* dx = reinterpret_cast<const T*>(state + lay.dx) + p * n;
  const T* g = reinterpret_cast<const T*>(state + lay.g) + p * n;
  double sd = 0, sg = 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const T d = dx[i], gi = g[i];
    sd += double(d * d);
    sg += double(gi * gi);
    if (dx_out) dx_out[p * n + i] = d;
    if (g_out) g_out[p * n + i] = gi;
  }
  sd = double(T(block_sum<T>(sd, red)));
  sg = double(T(block_sum<T>(sg, red)));
  if (threadIdx.x == 0) {
    if (err) err[p] = S.cost_val;
    if (dx2) dx2[p] = sd;
    if (g2) g2[p] = sg;
  }
}

template <typename T>
__global__ void __launch_bounds__(256) large_step_log_kernel(const char* __restrict__ state, const long long P, const int n, double* __restrict__ lambda,
                                                             int* __restrict__ nres, int* __restrict__ ninl) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const LargeStateLayout<T> lay(n, P);
  const LmState<T>& S = reinterpret_cast<const LmState<T>*>(state + lay.st)[p];
  if (lambda) lambda[p] = double(S.lambda);
  if (nres) nres[p] = S.cost_nres;
  if (ninl) ninl[p] = S.cost_ninl;
}

// mode 0: the whole solve.  1 / 2 / 3: toa_lm_begin / toa_lm_step / toa_lm_stop on `state` (LargeStateLayout).
template <typename T>
int large_lm_run_t(toa_handle h, int n, int m, int64_t P, const T* data, T* x, const toa_options& opt,
                   const toa_results& res, uint64_t* counters, int mode = 0, void* state = nullptr,
                   int32_t* active_dev = nullptr, const int32_t* stop_request = nullptr, const LargeSeamOut<T>* seam_out = nullptr) {
  const size_t nn = size_t(n) * n;
  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  const unsigned max_tries = opt.max_consec_failures > 0 ? (opt.max_consec_failures > 1 ? opt.max_consec_failures : 1) : 255;
  const long long max_passes = (long long)(opt.max_iters + 2 + (opt.check_final_cost ? 1 : 0)) * (long long)(max_tries + 1);
  const size_t b_st = al(size_t(P) * sizeof(LmState<T>)), b_i = al(size_t(P) * sizeof(int));
  const size_t b_vec = al(size_t(P) * n * sizeof(T)), b_mat = al(size_t(P) * nn * sizeof(T));
  const size_t b_J = al(size_t(P) * m * n * sizeof(T)), b_r = al(size_t(P) * m * sizeof(T));
  const bool seam = mode == 4;              // toa_accumulate: one data pass, no Build (seam_out says where the results go)
  const bool stepping = mode != 0 && !seam;
  const LargeStateLayout<T> lay(n, P);
  const long long step_passes = (long long)max_tries + 2;   // one iteration: a pass + its retries (optimizer.h:370-390)
  const size_t b_sum = stepping ? al(size_t(4) * sizeof(int) * size_t(step_passes + 1)) : al(size_t(2) * sizeof(int) * size_t(max_passes + 1) * toa_context::kLanes);
  // vectorised rows kernel (see large_rows_vec_kernel): geometry and the per-wave J^T r partials
  constexpr int VEC = 16 / int(sizeof(T));
  // (n > 1024, round 5: the general rows kernel + the library's GEMM / GEMV / factorisation — the vectorised rows kernel keeps at most
  //  eight 16-byte vectors of a row per lane, the hand-written Gram's LDS stage ends at 1024 columns)
  const bool vec_shape = n <= 1024 && n % VEC == 0 && (size_t(m) * (n + 1)) % VEC == 0 && reinterpret_cast<uintptr_t>(data) % 16 == 0;
  const int nv = n / VEC;
  int LPR = 1;
  while (LPR * 2 <= std::min(64, nv)) LPR *= 2;
  // (large_rows_vec_kernel works R trips at a time and hands trip u's scalar work to lane u of the row group: it needs R <= LPR
  //  lanes per row.  Narrow rows — n = 4 .. 16 in fp32, 2 .. 8 in fp64 — have fewer and take the general rows kernel)
  const int KV0 = vec_shape ? (nv + LPR - 1) / LPR : 0;
  const int Rtrips = KV0 >= 5 ? 1 : (KV0 >= 3 ? 2 : (KV0 == 2 ? 4 : 8));
  const bool vec_ok = vec_shape && LPR >= Rtrips;
  const int KV = vec_ok ? KV0 : 0;
  const int RPW = 64 / LPR;
  // (capped at one workgroup per compute unit and problem: every workgroup leaves four partial J^T r vectors that
  //  large_pre_kernel sums one after the other — 16 000 of them for ONE problem of 65 536 rows took longer than the data pass)
  const long long wg_target = std::min<long long>((long long)h->num_cus * 16 / std::max<long long>(1, P) + 1, h->num_cus);
  const unsigned row_blocks = unsigned(std::max<long long>(1, std::min<long long>((m + 4 * (vec_ok ? RPW : 4) - 1) / (4 * (vec_ok ? RPW : 4)), wg_target)));
  const int gslots = vec_ok ? int(row_blocks) * 4 : 0;
  const size_t b_gpart = al(size_t(P) * size_t(std::max(gslots, 1)) * n * sizeof(T));
  const size_t b_ptr = al(size_t(P) * sizeof(void*));
  // H = J^T J: the hand-written LDS-staged Gram (large_gram_kernel) for fp32 rows that are 16-byte aligned; the library GEMM
  // otherwise (fp64, odd shapes) or with toa_tuning::large_library_gram.  Same box, 128 problems x n = 256 x m = 8192 fp32
  // (profiles/r03_ab_log.md): rocblas_sgemm_batched 1.22 ms per pass with every problem rebuilding (the full square,
  // 113 TFLOP/s) plus the J = diag(s) A round trip through HBM in the rows kernel (0.43 -> 0.23 ms without it); this kernel
  // 0.74 ms (the lower tile triangle only: 56 % of the flops, at 104 TFLOP/s issued).
  const bool force_gemm = h->tune.large_library_gram != 0;
  const bool own_gram = sizeof(T) == 4 && vec_ok && !force_gemm && n % 4 == 0;
  int gram_R = 1, gram_rows = (m + 3) & ~3;
  const SyrkGeom geo = syrk_geom(n);
  // 224 < n <= 256: the operand-sharing deal of the tiles (syrk_stage_tri; toa_tuning::large_gram_plain_deal: the round-robin deal, same bits)
  const bool gram_tri = !h->tune.large_gram_plain_deal && geo.nt == 8 && geo.groups == 1 && geo.waves == 12 && geo.tpw == 3;
  if (own_gram) {   // about four workgroups per CU when every problem rebuilds (a late pass with few of them left is bound by ONE
                    // workgroup's run time, which shrinks with the chunk); the rows of a chunk a multiple of the LDS stage
    const long long want_wgs = (long long)h->num_cus * 4, per_chunk = P * geo.groups;
    gram_R = int(std::max<long long>(1, std::min<long long>(256, (want_wgs + per_chunk - 1) / per_chunk)));   // (a FEW huge problems: up to 256 row chunks each)
    gram_rows = (((m + gram_R - 1) / gram_R) + geo.K - 1) / geo.K * geo.K;
    gram_R = (m + gram_rows - 1) / gram_rows;
  }
  const size_t b_sc = own_gram ? al(size_t(P) * m * sizeof(T)) : 0;
  const size_t b_gp = own_gram ? al(size_t(P) * gram_R * geo.T * 1024 * sizeof(T)) : 0;
  const size_t b_Juse = own_gram ? 0 : b_J;   // J = diag(s) A is only materialised for the library GEMM
  // use_ldlt = false (gn.h:157-162): dx = -H.inverse() * g "without any checks on invertibility" (options.h:59) — a general LU
  // with partial pivoting from the library (what Eigen's inverse() is), its verdict ignored; the step is still refused if it is
  // not finite, as everywhere else in this path
  const bool lu = !opt.use_ldlt;
  const size_t b_piv = lu ? al(size_t(P) * n * sizeof(int)) : 0;
  // what outlives a pass (LargeStateLayout) sits in the caller's state block in the stepping form, in the scratch otherwise
  // memo of linearisations (LargeArgs::skip ...): the whole-solve form only — the stepping form's state block has one H slot
  const bool memo_on = !stepping && !seam && !h->tune.memo_off;
  const size_t b_dbl = al(size_t(P) * sizeof(double));
  const size_t b_memo = memo_on ? 5 * b_i + 4 * b_vec + 2 * b_dbl + 2 * b_mat : 0;
  const bool robust = h->loss != TOA_LOSS_L2;
  const size_t b_loss = robust ? b_r + b_i : 0;
  const size_t need = (stepping ? 0 : lay.total) + 2 * b_i + 2 * b_vec + 2 * b_mat + b_Juse + b_r + b_sum + b_gpart + 2 * b_ptr + b_sc + b_gp + b_piv + b_memo + b_loss;
  if (int rc = ensure_scratch(h, need, "large-n LM (the J scratch is P*m*n)")) return rc;
  char* q = static_cast<char*>(h->scratch);
  auto take = [&](size_t b) { char* r = q; q += b; return r; };
  char* keep = stepping ? static_cast<char*>(state) : take(lay.total);
  LargeArgs<T> a;
  a.data = data; a.x = x; a.n = n; a.m = m; a.P = P; a.opt = opt; a.res = res;
  a.counters = reinterpret_cast<unsigned long long*>(counters);
  a.st = reinterpret_cast<LmState<T>*>(keep + lay.st);
  a.active = reinterpret_cast<int*>(keep + lay.active);
  a.built = reinterpret_cast<int*>(keep + lay.built);
  a.g = reinterpret_cast<T*>(keep + lay.g); a.hd = reinterpret_cast<T*>(keep + lay.hd); a.dx = reinterpret_cast<T*>(keep + lay.dx);
  a.ldx = reinterpret_cast<T*>(keep + lay.ldx); a.H = reinterpret_cast<T*>(keep + lay.H);
  a.info = reinterpret_cast<int*>(take(b_i));
  a.stepped = stepping ? reinterpret_cast<int*>(take(b_i)) : (take(b_i), nullptr);
  a.active_out = active_dev; a.stop_request = stop_request;
  a.gnew = reinterpret_cast<T*>(take(b_vec)); a.rhs = reinterpret_cast<T*>(take(b_vec));
  a.Hnew = reinterpret_cast<T*>(take(b_mat)); a.work = reinterpret_cast<T*>(take(b_mat));
  a.J = reinterpret_cast<T*>(take(b_Juse)); a.r = reinterpret_cast<T*>(take(b_r));
  a.sc = reinterpret_cast<T*>(take(b_sc)); a.gram_part = reinterpret_cast<T*>(take(b_gp));
  a.own_gram = own_gram ? 1 : 0; a.gram_R = gram_R; a.gram_rows = gram_rows;
  a.summary = reinterpret_cast<int*>(take(b_sum));
  a.gpart = reinterpret_cast<T*>(take(b_gpart));
  a.gslots = gslots;
  a.jptr = reinterpret_cast<const T**>(take(b_ptr));
  a.hptr = reinterpret_cast<T**>(take(b_ptr));
  int* ipiv = reinterpret_cast<int*>(take(b_piv));
  a.skip = a.cur = a.midx = nullptr;
  a.hslots = 1;
  a.hdu = a.g_m = a.hdu_m = a.xs_m = nullptr;
  a.lin_cost = a.memo_cost = nullptr;
  a.lin_ninl = a.memo_ninl = nullptr;
  a.loss = h->loss; a.th2 = T(h->loss_th2); a.lossv = nullptr; a.ninl = nullptr;
  if (robust) { a.lossv = reinterpret_cast<T*>(take(b_r)); a.ninl = reinterpret_cast<int*>(take(b_i)); }
  if (memo_on) {
    a.lin_ninl = reinterpret_cast<int*>(take(b_i)); a.memo_ninl = reinterpret_cast<int*>(take(b_i));
    a.skip = reinterpret_cast<int*>(take(b_i)); a.cur = reinterpret_cast<int*>(take(b_i)); a.midx = reinterpret_cast<int*>(take(b_i));
    a.hdu = reinterpret_cast<T*>(take(b_vec)); a.g_m = reinterpret_cast<T*>(take(b_vec));
    a.hdu_m = reinterpret_cast<T*>(take(b_vec)); a.xs_m = reinterpret_cast<T*>(take(b_vec));
    a.lin_cost = reinterpret_cast<double*>(take(b_dbl)); a.memo_cost = reinterpret_cast<double*>(take(b_dbl));
    a.H = reinterpret_cast<T*>(take(2 * b_mat));   // two slots per problem (the state block's single one stays unused)
    a.hslots = 2;
  }
  hipStream_t st = h->stream;
  // 64 <= n <= 128: the workgroup LDL^T above; beyond (or with toa_tuning::large_library_solver) rocSOLVER
  const bool force_lib = h->tune.large_library_solver != 0;
  const size_t chol_lds = ldlt_image_bytes<T>(n);
  const bool own_chol = !lu && !force_lib && n <= 128 && chol_lds + 4096 <= size_t(h->max_lds);  // measured crossover (tools/k3_crossover.py)
  if (own_chol) {
    if (int rc = ensure_lds_attr(h, ldlt_solve_fn<T>(n), chol_lds)) return rc;
  }
  // beyond 128 unknowns: the one-workgroup blocked Cholesky + substitutions (large_chol_solve_kernel) while its panel fits the
  // LDS (fp32: n <= 1024, fp64: n <= 512); the library beyond, for use_ldlt = false, and with toa_tuning::large_library_solver
  const size_t chol2_lds = chol_solve_lds_bytes<T>(n);
  const bool own_chol2 = !own_chol && !lu && !force_lib && n > 128 && chol2_lds + 2048 <= size_t(h->max_lds);
  if (own_chol2) {
    if (int rc = ensure_lds_attr(h, (const void*)large_chol_solve_kernel<T>, chol2_lds)) return rc;
    if (int rc = ensure_lds_attr(h, (const void*)large_chol_solve_kernel<T, false>, chol2_lds)) return rc;
  }
  // rocBLAS / rocSOLVER are opened (dlopen, a handle: a cold load of their code objects, minutes in a bare process) only
  // when a stage of THIS solve is theirs — fp32 with aligned rows up to n = 1024 never touches them
  const bool needs_lib = seam ? !own_gram : !(own_gram && (own_chol || own_chol2));
  static RocApi no_api;
  RocApi& api = needs_lib ? roc_api() : no_api;
  if (needs_lib) {
    if (!api.ok) return toa_fail(TOA_E_UNSUPPORTED, "large-n LM needs rocBLAS + rocSOLVER for this shape: " + api.err);
    if (int rc = ensure_blas(h, api)) return rc;
  }
  if (mode == 3) {   // toa_lm_stop
    hipLaunchKernelGGL(large_stop_kernel<T>, dim3(unsigned(P)), dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    return TOA_OK;
  }
  if (mode <= 1 || seam) {   // the whole solve, or toa_lm_begin: construct the state (no data pass)
    HIP_TRY(hipMemsetAsync(a.ldx, 0, b_vec, st));
    HIP_TRY(hipMemsetAsync(a.dx, 0, b_vec, st));
    HIP_TRY(hipMemsetAsync(a.g, 0, b_vec, st));
    // counters ACCUMULATE on every path (fused, row-split, stepping, here): the caller zeroes them (include/tinyopt_amd.h)
    hipLaunchKernelGGL(large_init_kernel<T>, dim3(unsigned((P + 255) / 256)), dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    if (mode == 1) return TOA_OK;
    if (seam) hipLaunchKernelGGL(large_seam_prep_kernel<T>, dim3(unsigned((P + 255) / 256)), dim3(256), 0, st, a, seam_out->want_grad);
  }
  HIP_TRY(hipMemsetAsync(a.summary, 0, b_sum, st));
  if (stepping) HIP_TRY(hipMemsetAsync(a.stepped, 0, b_i, st));
  if (robust) HIP_TRY(hipMemsetAsync(a.ninl, 0, b_i, st));   // (the scratch block is shared between calls: the stepping form starts every step from zero)
  const T one = 1, zero = 0;
  // row slices of the n x n staging copies: ~8 workgroups per CU over the batch, at least 8 rows each
  const int stage_rows = int(std::max<long long>(8, (long long)n * P / std::max<long long>(1, (long long)h->num_cus * 8)));
  // Round 4: where every stage of a pass is a kernel of ours (hand-written Gram + own Cholesky: fp32, aligned rows, n <= 1024)
  // nothing in a pass needs a host-side count, every kernel skips the problems that have finished, and each pass leaves its
  // (active, want-Jacobian) pair in its own slot of `summary` — so the host enqueues kAhead passes ahead and looks at pass
  // k's pair (copied to pinned memory behind the pass) only before it enqueues pass k + kAhead: no hipStreamSynchronize in the
  // loop, the GPU always has the next pass queued (round 3: a blocking 8-byte read-back per pass, ~35 us of idle GPU each).
  // The at most kAhead surplus passes after the last problem has finished are launches of kernels that return at once.
  // With a LIBRARY stage in the pass (rocBLAS GEMM sized by want_j, rocSOLVER over all P matrices) the pass-by-pass
  // hand-shake stays: a surplus library pass would cost more than the read-back.
  constexpr int kAhead = 2, kRing = toa_context::kPassRing;
  const bool ahead = !stepping && own_gram && (own_chol || own_chol2);
  const int sum_stride = stepping ? 4 : 2;   // the stepping form also counts the retries still owed (slot 2)
  const long long pass_limit = stepping ? step_passes : max_passes;

  // One pass of the loop body for the problems of `la` (the whole batch, or one lane's share of it) on stream ls
  auto enqueue_pass = [&](const LargeArgs<T>& la, hipStream_t ls, const long long Pl, int* sum_dev, int& want_j,
                          hipEvent_t wait_first = nullptr, hipEvent_t record_after_gram = nullptr) -> int {
    if (wait_first) HIP_TRY(hipStreamWaitEvent(ls, wait_first, 0));
    const dim3 rgrid(row_blocks, unsigned(Pl));
    const size_t xs_bytes = size_t(n) * sizeof(T);
    if (!vec_ok) {
      hipLaunchKernelGGL(large_rows_kernel<T>, rgrid, dim3(256), xs_bytes, ls, la);
    } else {
      switch (KV) {
        case 1: hipLaunchKernelGGL((large_rows_vec_kernel<T, 1>), rgrid, dim3(256), xs_bytes, ls, la, LPR); break;
        case 2: hipLaunchKernelGGL((large_rows_vec_kernel<T, 2>), rgrid, dim3(256), xs_bytes, ls, la, LPR); break;
        case 3: hipLaunchKernelGGL((large_rows_vec_kernel<T, 3>), rgrid, dim3(256), xs_bytes, ls, la, LPR); break;
        case 4: hipLaunchKernelGGL((large_rows_vec_kernel<T, 4>), rgrid, dim3(256), xs_bytes, ls, la, LPR); break;
        case 5: hipLaunchKernelGGL((large_rows_vec_kernel<T, 5>), rgrid, dim3(256), xs_bytes, ls, la, LPR); break;
        case 6: hipLaunchKernelGGL((large_rows_vec_kernel<T, 6>), rgrid, dim3(256), xs_bytes, ls, la, LPR); break;
        case 7: hipLaunchKernelGGL((large_rows_vec_kernel<T, 7>), rgrid, dim3(256), xs_bytes, ls, la, LPR); break;
        default: hipLaunchKernelGGL((large_rows_vec_kernel<T, 8>), rgrid, dim3(256), xs_bytes, ls, la, LPR); break;
      }
    }
    if (own_gram) {
      if constexpr (sizeof(T) == 4) {
        const dim3 ggrid(unsigned(geo.groups * gram_R), unsigned(Pl)), gblock(unsigned(geo.waves * 64));
        const size_t glds = size_t(2) * geo.K * geo.ns * sizeof(T) + 256;   // (+ the one-step over-read of the operand prefetch)
        switch (geo.tpw) {
          case 1: hipLaunchKernelGGL((large_gram_kernel<T, 1>), ggrid, gblock, glds, ls, la, geo); break;
          case 2: hipLaunchKernelGGL((large_gram_kernel<T, 2>), ggrid, gblock, glds, ls, la, geo); break;
          case 3:
            if (gram_tri) hipLaunchKernelGGL((large_gram_kernel<T, 3, true>), ggrid, gblock, glds, ls, la, geo);
            else hipLaunchKernelGGL((large_gram_kernel<T, 3>), ggrid, gblock, glds, ls, la, geo);
            break;
          default: hipLaunchKernelGGL((large_gram_kernel<T, 4>), ggrid, gblock, glds, ls, la, geo); break;
        }
        hipLaunchKernelGGL(large_gram_reduce_kernel<T>, dim3(unsigned(geo.T), unsigned(Pl)), dim3(256), 0, ls, la, geo);
        if (record_after_gram) HIP_TRY(hipEventRecord(record_after_gram, ls));
      }
    } else if (want_j > 0) {
      if (stepping) {   // the host has no count from a previous pass here: the list's length comes back with it
        int* cnt_dev = sum_dev + 3;
        hipLaunchKernelGGL(large_compact_kernel<T>, dim3(1), dim3(256), 0, ls, la, cnt_dev);
        HIP_TRY(hipMemcpyAsync(&want_j, cnt_dev, sizeof(int), hipMemcpyDeviceToHost, ls));
        HIP_TRY(hipStreamSynchronize(ls));
      } else {
        hipLaunchKernelGGL(large_compact_kernel<T>, dim3(1), dim3(256), 0, ls, la);
      }
    }
    if (!own_gram && want_j > 0) {
      int rc;  // J row-major [m][n] == column-major n x m (ld n):  H = Jc Jc^T,  g = Jc r
      if constexpr (sizeof(T) == 4) {
        rc = api.sgemm_b(h->blas, kOpN, kOpT, n, n, m, &one, la.jptr, n, la.jptr, n, &zero, la.hptr, n, want_j);
        if (rc == 0 && !vec_ok) rc = api.sgemv(h->blas, kOpN, n, m, &one, la.J, n, int64_t(m) * n, la.r, 1, m, &zero, la.gnew, 1, n, int(Pl));
      } else {
        rc = api.dgemm_b(h->blas, kOpN, kOpT, n, n, m, &one, la.jptr, n, la.jptr, n, &zero, la.hptr, n, want_j);
        if (rc == 0 && !vec_ok) rc = api.dgemv(h->blas, kOpN, n, m, &one, la.J, n, int64_t(m) * n, la.r, 1, m, &zero, la.gnew, 1, n, int(Pl));
      }
      if (rc != 0) return toa_fail(TOA_E_HIP, "rocBLAS gemm/gemv returned status " + std::to_string(rc));
    }
    if (seam) {
      hipLaunchKernelGGL(large_seam_out_kernel<T>, dim3(unsigned(Pl)), dim3(256), 0, ls, la, *seam_out);
      HIP_TRY(hipGetLastError());
      return TOA_OK;
    }
    hipLaunchKernelGGL(large_pre_kernel<T>, dim3(unsigned(Pl)), dim3(256), 0, ls, la);
    hipLaunchKernelGGL(large_stage_kernel<T>, dim3(unsigned((n + stage_rows - 1) / stage_rows), unsigned(Pl)), dim3(256), 0, ls, la, stage_rows);
    int rc = 0;
    if (own_chol) {
      launch_ldlt_solve<T>(n, unsigned(Pl), chol_lds, ls, la);
    } else if (own_chol2) {
      if (h->tune.large_chol_no_lookahead) hipLaunchKernelGGL((large_chol_solve_kernel<T, false>), dim3(unsigned(Pl)), dim3(kCholThreads), chol2_lds, ls, la);
      else hipLaunchKernelGGL(large_chol_solve_kernel<T>, dim3(unsigned(Pl)), dim3(kCholThreads), chol2_lds, ls, la);
    } else if (lu) {
      if constexpr (sizeof(T) == 4) {
        rc = api.sgetrf(h->blas, n, n, la.work, n, int64_t(nn), ipiv, int64_t(n), la.info, int(Pl));
        if (rc == 0) rc = api.sgetrs(h->blas, kOpN, n, 1, la.work, n, int64_t(nn), ipiv, int64_t(n), la.rhs, n, int64_t(n), int(Pl));
      } else {
        rc = api.dgetrf(h->blas, n, n, la.work, n, int64_t(nn), ipiv, int64_t(n), la.info, int(Pl));
        if (rc == 0) rc = api.dgetrs(h->blas, kOpN, n, 1, la.work, n, int64_t(nn), ipiv, int64_t(n), la.rhs, n, int64_t(n), int(Pl));
      }
      HIP_TRY(hipMemsetAsync(la.info, 0, size_t(Pl) * sizeof(int), ls));   // unchecked: a singular pivot is not a failure by itself
    } else if constexpr (sizeof(T) == 4) {
      rc = api.spotrf(h->blas, kFillUpper, n, la.work, n, int64_t(nn), la.info, int(Pl));
      if (rc == 0) rc = api.spotrs(h->blas, kFillUpper, n, 1, la.work, n, int64_t(nn), la.rhs, n, int64_t(n), int(Pl));
    } else {
      rc = api.dpotrf(h->blas, kFillUpper, n, la.work, n, int64_t(nn), la.info, int(Pl));
      if (rc == 0) rc = api.dpotrs(h->blas, kFillUpper, n, 1, la.work, n, int64_t(nn), la.rhs, n, int64_t(n), int(Pl));
    }
    if (rc != 0) return toa_fail(TOA_E_HIP, "rocSOLVER potrf/potrs returned status " + std::to_string(rc));
    hipLaunchKernelGGL(large_post_kernel<T>, dim3(unsigned(Pl)), dim3(256), 0, ls, la, sum_dev);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipGetLastError());
    return TOA_OK;
  };

  if (seam) {
    int want_j = seam_out->want_grad ? int(P) : 0;
    return enqueue_pass(a, st, P, a.summary, want_j);
  }
  // Under stream capture (hipGraph) the host can look at nothing: where every stage is ours the whole pass budget is recorded —
  // max_passes passes, each kernel of which returns at once for the problems that have finished — and the graph replays the
  // solve with no host in the loop (one lane: a captured fork / join would only add edges).  A typical solve needs a fifth of
  // the budget; the rest costs ~40 us of empty launches per pass.  With a library stage in the pass there is no such form.
  {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
      if (!ahead)
        return toa_fail(TOA_E_UNSUPPORTED, "large-n LM under stream capture: only the solves whose every stage is a kernel of this library "
                                           "(fp32, 16-byte aligned rows, use_ldlt, n <= 1024, not the stepping form) can be captured");
      // The budget is an internal bound, not something the caller chose: with max_consec_failures == 0 a Build may be retried
      // 255 times per iteration, i.e. tens of thousands of passes of ~7 kernels each — a graph of 10^5 nodes whose every replay
      // pays ~40 us per empty pass.  Such a budget is refused here with the numbers; the eager call has no such limit.
      constexpr long long kMaxCapturedPasses = 1024;
      if (max_passes > kMaxCapturedPasses)
        return toa_fail(TOA_E_UNSUPPORTED, "large-n LM under stream capture: the pass budget of these options is " + std::to_string(max_passes) +
                                           " passes (~" + std::to_string(max_passes * 7) + " graph nodes, ~" + std::to_string(max_passes * 40 / 1000) +
                                           " ms of empty launches per replay); at most " + std::to_string(kMaxCapturedPasses) +
                                           " are recorded — set max_consec_failures > 0 (the retry bound per iteration) or lower max_iters");
      h->shadow_retired = true;   // (a graph of this handle now exists: its workspaces are never freed under it, toa_release_workspace)
      for (long long pass = 0; pass < max_passes; ++pass) {
        int want_all = int(P);
        if (int rc = enqueue_pass(a, st, P, a.summary + 2 * pass, want_all)) return rc;
      }
      return TOA_OK;
    }
  }
  if (!ahead) {   // a library stage in the pass, or the stepping form: pass by pass, the counts read back behind each
    int active = int(P), want_j = int(P);
    for (long long pass = 0; pass < pass_limit && active > 0; ++pass) {
      int* sum_dev = a.summary + sum_stride * pass;
      if (int rc = enqueue_pass(a, st, P, sum_dev, want_j)) return rc;
      int sum_host[3] = {0, 0, 0};
      HIP_TRY(hipMemcpyAsync(sum_host, sum_dev, size_t(stepping ? 3 : 2) * sizeof(int), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      active = sum_host[0];
      want_j = sum_host[1];
      if (stepping) {   // another pass only for the problems whose solve is being retried
        if (sum_host[2] == 0) return TOA_OK;
        active = sum_host[2];
        want_j = int(P);
      }
    }
    if (stepping) return TOA_OK;   // (retries are bounded by max_tries inside the state machine)
    if (active > 0) return toa_fail(TOA_E_HIP, "large-n LM: pass budget exhausted with active problems (internal error)");
    return TOA_OK;
  }

  // ---- own kernels only: enqueue-ahead, and the batch in LANES on separate streams.  The one-workgroup Cholesky is a latency
  // chain (0.33 ms per n = 256 matrix whatever the batch, on a fraction of the compute units) that the Gram of the SAME
  // problems has to wait for; with the batch split, one lane's factorisations run beside another lane's rows + Gram kernels
  // (lane l's pass starts behind lane l - 1's Gram: the stagger).  The geometry (row chunks per problem, row blocks, stage
  // slices) stays that of the whole batch, so every problem sees the arithmetic of the one-lane run: the bits do not depend
  // on the split (toa_tuning::large_one_lane: 1 = one lane, k > 1 = k lanes, for the A/B).  What it buys is modest — +6 % at
  // 128 x n = 256 — because a Gram workgroup (12 waves x 114 VGPRs at n = 256) leaves 152 registers per SIMD and the 8-wave
  // factorisation needs 224: the factorisations run on compute units the Gram does not have, the overlap is a partition of the chip and the chip-time of
  // a pass is conserved (profiles/r04_ab_log.md section 3).
  if (int rc = ensure_pass_ring(h)) return rc;
  int nlanes = h->tune.large_one_lane ? 1 : (P >= 16 ? 2 : 1);   // (measured, 128 x n = 256: 1 lane 80.5 k, 2: 85.5 k, 3: 85 k, 4: 85 k it/s)
  // ... with the operand-sharing Gram (one lane's Gram 451 -> 420 us) ONE lane is the faster schedule: 9.59 ms per 128-problem solve
  // against 9.93 with two lanes (plain deal, same call: 9.92 / 9.79) — profiles/r04_ab_log.md section 8d
  if (gram_tri && h->tune.large_one_lane == 0) nlanes = 1;
  if (h->tune.large_one_lane > 1) nlanes = std::min<int>(h->tune.large_one_lane, toa_context::kLanes);   // (A/B: an explicit lane count)
  nlanes = int(std::min<long long>(nlanes, std::max<long long>(1, P)));
  struct Lane { LargeArgs<T> a; hipStream_t st; long long P, pass; int active; bool done; int* host; hipEvent_t* ev; int* sums; };
  Lane lane[toa_context::kLanes];
  for (int l = 0; l < nlanes; ++l) {
    const long long p0 = P * l / nlanes, p1 = P * (l + 1) / nlanes;
    Lane& L = lane[l];
    L.a = a;
    L.P = p1 - p0; L.pass = 0; L.active = int(L.P); L.done = false;
    L.st = l == 0 ? st : h->lane_stream[l - 1];
    L.host = h->pass_flags + l * 2 * kRing;
    L.ev = h->pass_done + l * kRing;
    L.sums = a.summary + size_t(l) * 2 * size_t(max_passes + 1);
    LargeArgs<T>& b = L.a;
    b.P = L.P;
    b.data += size_t(p0) * m * (size_t(n) + 1); b.x += p0 * n;
    b.st += p0; b.active += p0; b.built += p0; b.info += p0;
    b.g += p0 * n; b.hd += p0 * n; b.dx += p0 * n; b.ldx += p0 * n; b.gnew += p0 * n; b.rhs += p0 * n;
    b.H += size_t(p0) * nn * size_t(a.hslots > 1 ? 2 : 1); b.Hnew += size_t(p0) * nn; b.work += size_t(p0) * nn;
    if (a.skip) {
      b.skip += p0; b.cur += p0; b.midx += p0; b.lin_cost += p0; b.memo_cost += p0; b.lin_ninl += p0; b.memo_ninl += p0;
      b.hdu += p0 * n; b.g_m += p0 * n; b.hdu_m += p0 * n; b.xs_m += p0 * n;
    }
    b.r += size_t(p0) * m; b.sc += size_t(p0) * m;
    if (b.lossv) { b.lossv += size_t(p0) * m; b.ninl += p0; }
    b.gram_part += size_t(p0) * gram_R * geo.T * 1024;
    b.gpart += size_t(p0) * size_t(std::max(gslots, 0)) * n;
    toa_results& r = b.res;
    auto adv = [&](auto*& ptr, size_t per) { if (ptr) ptr += size_t(p0) * per; };
    adv(r.stop_reason, 1); adv(r.num_iters, 1); adv(r.num_failures, 1); adv(r.num_consec_failures, 1);
    adv(r.final_cost, 1); adv(r.final_num_residuals, 1); adv(r.final_rerr_dec, 1); adv(r.final_hessian, nn);
    adv(r.errs, size_t(r.hist_stride)); adv(r.deltas2, size_t(r.hist_stride)); adv(r.successes, size_t(r.hist_stride));
    adv(r.final_inlier_ratio, 1);
  }
  struct Join {   // whatever happens below, nothing of ours is still running on the lane stream when the call returns
    toa_handle h; bool forked = false, joined = false;
    ~Join() { if (forked && !joined) for (hipStream_t s : h->lane_stream) if (s) (void)hipStreamSynchronize(s); }
  } join{h};
  if (nlanes > 1) {
    HIP_TRY(hipEventRecord(h->lane_fork, st));
    for (int l = 1; l < nlanes; ++l) HIP_TRY(hipStreamWaitEvent(lane[l].st, h->lane_fork, 0));
    join.forked = true;
  }
  bool all_done = false;
  while (!all_done) {
    all_done = true;
    for (int l = 0; l < nlanes; ++l) {
      Lane& L = lane[l];
      if (L.done) continue;
      if (L.pass >= max_passes) { L.done = true; continue; }
      if (L.pass >= kAhead) {
        const int slot = int((L.pass - kAhead) % kRing);
        HIP_TRY(hipEventSynchronize(L.ev[slot]));
        L.active = L.host[2 * slot];
        if (L.active == 0) { L.done = true; continue; }
      }
      int* sum_dev = L.sums + 2 * L.pass;
      int want_all = int(L.P);
      // the stagger: lane l's pass k starts when lane l - 1's Gram of pass k is done, so that from then on one lane's solve
      // runs beside another's data pass instead of all the Grams (and then all the solves) sharing the chip in lockstep
      hipEvent_t wait_ev = (l > 0 && !lane[l - 1].done) ? h->lane_gram[(l - 1) * kRing + L.pass % kRing] : nullptr;
      hipEvent_t rec_ev = (l + 1 < nlanes) ? h->lane_gram[l * kRing + L.pass % kRing] : nullptr;
      if (int rc = enqueue_pass(L.a, L.st, L.P, sum_dev, want_all, wait_ev, rec_ev)) return rc;
      const int slot = int(L.pass % kRing);
      HIP_TRY(hipMemcpyAsync(L.host + 2 * slot, sum_dev, 2 * sizeof(int), hipMemcpyDeviceToHost, L.st));
      HIP_TRY(hipEventRecord(L.ev[slot], L.st));
      ++L.pass;
      all_done = false;
    }
  }
  if (nlanes > 1) {
    for (int l = 1; l < nlanes; ++l) {
      HIP_TRY(hipEventRecord(h->lane_join[l - 1], lane[l].st));
      HIP_TRY(hipStreamWaitEvent(st, h->lane_join[l - 1], 0));
    }
    join.joined = true;
  }
  for (int l = 0; l < nlanes; ++l) {
    Lane& L = lane[l];
    if (L.active > 0 && L.pass > 0) {   // the pass budget ran out before a zero was seen: the LAST pass decides
      HIP_TRY(hipStreamSynchronize(L.st));
      L.active = L.host[2 * int((L.pass - 1) % kRing)];
    }
    if (L.active > 0) return toa_fail(TOA_E_HIP, "large-n LM: pass budget exhausted with active problems (internal error)");
  }
  return TOA_OK;
}


// The K3 seam (toa_solve_damped) for 64 <= n <= 128 on the workgroup LDL^T above instead of the library.
template <typename T>
__global__ void __launch_bounds__(256) large_fill_ones_kernel(int* __restrict__ v, const long long count,
                                                              const int32_t* __restrict__ mask, const long long mask_stride) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i < count) v[i] = mask ? (mask[i * mask_stride] != 0) : 1;
}

template <typename T>
int large_solve_own_t(toa_handle h, int n, int64_t P, const T* H, const T* g, double scale, T* dx, int32_t* ok) {
  const size_t nn = size_t(n) * n;
  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  const size_t b_work = al(size_t(P) * nn * sizeof(T)), b_rhs = al(size_t(P) * n * sizeof(T)), b_i = al(size_t(P) * sizeof(int));
  if (int rc = ensure_scratch(h, b_work + b_rhs + 2 * b_i, "large-n solve")) return rc;
  char* base = static_cast<char*>(h->scratch);
  LargeArgs<T> a;
  std::memset(&a, 0, sizeof(a));
  a.n = n;
  a.P = P;
  a.work = reinterpret_cast<T*>(base);
  a.rhs = reinterpret_cast<T*>(base + b_work);
  a.info = reinterpret_cast<int*>(base + b_work + b_rhs);
  a.active = reinterpret_cast<int*>(base + b_work + b_rhs + b_i);  // every matrix is solved, unless the caller masks some out
  a.built = a.active;
  const size_t chol_lds = n <= 128 ? ldlt_image_bytes<T>(n) : chol_solve_lds_bytes<T>(n);
  if (int rc = ensure_lds_attr(h, n <= 128 ? ldlt_solve_fn<T>(n) : (const void*)large_chol_solve_kernel<T>, chol_lds)) return rc;
  if (n > 128) { if (int rc = ensure_lds_attr(h, (const void*)large_chol_solve_kernel<T, false>, chol_lds)) return rc; }
  hipLaunchKernelGGL(large_fill_ones_kernel<T>, dim3(unsigned((P + 255) / 256)), dim3(256), 0, h->stream, a.active, (long long)P,
                     h->solve_mask, (long long)h->solve_mask_stride);
  const unsigned gx = unsigned(std::min<size_t>((nn + 255) / 256, 64));
  hipLaunchKernelGGL(large_damp_kernel<T>, dim3(gx, unsigned(P)), dim3(256), 0, h->stream, H, g, a.work, a.rhs, n, scale);
  if (n <= 128) launch_ldlt_solve<T>(n, unsigned(P), chol_lds, h->stream, a);
  else if (h->tune.large_chol_no_lookahead) hipLaunchKernelGGL((large_chol_solve_kernel<T, false>), dim3(unsigned(P)), dim3(kCholThreads), chol_lds, h->stream, a);
  else hipLaunchKernelGGL(large_chol_solve_kernel<T>, dim3(unsigned(P)), dim3(kCholThreads), chol_lds, h->stream, a);
  hipLaunchKernelGGL(large_finish_kernel<T>, dim3(unsigned(P)), dim3(256), 0, h->stream, a.rhs, a.info, dx, ok, n,
                     (const int*)a.active);
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}

}  // namespace
}  // namespace toa

int toa_large_solve_each(toa_handle h, int dtype, int n, int64_t P, const void* H, const void* g, double scale, void* dx, int32_t* ok);
// (H + 0) dx = -g for 128 < n with the matrices factorised IN PLACE (H is destroyed) and no launch but the factorisation's: for a
// caller that rebuilds H before every call anyway (bundle adjustment with lists: the reduced camera system of a pass).  Falls back to
// toa_large_solve_each (one library call per matrix: batch-independent bits) where the one-workgroup Cholesky does not apply.  h->solve_mask as there.
int toa_large_solve_inplace(toa_handle h, int dtype, int n, int64_t P, void* H, const void* g, void* dx, int32_t* ok) {
  const size_t lds = (size_t(32) * 36 + size_t(n) * 37 + 96) * (dtype == TOA_F32 ? 4 : 8) + 64;   // chol_solve_lds_bytes
  if (h->tune.large_library_solver != 0 || n <= 128 || P > 65535 || lds + 2048 > size_t(h->max_lds)) return toa_large_solve_each(h, dtype, n, P, H, g, 1.0, dx, ok);
  auto run = [&](auto tag) -> int {
    using T = decltype(tag);
    toa::LargeArgs<T> a;
    std::memset(static_cast<void*>(&a), 0, sizeof(a));
    a.n = n;
    a.P = P;
    a.work = static_cast<T*>(H);
    a.rhs = const_cast<T*>(static_cast<const T*>(g));
    a.ddx = static_cast<T*>(dx);
    a.dok = ok;
    a.dmask = h->solve_mask;
    a.dmask_stride = (long long)h->solve_mask_stride;
    const size_t chol_lds = toa::chol_solve_lds_bytes<T>(n);
    if (int rc = toa::ensure_lds_attr(h, (const void*)toa::large_chol_solve_kernel<T>, chol_lds)) return rc;
    if (int rc = toa::ensure_lds_attr(h, (const void*)toa::large_chol_solve_kernel<T, false>, chol_lds)) return rc;
    if (h->tune.large_chol_no_lookahead) hipLaunchKernelGGL((toa::large_chol_solve_kernel<T, false>), dim3(unsigned(P)), dim3(toa::kCholThreads), chol_lds, h->stream, a);
    else hipLaunchKernelGGL(toa::large_chol_solve_kernel<T>, dim3(unsigned(P)), dim3(toa::kCholThreads), chol_lds, h->stream, a);
    HIP_TRY(hipGetLastError());
    return TOA_OK;
  };
  return dtype == TOA_F32 ? run(float()) : run(double());
}

int toa_large_solve(toa_handle h, int dtype, int n, int64_t P, const void* H, const void* g, double scale, void* dx,
                    int32_t* ok) {
  const bool force_lib = h->tune.large_library_solver != 0;
  const size_t chol_lds = ((size_t(n) * (n | 1) + 16) * (dtype == TOA_F32 ? 4 : 8) + 15) & ~size_t(15);
  const size_t chol2_lds = (size_t(32) * 36 + size_t(n) * 37 + 96) * (dtype == TOA_F32 ? 4 : 8) + 64;   // chol_solve_lds_bytes
  const bool own2 = n > 128 && P <= 65535 && chol2_lds + 2048 <= size_t(h->max_lds);   // the one-workgroup blocked Cholesky (fp32: n <= 1024, fp64: n <= 512)
  if (!force_lib && ((n <= 128 && chol_lds + 4096 <= size_t(h->max_lds)) || own2)) {  // the workgroup LDL^T (ldlt_wg.hpp) / blocked Cholesky; the library beyond
    if (dtype == TOA_F32)
      return toa::large_solve_own_t<float>(h, n, P, static_cast<const float*>(H), static_cast<const float*>(g), scale, static_cast<float*>(dx), ok);
    return toa::large_solve_own_t<double>(h, n, P, static_cast<const double*>(H), static_cast<const double*>(g), scale, static_cast<double*>(dx), ok);
  }
  toa::RocApi& api = toa::roc_api();
  if (!api.ok) return toa_fail(TOA_E_UNSUPPORTED, "large-n solve needs rocSOLVER: " + api.err);
  if (dtype == TOA_F32)
    return toa::large_solve_t<float>(h, api, n, P, static_cast<const float*>(H), static_cast<const float*>(g), scale,
                                     static_cast<float*>(dx), ok);
  return toa::large_solve_t<double>(h, api, n, P, static_cast<const double*>(H), static_cast<const double*>(g), scale,
                                    static_cast<double*>(dx), ok);
}

int toa_large_solve_unchecked(toa_handle h, int dtype, int n, int64_t P, const void* H, const void* g, double scale, void* dx, int32_t* ok) {
  toa::RocApi& api = toa::roc_api();
  if (!api.ok) return toa_fail(TOA_E_UNSUPPORTED, "use_ldlt = false beyond one wavefront needs rocSOLVER (general LU): " + api.err);
  if (dtype == TOA_F32)
    return toa::large_solve_lu_t<float>(h, api, n, P, static_cast<const float*>(H), static_cast<const float*>(g), scale, static_cast<float*>(dx), ok);
  return toa::large_solve_lu_t<double>(h, api, n, P, static_cast<const double*>(H), static_cast<const double*>(g), scale, static_cast<double*>(dx), ok);
}

int toa_large_solve_each(toa_handle h, int dtype, int n, int64_t P, const void* H, const void* g, double scale, void* dx, int32_t* ok) {
  {  // our own kernels are batch-independent by construction (one workgroup per matrix, fixed-order sums): one launch for all
    const bool force_lib = h->tune.large_library_solver != 0;
    const size_t chol2_lds = (size_t(32) * 36 + size_t(n) * 37 + 96) * (dtype == TOA_F32 ? 4 : 8) + 64;
    if (!force_lib && (n <= 128 || (P <= 65535 && chol2_lds + 2048 <= size_t(h->max_lds)))) return toa_large_solve(h, dtype, n, P, H, g, scale, dx, ok);
  }
  toa::RocApi& api = toa::roc_api();
  if (!api.ok) return toa_fail(TOA_E_UNSUPPORTED, "large-n solve needs rocSOLVER: " + api.err);
  if (dtype == TOA_F32)
    return toa::large_solve_each_t<float>(h, api, n, P, static_cast<const float*>(H), static_cast<const float*>(g), scale, static_cast<float*>(dx), ok);
  return toa::large_solve_each_t<double>(h, api, n, P, static_cast<const double*>(H), static_cast<const double*>(g), scale, static_cast<double*>(dx), ok);
}

int toa_large_lm_run(toa_handle h, int dtype, int n, int m, int64_t P, const void* data, void* x, const toa_options* options,
                     const toa_results* results, uint64_t* counters) {
  // 64 <= n <= 128: the whole loop in one persistent kernel, Gram on the matrix cores (large_fused.hip)
  // (use_ldlt = false needs the library's general LU: the launch-per-stage pipeline below, for every n >= 64)
  // ... unless the batch is a FEW huge problems: a workgroup per problem leaves the chip idle and lasts as long as ONE problem
  // (~0.35 us per row at n = 128), while the pipeline below splits the rows of every problem over the compute units (rows
  // kernel: one workgroup per CU and problem; Gram: up to 256 row chunks) at a latency floor of ~1.2 ms per solve.  Measured
  // crossover, fp32 (ms per solve, pipeline / one kernel): 1 x 65 536 x n = 128: 2.0 / 19.1; 1 x 20 000 x 96: 1.2 / 3.3;
  // 4 x 16 384 x 128: 1.6 / 7.5; 32 x 8 192 x 128: 2.0 / 4.0; 64 x 4 096 x 128: 1.9 / 2.7; but 24 x 2 048 x 128: 1.5 / 1.25,
  // 64 x 3 000 x 96: 1.45 / 1.14, 128 x 4 096 x 128: 2.8 / 2.5 — so: enough rows per problem to outlast the floor, and a batch
  // small enough that the pipeline's own data passes stay under it.  The M-estimator lives in the one-kernel form only.
  // (fp32 with 16-byte rows: every stage of the pipeline is then a kernel of this library — no rocBLAS / rocSOLVER load, no per-pass read-back)
  const bool few_huge = int64_t(m) * n >= 393216 && P * int64_t(m) * n <= (int64_t(1) << 25) && !h->tune.wide_no_autosplit &&
                        dtype == TOA_F32 && n % 4 == 0 && (int64_t(m) * (n + 1)) % 4 == 0 && reinterpret_cast<uintptr_t>(data) % 16 == 0;
  // (a loss on fp64 rows beyond n = 96: the one-kernel form's two half-tile passes have no M-estimator variant — the pipeline does)
  const bool loss_needs_pipeline = h->loss != TOA_LOSS_L2 && dtype == TOA_F64 && n > 96;
  if (options->use_ldlt && !few_huge && !loss_needs_pipeline && toa_large_fused_eligible(h, dtype, n, m))
    return toa_large_fused_lm_run(h, dtype, n, m, P, data, x, options, results, counters);
  // The pipeline's kernels index problems through grid.y (65 535): a larger batch goes through it slice by slice — the
  // problems are independent, so the slices are just shorter batches (same bits), and the workspace is sized for one slice.
  constexpr int64_t kSlice = 65535;
  const size_t es = dtype == TOA_F32 ? 4 : 8;
  for (int64_t p0 = 0; p0 < P || p0 == 0; p0 += kSlice) {
    const int64_t Ps = std::min<int64_t>(kSlice, P - p0);
    toa_results r = *results;
    auto adv = [&](auto*& ptr, size_t per) { if (ptr) ptr += size_t(p0) * per; };
    adv(r.stop_reason, 1); adv(r.num_iters, 1); adv(r.num_failures, 1); adv(r.num_consec_failures, 1);
    adv(r.final_cost, 1); adv(r.final_num_residuals, 1); adv(r.final_rerr_dec, 1); adv(r.final_hessian, size_t(n) * n);
    adv(r.errs, size_t(r.hist_stride)); adv(r.deltas2, size_t(r.hist_stride)); adv(r.successes, size_t(r.hist_stride));
    adv(r.final_inlier_ratio, 1);
    const char* d = static_cast<const char*>(data) + size_t(p0) * size_t(m) * (size_t(n) + 1) * es;
    char* xs = static_cast<char*>(x) + size_t(p0) * size_t(n) * es;
    int rc;
    if (dtype == TOA_F32)
      rc = toa::large_lm_run_t<float>(h, n, m, Ps, reinterpret_cast<const float*>(d), reinterpret_cast<float*>(xs), *options, r, counters);
    else
      rc = toa::large_lm_run_t<double>(h, n, m, Ps, reinterpret_cast<const double*>(d), reinterpret_cast<double*>(xs), *options, r, counters);
    if (rc != TOA_OK) return rc;
    if (P <= 0) break;
  }
  return TOA_OK;
}

// The stepping form at n >= 64 (toa_lm_begin / toa_lm_step / toa_lm_stop, capi.hip): mode 1 / 2 / 3 on the launch-per-stage
// pipeline — every n >= 64, either solver flavour — with what outlives a pass kept in the caller's state block.
size_t toa_large_state_bytes(int dtype, int n, int64_t P) {
  return dtype == TOA_F32 ? toa::LargeStateLayout<float>(n, P).total : toa::LargeStateLayout<double>(n, P).total;
}

int toa_large_lm_step(toa_handle h, int dtype, int n, int m, int64_t P, const void* data, void* x, const toa_options* options,
                      const toa_results* results, uint64_t* counters, int mode, void* state, int32_t* active_dev,
                      const int32_t* stop_request) {
  if (P > 65535) return toa_fail(TOA_E_UNSUPPORTED, "stepping form at n >= 64: at most 65 535 problems per call");
  if (dtype == TOA_F32)
    return toa::large_lm_run_t<float>(h, n, m, P, static_cast<const float*>(data), static_cast<float*>(x), *options, *results,
                                      counters, mode, state, active_dev, stop_request);
  return toa::large_lm_run_t<double>(h, n, m, P, static_cast<const double*>(data), static_cast<double*>(x), *options, *results,
                                     counters, mode, state, active_dev, stop_request);
}

// toa_accumulate for the natural layout where the one-launch kernel of large_fused.hip does not reach: n > 128, or an M-estimator
// (one data pass of the pipeline; batches beyond grid.y = 65 535 problems slice by slice like toa_large_lm_run)
int toa_large_accumulate_pipeline(toa_handle h, int dtype, int n, int m, int64_t P, const void* data, const void* x, int want_grad, void* g,
                                  void* H, double* cost, int32_t* nres) {
  toa_options opt;
  toa_options_default(&opt);
  toa_results res{};
  constexpr int64_t kSlice = 65535;
  const size_t es = dtype == TOA_F32 ? 4 : 8;
  for (int64_t p0 = 0; p0 < P; p0 += kSlice) {
    const int64_t Ps = std::min<int64_t>(kSlice, P - p0);
    const char* d = static_cast<const char*>(data) + size_t(p0) * size_t(m) * (size_t(n) + 1) * es;
    char* xs = const_cast<char*>(static_cast<const char*>(x)) + size_t(p0) * size_t(n) * es;
    char* gs = g ? static_cast<char*>(g) + size_t(p0) * size_t(n) * es : nullptr;
    char* Hs = H ? static_cast<char*>(H) + size_t(p0) * size_t(n) * size_t(n) * es : nullptr;
    int32_t* nr = nres ? nres + p0 : nullptr;
    int rc;
    if (dtype == TOA_F32) {
      const toa::LargeSeamOut<float> o{reinterpret_cast<float*>(gs), reinterpret_cast<float*>(Hs), cost + p0, nr, want_grad};
      rc = toa::large_lm_run_t<float>(h, n, m, Ps, reinterpret_cast<const float*>(d), reinterpret_cast<float*>(xs), opt, res, nullptr, 4, nullptr, nullptr, nullptr, &o);
    } else {
      const toa::LargeSeamOut<double> o{reinterpret_cast<double*>(gs), reinterpret_cast<double*>(Hs), cost + p0, nr, want_grad};
      rc = toa::large_lm_run_t<double>(h, n, m, Ps, reinterpret_cast<const double*>(d), reinterpret_cast<double*>(xs), opt, res, nullptr, 4, nullptr, nullptr, nullptr, &o);
    }
    if (rc != TOA_OK) return rc;
  }
  return TOA_OK;
}

int toa_large_step_info(toa_handle h, int dtype, int n, int64_t P, const void* state, double* err, double* dx2, double* g2,
                        void* dx_out, void* g_out) {
  if (dtype == TOA_F32)
    hipLaunchKernelGGL(toa::large_step_info_kernel<float>, dim3(unsigned(P)), dim3(256), 0, h->stream, static_cast<const char*>(state),
                       (long long)P, n, err, dx2, g2, static_cast<float*>(dx_out), static_cast<float*>(g_out));
  else
    hipLaunchKernelGGL(toa::large_step_info_kernel<double>, dim3(unsigned(P)), dim3(256), 0, h->stream, static_cast<const char*>(state),
                       (long long)P, n, err, dx2, g2, static_cast<double*>(dx_out), static_cast<double*>(g_out));
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}

int toa_large_step_log(toa_handle h, int dtype, int n, int64_t P, const void* state, double* lambda, int32_t* nres, int32_t* ninl) {
  const unsigned grid = unsigned((P + 255) / 256);
  if (dtype == TOA_F32)
    hipLaunchKernelGGL(toa::large_step_log_kernel<float>, dim3(grid), dim3(256), 0, h->stream, static_cast<const char*>(state), (long long)P, n, lambda, nres, ninl);
  else
    hipLaunchKernelGGL(toa::large_step_log_kernel<double>, dim3(grid), dim3(256), 0, h->stream, static_cast<const char*>(state), (long long)P, n, lambda, nres, ninl);
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}

int toa_large_inv_cov(toa_handle h, int dtype, int n, int64_t P, const void* H, void* C, int32_t* ok) {
  toa::RocApi& api = toa::roc_api();
  if (!api.ok) return toa_fail(TOA_E_UNSUPPORTED, "large-n covariance needs rocSOLVER: " + api.err);
  if (dtype == TOA_F32) return toa::large_inv_t<float>(h, api, n, P, static_cast<const float*>(H), static_cast<float*>(C), ok);
  return toa::large_inv_t<double>(h, api, n, P, static_cast<const double*>(H), static_cast<double*>(C), ok);
}
