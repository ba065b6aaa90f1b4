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
#include <dlfcn.h>

#include <mutex>
#include <string>

#include "kernels.hpp"

namespace toa {
namespace {

using rb_handle = void*;
constexpr int kFillUpper = 121;  // rocblas_fill_upper (rocblas-types.h)

struct RocApi {
  int (*create)(rb_handle*) = nullptr;
  int (*destroy)(rb_handle) = nullptr;
  int (*set_stream)(rb_handle, hipStream_t) = nullptr;
  int (*spotrf)(rb_handle, int, int, float*, int, int64_t, int*, int) = nullptr;
  int (*dpotrf)(rb_handle, int, int, double*, int, int64_t, int*, int) = nullptr;
  int (*spotrs)(rb_handle, int, int, int, float*, int, int64_t, float*, int, int64_t, int) = nullptr;
  int (*dpotrs)(rb_handle, int, int, int, double*, int, int64_t, double*, int, int64_t, int) = nullptr;
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
    api.ok = api.err.empty();
  });
  return api;
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
                                                           T* __restrict__ dx, int32_t* __restrict__ ok, const int n) {
  const size_t p = blockIdx.x;
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
  if (need > h->scratch_bytes) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->scratch) (void)hipFree(h->scratch);
    h->scratch = nullptr;
    h->scratch_bytes = 0;
    HIP_TRY(hipMalloc(&h->scratch, need));
    h->scratch_bytes = need;
  }
  char* base = static_cast<char*>(h->scratch);
  T* work = reinterpret_cast<T*>(base);
  T* rhs = reinterpret_cast<T*>(base + b_work);
  int* info = reinterpret_cast<int*>(base + b_work + b_rhs);
  if (!h->blas) {
    if (api.create(&h->blas) != 0) return toa_fail(TOA_E_HIP, "rocblas_create_handle failed");
    h->blas_destroy = api.destroy;
  }
  if (api.set_stream(h->blas, h->stream) != 0) return toa_fail(TOA_E_HIP, "rocblas_set_stream failed");
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

}  // namespace
}  // namespace toa

int toa_large_solve(toa_handle h, int dtype, int n, int64_t P, const void* H, const void* g, double scale, void* dx,
                    int32_t* ok) {
  toa::RocApi& api = toa::roc_api();
  if (!api.ok) return toa_fail(TOA_E_UNSUPPORTED, "large-n solve needs rocSOLVER: " + api.err);
  if (dtype == TOA_F32)
    return toa::large_solve_t<float>(h, api, n, P, static_cast<const float*>(H), static_cast<const float*>(g), scale,
                                     static_cast<float*>(dx), ok);
  return toa::large_solve_t<double>(h, api, n, P, static_cast<const double*>(H), static_cast<const double*>(g), scale,
                                    static_cast<double*>(dx), ok);
}
