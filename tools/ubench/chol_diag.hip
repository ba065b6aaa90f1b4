// The 32 x 32 diagonal block of large_chol_solve_kernel alone: one wavefront, the block in LDS, REPS factorisations back to back
// (the block restored from registers in between), shader cycles per block by s_memtime.  The chain of 32 pivots is the longest
// serial piece of the blocked Cholesky (round 5: 8.6 us per block fp32, 12 us fp64, of a 14-17 us step).
//   build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Itinyopt_amd/csrc [-DTOA_DIAG_VARIANT=k] tools/ubench/chol_diag.hip -o tools/ubench/chol_diag
//   run:    tools/ubench/chol_diag
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "chol_diag.hpp"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <typename T>
__global__ void __launch_bounds__(64) diag_kernel(const T* __restrict__ src, T* __restrict__ dst, T* __restrict__ rsd, long long* __restrict__ cyc, int reps) {
  constexpr int B = 32, LS = 36;
  __shared__ __attribute__((aligned(16))) T Ld[B * LS];
  __shared__ T rs[64];
  __shared__ int fail;
  const int lane = threadIdx.x;
  T keep[B];
  for (int c = 0; c < B; ++c) keep[c] = lane < B ? src[lane * B + c] : T(0);
  if (lane == 0) fail = 0;
  long long total = 0;
  using L3 = __attribute__((address_space(3))) T*;
  for (int rep = 0; rep < reps; ++rep) {
    if (lane < B)
      for (int c = 0; c < B; ++c) Ld[lane * LS + c] = keep[c];
    __syncthreads();
    const long long t0 = clock64();
    toa::chol_diag_block<T, LS>((L3)Ld, (L3)rs, (__attribute__((address_space(3))) int*)&fail, B);
    __builtin_amdgcn_s_waitcnt(0);
    total += clock64() - t0;
    __syncthreads();
  }
  if (lane < B) {
    for (int c = 0; c < B; ++c) dst[lane * B + c] = c <= lane ? Ld[lane * LS + c] : T(0);
    rsd[lane] = rs[lane];
  }
  if (lane == 0) { cyc[0] = total / reps; cyc[1] = fail; }
}

template <typename T>
void run(const char* name) {
  const int B = 32;
  std::vector<T> h(B * B);
  std::vector<double> m(B * (B + 8));
  srand(7);
  for (auto& v : m) v = rand() / double(RAND_MAX) * 2 - 1;
  for (int i = 0; i < B; ++i)
    for (int j = 0; j < B; ++j) {
      double s = i == j ? 0.5 : 0.0;
      for (int k = 0; k < B + 8; ++k) s += m[i * (B + 8) + k] * m[j * (B + 8) + k] / B;
      h[i * B + j] = T(s);
    }
  T *src, *dst, *rs;
  long long* cyc;
  CHECK(hipMalloc(&src, sizeof(T) * B * B)); CHECK(hipMalloc(&dst, sizeof(T) * B * B)); CHECK(hipMalloc(&rs, sizeof(T) * 64)); CHECK(hipMalloc(&cyc, 16));
  CHECK(hipMemcpy(src, h.data(), sizeof(T) * B * B, hipMemcpyHostToDevice));
  for (int it = 0; it < 2; ++it) { hipLaunchKernelGGL(diag_kernel<T>, dim3(1), dim3(64), 0, 0, src, dst, rs, cyc, 200); CHECK(hipDeviceSynchronize()); }
  std::vector<T> L(B * B);
  long long c[2];
  CHECK(hipMemcpy(L.data(), dst, sizeof(T) * B * B, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost));
  double err = 0, sum = 0;
  for (int i = 0; i < B; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = 0;
      for (int k = 0; k <= j; ++k) s += double(L[i * B + k]) * double(L[j * B + k]);
      err = std::fmax(err, std::fabs(s - double(h[i * B + j])));
      sum += double(L[i * B + j]) * (1 + i + 3 * j);
    }
  std::printf("%s: %lld shader cycles per 32 x 32 block (%.2f us at 2.4 GHz), fail %lld, |L L^T - A| %.3g, checksum %.17g\n", name, c[0], c[0] / 2400.0, c[1], err, sum);
}

int main() {
  run<float>("fp32");
  run<double>("fp64");
  return 0;
}
