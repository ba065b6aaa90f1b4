// Microbenchmark (MI355X): do f32 MFMA and VALU instructions of co-resident waves overlap, and what is the shader
// clock under that load?  Each wave runs a fixed instruction mix; s_memtime (shader clock) and s_memrealtime
// (100 MHz constant) bracket the loop.  Build: hipcc --offload-arch=gfx950 -O3 issue_probe.hip -o issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int NMFMA, int NVALU>
__global__ void __launch_bounds__(256) mix(float* out, unsigned long long* clk, int iters) {
  f4 acc[6];
  for (int i = 0; i < 6; ++i) acc[i] = f4{0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  const unsigned long long t0 = __builtin_readcyclecounter();       // s_memtime
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < NMFMA; ++k)
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[k % 6]) : "v"(a), "v"(b));
#pragma unroll
    for (int k = 0; k < NVALU; ++k) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[k % 8]) : "v"(a), "v"(b));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  float s = 0;
  for (int i = 0; i < 6; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NM, int NV>
void run(const char* name, int wgs, int iters) {
  float* out; unsigned long long* clk;
  hipMalloc(&out, size_t(wgs) * 256 * 4);
  hipMalloc(&clk, size_t(wgs) * 16);
  hipLaunchKernelGGL((mix<NM, NV>), dim3(wgs), dim3(256), 0, 0, out, clk, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((mix<NM, NV>), dim3(wgs), dim3(256), 0, 0, out, clk, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(size_t(wgs) * 2);
  hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
  double cyc = 0, real = 0;
  for (int i = 0; i < wgs; ++i) { cyc += h[2 * i]; real += h[2 * i + 1]; }
  cyc /= wgs; real /= wgs;
  const double waves_per_simd = double(wgs) / 256.0;  // 4 waves per WG = 1 per SIMD per WG on a CU
  const double per_iter = cyc / iters;
  std::printf("%-28s wgs=%5d  kernel %.3f ms  cycles/iter/wave %.1f  memtime/realtime = %.2f MHz  => SIMD cycles per (wave-iter) %.1f (model: MFMA %d + VALU %d)\n",
              name, wgs, ms, per_iter, cyc / real * 100.0, per_iter / (waves_per_simd < 1 ? 1 : waves_per_simd), NM * 32, NV * 4);
  hipFree(out); hipFree(clk);
}

int main() {
  const int iters = 20000;
  for (int wgs : {256, 1024}) {   // 1 or 4 waves per SIMD
    run<6, 0>("6 MFMA", wgs, iters);
    run<0, 24>("24 VALU", wgs, iters);
    run<6, 24>("6 MFMA + 24 VALU", wgs, iters);
    run<6, 48>("6 MFMA + 48 VALU", wgs, iters);
  }
  return 0;
}
