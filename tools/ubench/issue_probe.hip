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

// ---- second probe: issue cost of individual VALU flavours at 4 waves / SIMD (ns per wave-instruction per SIMD)
template <int KIND>
__global__ void __launch_bounds__(256) flavour(float* out, int iters) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  f2 p[8];
  float v[16];
  for (int i = 0; i < 8; ++i) p[i] = f2{a + i, a - i};
  for (int i = 0; i < 16; ++i) v[i] = a + i;
  f2 ab = {a, b};
  float sres = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      if (KIND == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[k % 16]) : "v"(a), "v"(b));
      if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[k % 8]) : "v"(ab), "v"(ab));
      if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %1, s[20:21], %0" : "+v"(p[k % 8]) : "v"(ab) : "s20", "s21");
      if (KIND == 3) asm volatile("v_add_f32_dpp %0, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(v[k % 16]) : "v"(a));
      if (KIND == 4) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(v[k % 16]) : "s20");
      if (KIND == 5) asm volatile("v_sin_f32 %0, %1" : "=v"(v[k % 16]) : "v"(a));
      if (KIND == 6) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(*(double*)&p[k % 8]) : "v"(*(double*)&ab));
      if (KIND == 7) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(v[k % 16]) : "v"(a), "v"(b) : "vcc");
      if (KIND == 8) asm volatile("v_fmac_f32 %0, s20, %1" : "+v"(v[k % 16]) : "v"(b) : "s20");
    }
  }
  float s = sres;
  for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND>
void runf(const char* name) {
  const int wgs = 1024, iters = 4000;
  float* out;
  hipMalloc(&out, size_t(wgs) * 256 * 4);
  hipLaunchKernelGGL((flavour<KIND>), dim3(wgs), dim3(256), 0, 0, out, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((flavour<KIND>), dim3(wgs), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // 4 waves per SIMD, each issuing iters*32 instructions
  std::printf("%-34s %.3f ms  -> %.2f ns per wave-instruction per SIMD\n", name, ms, ms * 1e6 / (4.0 * iters * 32));
  hipFree(out);
}

int main() {
  runf<0>("v_fmac_f32");
  runf<8>("v_fmac_f32 (SGPR operand)");
  runf<1>("v_pk_fma_f32");
  runf<2>("v_pk_fma_f32 (SGPR pair operand)");
  runf<3>("v_add_f32_dpp row_ror");
  runf<4>("v_readlane_b32");
  runf<5>("v_sin_f32");
  runf<6>("v_fma_f64");
  runf<7>("v_cndmask_b32");
  const int iters = 20000;
  for (int wgs : {256, 1024}) {   // 1 or 4 waves per SIMD
    run<6, 0>("6 MFMA", wgs, iters);
    run<0, 24>("24 VALU", wgs, iters);
    run<6, 24>("6 MFMA + 24 VALU", wgs, iters);
    run<6, 48>("6 MFMA + 48 VALU", wgs, iters);
  }
  return 0;
}
