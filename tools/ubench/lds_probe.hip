// Microbenchmark (MI355X): facts the staged fp32 accumulate pass depends on.
//   T1  buffer_load_dwordx4 ... lds under an EXEC mask of 51 lanes: which LDS bytes are written?
//   T2  ds_read_b64 / b96 / b128 at addresses that are only 4-byte aligned: do they work, what do they cost?
//   T3  LDS reads issued between the MFMAs of a wave: do they extend the MFMA stream?
// Build: hipcc --offload-arch=gfx950 -O3 lds_probe.hip -o lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u3 __attribute__((ext_vector_type(3)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef int i4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned lds_off(const void* p) {
  return unsigned(reinterpret_cast<size_t>((__attribute__((address_space(3))) const char*)(p)));
}
__device__ __forceinline__ i4 make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  i4 r;
  r[0] = __builtin_amdgcn_readfirstlane(int(unsigned(a)));
  r[1] = __builtin_amdgcn_readfirstlane(int(unsigned(a >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane(int(bytes));
  r[3] = 0x00020000;
  return r;
}

// ---- T1 -----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) t1_kernel(const float* src, float* out, int lanes) {
  __shared__ __attribute__((aligned(16))) float buf[1024];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) buf[i] = -1.0f;
  __syncthreads();
  const i4 rsrc = make_rsrc(src, 4096);
  const unsigned voff = lane * 16;
  const unsigned base = unsigned(__builtin_amdgcn_readfirstlane(int(lds_off(buf) + 64)));   // destination: buf + 16 floats
  const unsigned long long mask = lanes >= 64 ? ~0ull : ((1ull << lanes) - 1ull);
  const unsigned mlo = unsigned(__builtin_amdgcn_readfirstlane(int(unsigned(mask)))), mhi = unsigned(__builtin_amdgcn_readfirstlane(int(unsigned(mask >> 32))));
  asm volatile(
      "s_mov_b32 exec_lo, %[mlo]\n\ts_mov_b32 exec_hi, %[mhi]\n\t"
      "s_mov_b32 m0, %[base]\n\ts_nop 4\n\t"
      "buffer_load_dwordx4 %[voff], %[rsrc], 0 offen lds\n\t"
      "s_mov_b64 exec, -1\n\t"
      "s_waitcnt vmcnt(0)"
      : : [mlo] "s"(mlo), [mhi] "s"(mhi), [base] "s"(base), [voff] "v"(voff), [rsrc] "s"(rsrc) : "memory");
  __syncthreads();
  for (int i = lane; i < 1024; i += 64) out[i] = buf[i];
}

// ---- T2 -----------------------------------------------------------------------------------------
template <int W>   // dwords per read: 1, 2, 3, 4
__global__ void __launch_bounds__(256) t2_kernel(float* out, int iters, int misalign_dwords, int stride_dwords) {
  __shared__ __attribute__((aligned(16))) float buf[16384];
  for (int i = threadIdx.x; i < 16384; i += 256) buf[i] = float(i);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned a = lds_off(buf) + unsigned(wave * 4096 * 4 + (lane * stride_dwords + misalign_dwords) * 4);
  float s = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if constexpr (W == 1) { unsigned v; asm volatile("ds_read_b32 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a), "n"(k * 16) : "memory"); s += __uint_as_float(v); }
      if constexpr (W == 2) { u2 v; asm volatile("ds_read_b64 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a), "n"(k * 16) : "memory"); s += __uint_as_float(v[0]) + __uint_as_float(v[1]); }
      if constexpr (W == 3) { u3 v; asm volatile("ds_read_b96 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a), "n"(k * 16) : "memory"); s += __uint_as_float(v[0]) + __uint_as_float(v[2]); }
      if constexpr (W == 4) { u4 v; asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a), "n"(k * 16) : "memory"); s += __uint_as_float(v[0]) + __uint_as_float(v[3]); }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// throughput form: 8 reads in flight per wait
template <int W>
__global__ void __launch_bounds__(256) t2b_kernel(float* out, int iters, int misalign_dwords, int stride_dwords) {
  __shared__ __attribute__((aligned(16))) float buf[16384];
  for (int i = threadIdx.x; i < 16384; i += 256) buf[i] = float(i);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned a = lds_off(buf) + unsigned(wave * 4096 * 4 + (lane * stride_dwords + misalign_dwords) * 4);
  float s = 0;
  for (int it = 0; it < iters; ++it) {
    if constexpr (W == 1) {
      unsigned v[8];
      asm volatile("ds_read_b32 %0, %8 offset:0\n\tds_read_b32 %1, %8 offset:16\n\tds_read_b32 %2, %8 offset:32\n\tds_read_b32 %3, %8 offset:48\n\t"
                   "ds_read_b32 %4, %8 offset:64\n\tds_read_b32 %5, %8 offset:80\n\tds_read_b32 %6, %8 offset:96\n\tds_read_b32 %7, %8 offset:112\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) : "v"(a) : "memory");
      for (int k = 0; k < 8; ++k) s += __uint_as_float(v[k]);
    }
    if constexpr (W == 2) {
      u2 v[8];
      asm volatile("ds_read_b64 %0, %8 offset:0\n\tds_read_b64 %1, %8 offset:16\n\tds_read_b64 %2, %8 offset:32\n\tds_read_b64 %3, %8 offset:48\n\t"
                   "ds_read_b64 %4, %8 offset:64\n\tds_read_b64 %5, %8 offset:80\n\tds_read_b64 %6, %8 offset:96\n\tds_read_b64 %7, %8 offset:112\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) : "v"(a) : "memory");
      for (int k = 0; k < 8; ++k) s += __uint_as_float(v[k][0]) + __uint_as_float(v[k][1]);
    }
    if constexpr (W == 4) {
      u4 v[8];
      asm volatile("ds_read_b128 %0, %8 offset:0\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:32\n\tds_read_b128 %3, %8 offset:48\n\t"
                   "ds_read_b128 %4, %8 offset:64\n\tds_read_b128 %5, %8 offset:80\n\tds_read_b128 %6, %8 offset:96\n\tds_read_b128 %7, %8 offset:112\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) : "v"(a) : "memory");
      for (int k = 0; k < 8; ++k) s += __uint_as_float(v[k][0]) + __uint_as_float(v[k][3]);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ---- T3 -----------------------------------------------------------------------------------------
template <int NLDS, int NVALU>
__global__ void __launch_bounds__(256) t3_kernel(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float buf[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) buf[i] = float(i & 15);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned a = lds_off(buf) + unsigned(wave * 2048 * 4 + lane * 4);
  f4 acc[6];
  for (int i = 0; i < 6; ++i) acc[i] = f4{0, 0, 0, 0};
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  unsigned w[8];
  float v[8];
  for (int i = 0; i < 8; ++i) { w[i] = 0; v[i] = x + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[k]) : "v"(x), "v"(y));
      if (k < NLDS) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(w[k]) : "v"(a), "n"(k * 256) : "memory");
      if (k + 6 < NLDS) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(w[(k + 6) & 7]) : "v"(a), "n"(k * 256 + 2048) : "memory");
    }
    if (NLDS) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7])::"memory");
#pragma unroll
    for (int k = 0; k < NVALU; ++k) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[k % 8]) : "v"(x), "v"(y));
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  float s = 0;
  for (int i = 0; i < 6; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += __uint_as_float(w[i]) + v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
float time_kernel(F launch) {
  launch();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* out; hipMalloc(&out, 1 << 22);
  // T1
  {
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = float(i);
    float* src; hipMalloc(&src, 4096); hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
    for (int lanes : {64, 51, 13}) {
      hipLaunchKernelGGL(t1_kernel, dim3(1), dim3(64), 0, 0, src, out, lanes);
      std::vector<float> r(1024);
      hipMemcpy(r.data(), out, 4096, hipMemcpyDeviceToHost);
      int first = -1, last = -1, bad = 0;
      for (int i = 0; i < 1024; ++i) if (r[i] != -1.0f) { if (first < 0) first = i; last = i; if (r[i] != float(i - 16)) ++bad; }
      std::printf("T1 lanes=%d: written floats [%d, %d] (expect [16, %d]) mismatches=%d\n", lanes, first, last, 16 + lanes * 4 - 1, bad);
    }
  }
  // T2 correctness + latency (serial) + throughput
  {
    const int wgs = 1024, iters = 2000;
    for (int mis = 0; mis < 4; ++mis) {
      // correctness of one read: lane 0 of wave 0 reads buf[mis ...]
      float ms1 = time_kernel([&] { hipLaunchKernelGGL((t2_kernel<1>), dim3(wgs), dim3(256), 0, 0, out, iters, mis, 1); });
      float ms2 = time_kernel([&] { hipLaunchKernelGGL((t2_kernel<2>), dim3(wgs), dim3(256), 0, 0, out, iters, mis, 2); });
      float ms3 = time_kernel([&] { hipLaunchKernelGGL((t2_kernel<3>), dim3(wgs), dim3(256), 0, 0, out, iters, mis, 4); });
      float ms4 = time_kernel([&] { hipLaunchKernelGGL((t2_kernel<4>), dim3(wgs), dim3(256), 0, 0, out, iters, mis, 4); });
      float v4; hipMemcpy(&v4, out, 4, hipMemcpyDeviceToHost);
      std::printf("T2 serial misalign=%d dwords: b32 %.3f  b64 %.3f  b96 %.3f  b128 %.3f ms  (b128 lane0 sum/iters = %.1f, expect %.1f)\n", mis, ms1, ms2, ms3, ms4,
                  v4 / iters, [&] { double e = 0; for (int k = 0; k < 8; ++k) e += (mis + 4 * k) + (mis + 4 * k + 3); return e; }());
      float t1 = time_kernel([&] { hipLaunchKernelGGL((t2b_kernel<1>), dim3(wgs), dim3(256), 0, 0, out, iters, mis, 1); });
      float t2 = time_kernel([&] { hipLaunchKernelGGL((t2b_kernel<2>), dim3(wgs), dim3(256), 0, 0, out, iters, mis, 2); });
      float t4 = time_kernel([&] { hipLaunchKernelGGL((t2b_kernel<4>), dim3(wgs), dim3(256), 0, 0, out, iters, mis, 4); });
      std::printf("T2 pipelined misalign=%d: b32 %.3f  b64 %.3f  b128 %.3f ms  (per 8 reads x %d iters x 4 waves/SIMD)\n", mis, t1, t2, t4, iters);
    }
    // odd row stride 51 dwords, b32 vs b64 (the row-per-lane read pattern)
    float s1 = time_kernel([&] { hipLaunchKernelGGL((t2b_kernel<1>), dim3(wgs), dim3(256), 0, 0, out, iters, 0, 51); });
    float s2 = time_kernel([&] { hipLaunchKernelGGL((t2b_kernel<2>), dim3(wgs), dim3(256), 0, 0, out, iters, 0, 51); });
    float s3 = time_kernel([&] { hipLaunchKernelGGL((t2b_kernel<1>), dim3(wgs), dim3(256), 0, 0, out, iters, 0, 13); });
    std::printf("T2 stride 51: b32 %.3f  b64(unaligned on odd lanes) %.3f ; stride 13 b32 %.3f ms\n", s1, s2, s3);
  }
  // T3
  {
    const int iters = 20000;
    for (int wgs : {256, 768}) {
      float a = time_kernel([&] { hipLaunchKernelGGL((t3_kernel<0, 0>), dim3(wgs), dim3(256), 0, 0, out, iters); });
      float b = time_kernel([&] { hipLaunchKernelGGL((t3_kernel<6, 0>), dim3(wgs), dim3(256), 0, 0, out, iters); });
      float c = time_kernel([&] { hipLaunchKernelGGL((t3_kernel<12, 0>), dim3(wgs), dim3(256), 0, 0, out, iters); });
      float d = time_kernel([&] { hipLaunchKernelGGL((t3_kernel<0, 12>), dim3(wgs), dim3(256), 0, 0, out, iters); });
      float e = time_kernel([&] { hipLaunchKernelGGL((t3_kernel<12, 12>), dim3(wgs), dim3(256), 0, 0, out, iters); });
      std::printf("T3 wgs=%d (%d waves/SIMD): 6 MFMA %.3f | +6 ds_read %.3f | +12 ds_read %.3f | +12 VALU %.3f | +12 ds_read +12 VALU %.3f ms\n", wgs, wgs / 256, a, b, c, d, e);
    }
  }
  return 0;
}
