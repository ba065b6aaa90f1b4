// Infinity-Cache (LLC, 256 MiB memory-side) read ceiling on gfx950: a STREAM-like read of a buffer of S MB, repeated back to
// back, for S below / around / above the cache size.  VERDICT r04 #1 go / no-go (i): does a working set of ~200 MB that is
// re-read cyclically come from the die instead of HBM, and at what rate?
//   build:  hipcc --offload-arch=gfx950 -O3 tools/ubench/llc_probe.hip -o tools/ubench/llc_probe
//   run:    tools/ubench/llc_probe [reps]
// Three access shapes:
//   sweep    every workgroup grid-strides over the whole buffer (the bench's read-ceiling kernel), default cache policy
//   sweep_nt the same with nontemporal loads (what toa_hbm_read_probe uses)
//   owner    wave w streams ITS contiguous 408 000-byte slice (a C4 problem) start to end, `reps` times in a row — the access
//            pattern of a problem re-read by its owner iteration after iteration, with S / 408 000 problems in flight
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <bool NT>
__global__ void __launch_bounds__(256) sweep_kernel(const u32x4* __restrict__ src, size_t n16, unsigned* __restrict__ sink) {
  const size_t stride = size_t(gridDim.x) * 256;
  size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
  unsigned acc = 0;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    u32x4 a, b, c, d;
    if (NT) {
      a = __builtin_nontemporal_load(src + i); b = __builtin_nontemporal_load(src + i + stride);
      c = __builtin_nontemporal_load(src + i + 2 * stride); d = __builtin_nontemporal_load(src + i + 3 * stride);
    } else {
      a = src[i]; b = src[i + stride]; c = src[i + 2 * stride]; d = src[i + 3 * stride];
    }
    acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
  }
  for (; i < n16; i += stride) { const u32x4 a = src[i]; acc ^= a.x ^ a.y ^ a.z ^ a.w; }
  if (acc == 0x9E3779B9u) *sink = acc;
}

// wave-per-slice: slice = 408 000 bytes = 25 500 x 16 B; a wave reads 64 x 16 B = 1 KB per load instruction, 4 in flight
__global__ void __launch_bounds__(256) owner_kernel(const u32x4* __restrict__ src, int slices, int passes, int waves_per_slice,
                                                    unsigned* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  const int gw = (blockIdx.x * 256 + threadIdx.x) >> 6;          // global wave
  const int s = gw / waves_per_slice, part = gw % waves_per_slice;
  if (s >= slices) return;
  const size_t per = 25500;                                      // 16-byte words per slice
  const size_t lo = per * part / waves_per_slice, hi = per * (part + 1) / waves_per_slice;
  const u32x4* base = src + size_t(s) * per;
  unsigned acc = 0;
  for (int r = 0; r < passes; ++r) {
    size_t i = lo + lane;
    for (; i + 192 < hi; i += 256) {
      const u32x4 a = base[i], b = base[i + 64], c = base[i + 128], d = base[i + 192];
      acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < hi; i += 64) { const u32x4 a = base[i]; acc ^= a.x ^ a.y ^ a.z ^ a.w; }
    asm volatile("" : "+v"(acc));
  }
  if (acc == 0x9E3779B9u) *sink = acc;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? std::atoi(argv[1]) : 20;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  std::printf("# %s, %d CUs, L2 %d KB\n", prop.name, cus, prop.l2CacheSize / 1024);
  const size_t maxb = size_t(4096) << 20;
  void* buf = nullptr;
  unsigned* sink = nullptr;
  CHECK(hipMalloc(&buf, maxb));
  CHECK(hipMalloc(&sink, 256));
  CHECK(hipMemset(buf, 0x5a, maxb));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int sizes_mb[] = {16, 32, 64, 96, 128, 160, 192, 208, 224, 256, 288, 320, 384, 512, 1024, 2048, 4096};
  std::printf("# sweep: whole-buffer grid-stride read, %d back-to-back launches after one warm launch; GB/s\n", reps);
  std::printf("# %8s %12s %12s\n", "MB", "default", "nontemporal");
  for (int mb : sizes_mb) {
    const size_t n16 = (size_t(mb) << 20) / 16;
    double gbs[2];
    for (int nt = 0; nt < 2; ++nt) {
      auto launch = [&]() {
        if (nt) hipLaunchKernelGGL(sweep_kernel<true>, dim3(cus * 16), dim3(256), 0, 0, (const u32x4*)buf, n16, sink);
        else hipLaunchKernelGGL(sweep_kernel<false>, dim3(cus * 16), dim3(256), 0, 0, (const u32x4*)buf, n16, sink);
      };
      launch();
      CHECK(hipEventRecord(e0, 0));
      for (int r = 0; r < reps; ++r) launch();
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      gbs[nt] = double(n16) * 16 * reps / (ms * 1e-3) * 1e-9;
    }
    std::printf("  %8d %12.0f %12.0f\n", mb, gbs[0], gbs[1]);
  }
  // owner pattern: `slices` problems of 408 000 B in flight, each streamed `passes` times by its own wave(s), ONE launch.
  // time(passes = 7) - time(passes = 1) = six re-reads; 3072 waves resident on 256 CUs (12 per CU, as the fused kernel).
  std::printf("# owner: slices of 408000 B, each streamed by W wave(s) `passes` times within one launch (3072 waves in all)\n");
  std::printf("# %8s %6s %10s %14s %14s %16s\n", "slices", "W", "MB", "first GB/s", "re-read GB/s", "us / re-read pass");
  const int cfg[][2] = {{3072, 1}, {1536, 2}, {768, 4}, {512, 6}, {384, 8}, {256, 12}, {128, 24}};
  for (auto& c : cfg) {
    const int slices = c[0], W = c[1];
    const int waves = slices * W, grid = (waves + 3) / 4;
    auto run = [&](int passes) {
      hipLaunchKernelGGL(owner_kernel, dim3(grid), dim3(256), 0, 0, (const u32x4*)buf, slices, passes, W, sink);   // warm
      float best = 1e30f;
      for (int t = 0; t < 5; ++t) {
        // flush: read 1 GB of something else so that the first pass really comes from HBM
        hipLaunchKernelGGL(sweep_kernel<false>, dim3(cus * 16), dim3(256), 0, 0, (const u32x4*)((char*)buf + (size_t(2048) << 20)), (size_t(1024) << 20) / 16, sink);
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(owner_kernel, dim3(grid), dim3(256), 0, 0, (const u32x4*)buf, slices, passes, W, sink);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      return best;
    };
    const float t1 = run(1), t7 = run(7);
    const double bytes = double(slices) * 408000.0;
    std::printf("  %8d %6d %10.1f %14.0f %14.0f %16.1f\n", slices, W, bytes / 1048576.0, bytes / (t1 * 1e-3) * 1e-9,
                bytes * 6 / ((t7 - t1) * 1e-3) * 1e-9, (t7 - t1) / 6 * 1e3);
  }
  CHECK(hipFree(buf));
  CHECK(hipFree(sink));
  return 0;
}
