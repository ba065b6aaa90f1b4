// Wavefront-level primitives for gfx950 (CDNA4): 64-lane waves, DPP row (16-lane) rotations,
// readlane broadcasts.  Everything here is wave-synchronous: a workgroup's waves never
// synchronise with each other in this library (each wave owns whole problems), so there is no
// __syncthreads() anywhere — only `wave_sync()` to order one wave's own LDS traffic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace toa {

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Orders this wave's LDS writes before its later LDS reads (cross-lane hand-off inside ONE wave).
// DS operations of a wave execute in issue order; this only has to stop the compiler from
// reordering and make it wait for outstanding LDS ops (wavefront-scope fence).
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- DPP row rotate (within each 16-lane row): returns src from lane ((l + N) mod 16) of the row.
template <int N>
__device__ __forceinline__ int dpp_row_ror_i32(int v) {
  // dpp_ctrl 0x120 + N = row_ror:N ; row_mask = bank_mask = 0xf ; bound_ctrl = false
  return __builtin_amdgcn_update_dpp(v, v, 0x120 + N, 0xf, 0xf, false);
}
template <int N>
__device__ __forceinline__ float dpp_row_ror(float v) {
  // old = 0 + bound_ctrl lets GCNDPPCombine fold the DPP move into its consumer
  // (v_add_f32_dpp / v_max_f32_dpp: one instruction per reduction stage instead of three)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, true));
}
template <int N>
__device__ __forceinline__ double dpp_row_ror(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = dpp_row_ror_i32<N>(int(b & 0xffffffffll));
  const int hi = dpp_row_ror_i32<N>(int(b >> 32));
  return __longlong_as_double((long long)(unsigned)lo | ((long long)hi << 32));
}

// All-reduce (sum) across each 16-lane DPP row; every lane of the row gets the row total.
template <typename T>
__device__ __forceinline__ T row16_allreduce_sum(T v) {
  v += dpp_row_ror<8>(v);
  v += dpp_row_ror<4>(v);
  v += dpp_row_ror<2>(v);
  v += dpp_row_ror<1>(v);
  return v;
}
template <typename T>
__device__ __forceinline__ T row16_allreduce_max(T v) {
  v = fmax(v, dpp_row_ror<8>(v));
  v = fmax(v, dpp_row_ror<4>(v));
  v = fmax(v, dpp_row_ror<2>(v));
  v = fmax(v, dpp_row_ror<1>(v));
  return v;
}

// Broadcast lane `src` (wave-uniform index) to all lanes.
__device__ __forceinline__ float wave_bcast(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
__device__ __forceinline__ int wave_bcast(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ double wave_bcast(double v, int src) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane(int(b & 0xffffffffll), src);
  const int hi = __builtin_amdgcn_readlane(int(b >> 32), src);
  return __longlong_as_double((long long)(unsigned)lo | ((long long)hi << 32));
}

// Wave-wide all-reduce: rows by DPP, then the 4 row totals via readlane (SGPR broadcast).
template <typename T>
__device__ __forceinline__ T wave_allreduce_sum(T v) {
  v = row16_allreduce_sum(v);
  return (wave_bcast(v, 0) + wave_bcast(v, 16)) + (wave_bcast(v, 32) + wave_bcast(v, 48));
}
template <typename T>
__device__ __forceinline__ T wave_allreduce_max(T v) {
  v = row16_allreduce_max(v);
  return fmax(fmax(wave_bcast(v, 0), wave_bcast(v, 16)), fmax(wave_bcast(v, 32), wave_bcast(v, 48)));
}

// Wave-uniform value helpers
__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

}  // namespace toa
