// Wavefront-level primitives for gfx950 (CDNA4): 64-lane waves, DPP row (16-lane) rotations,
// readlane broadcasts.  Everything here is wave-synchronous: a workgroup's waves never
// synchronise with each other in this library (each wave owns whole problems), so there is no
// __syncthreads() anywhere — only `wave_sync()` to order one wave's own LDS traffic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

namespace toa {

constexpr int kWave = 64;

// Compile-time loop: f(std::integral_constant<int, I>{}) for I = 0..N-1.
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

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
  // dpp_ctrl 0x120 + N = row_ror:N ; row_mask = bank_mask = 0xf ; every lane has a source, so `old` is never
  // used: old = 0 + bound_ctrl avoids the register copy that seeding `old` with v costs
  return __builtin_amdgcn_update_dpp(0, v, 0x120 + N, 0xf, 0xf, true);
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

// ---- DPP quad permute (within each group of 4 adjacent lanes): lane i of the quad reads lane SEL_i.
template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_quad(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, int(b & 0xffffffffll), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, int(b >> 32), CTRL, 0xf, 0xf, true);
  return __longlong_as_double((long long)(unsigned)lo | ((long long)hi << 32));
}
constexpr int kQuadSwap1 = 0xB1;  // quad_perm:[1,0,3,2]  (partner = lane ^ 1)
constexpr int kQuadSwap2 = 0x4E;  // quad_perm:[2,3,0,1]  (partner = lane ^ 2)
// Broadcast quad lane U to the whole quad.
template <int U, typename T>
__device__ __forceinline__ T quad_bcast(T v) { return dpp_quad<U * 0x55>(v); }

// Transposed row reduction of FOUR values at once: p[0..3] are per-lane partial sums of four independent
// 16-lane row reductions; on return the lane at quad position q = lane & 3 holds the complete row total of p[q]
// (all 16 lanes of a row that share q hold the same number).  11 DPP-adds / selects for f32 instead of the 16 that
// four separate all-reduces cost, and the result arrives already distributed "one value per quad lane", which is
// what lets the caller run the expensive per-row function (sin/cos) once per batch instead of four times.
template <typename T>
__device__ __forceinline__ T quad_transpose_reduce(const T (&p)[4], const bool bit0, const bool bit1) {
  const T keep01 = bit0 ? p[1] : p[0], send01 = bit0 ? p[0] : p[1];
  const T keep23 = bit0 ? p[3] : p[2], send23 = bit0 ? p[2] : p[3];
  const T r01 = keep01 + dpp_quad<kQuadSwap1>(send01);
  const T r23 = keep23 + dpp_quad<kQuadSwap1>(send23);
  const T keep = bit1 ? r23 : r01, send = bit1 ? r01 : r23;
  T s = keep + dpp_quad<kQuadSwap2>(send);
  s += dpp_row_ror<4>(s);
  s += dpp_row_ror<8>(s);
  return s;
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

// Value of lane (l ^ 4) of the same 16-lane row: rotation by 12 (source l - 4) everywhere, then rotation by 4 (source l + 4)
// written over the banks whose lanes have bit 2 clear (banks 0 and 2 = lanes 0-3, 8-11).
__device__ __forceinline__ int dpp_row_xor4_i32(int v) {
  const int a = __builtin_amdgcn_update_dpp(0, v, 0x120 + 12, 0xf, 0xf, true);
  return __builtin_amdgcn_update_dpp(a, v, 0x120 + 4, 0xf, 0x5, false);
}
__device__ __forceinline__ float dpp_row_xor4(float v) { return __int_as_float(dpp_row_xor4_i32(__float_as_int(v))); }
__device__ __forceinline__ double dpp_row_xor4(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = dpp_row_xor4_i32(int(b & 0xffffffffll)), hi = dpp_row_xor4_i32(int(b >> 32));
  return __longlong_as_double((long long)(unsigned)lo | ((long long)hi << 32));
}

// Transposed reduction of 32 values over the 64 lanes of a wave: p[i] are per-lane partial sums of 32 independent
// wave-wide sums; on return EVERY lane l holds the complete total of p[l & 31].  A butterfly over the lane bits (quad swaps,
// row rotations, two cross-row shuffles): ~230 instructions for fp64 where 32 separate all-reduces cost ~800, and a
// fixed order of additions (deterministic).
// (`get(std::integral_constant<int, i>)` = this lane's partial of sum i: every index is a compile-time constant by construction — the
//  run-time compiler kept a 32-element operand ARRAY of the round-3 form in scratch memory, 144 / 272 B per lane, in every JetModel build)
template <typename T, typename Get>
__device__ __forceinline__ T wave_transposed_reduce32_of(Get&& get, const int lane) {
  const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0, b2 = (lane & 4) != 0, b3 = (lane & 8) != 0, b4 = (lane & 16) != 0;
  T r1[16], r2[8], r3[4], r4[2];
  static_for<16>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    const T pa = get(std::integral_constant<int, 2 * i>{}), pb = get(std::integral_constant<int, 2 * i + 1>{});
    const T keep = b0 ? pb : pa, send = b0 ? pa : pb;
    r1[i] = keep + dpp_quad<kQuadSwap1>(send);
  });
  static_for<8>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    const T keep = b1 ? r1[2 * i + 1] : r1[2 * i], send = b1 ? r1[2 * i] : r1[2 * i + 1];
    r2[i] = keep + dpp_quad<kQuadSwap2>(send);
  });
  static_for<4>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    const T keep = b2 ? r2[2 * i + 1] : r2[2 * i], send = b2 ? r2[2 * i] : r2[2 * i + 1];
    r3[i] = keep + dpp_row_xor4(send);
  });
  static_for<2>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    const T keep = b3 ? r3[2 * i + 1] : r3[2 * i], send = b3 ? r3[2 * i] : r3[2 * i + 1];
    r4[i] = keep + dpp_row_ror<8>(send);
  });
  const T keep = b4 ? r4[1] : r4[0], send = b4 ? r4[0] : r4[1];
  const T r5 = keep + __shfl_xor(send, 16, 64);
  return r5 + __shfl_xor(r5, 32, 64);
}
template <typename T>
__device__ __forceinline__ T wave_transposed_reduce32(const T (&p)[32], const int lane) {
  return wave_transposed_reduce32_of<T>([&](auto ic) __attribute__((always_inline)) { return p[decltype(ic)::value]; }, lane);
}

// All-reduce N independent per-lane partial sums: on return every lane holds every total.  N >= 12: transposed reductions of
// 32 values at a time, then one broadcast per value (~290 instructions per 32 fp64 values instead of ~800); below that the
// plain all-reduces are cheaper.
template <int N, typename T>
__device__ __forceinline__ void wave_allreduce_many(T (&G)[N], const int lane) {
  if constexpr (N < 12) {
#pragma unroll
    for (int i = 0; i < N; ++i) G[i] = wave_allreduce_sum(G[i]);
  } else {
    static_for<(N + 31) / 32>([&](auto cc) __attribute__((always_inline)) {
      constexpr int c0 = decltype(cc)::value * 32;
      const T r = wave_transposed_reduce32_of<T>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        if constexpr (c0 + i < N) return G[c0 + i];
        else return T(0);
      }, lane);
      static_for<32>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        if constexpr (c0 + i < N) G[c0 + i] = wave_bcast(r, i);
      });
    });
  }
}

// Wave-uniform value helpers
__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

}  // namespace toa
