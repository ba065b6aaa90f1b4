// The diagonal block of large_chol_solve_kernel (csrc/large_n.hip) and its addressing helpers — a header of their own so that
// tools/ubench/chol_diag.hip can time the block's chain alone (one wavefront, the block in LDS).
#pragma once
#include "wave_utils.hpp"
#include "ldlt_lds.hpp"

namespace toa {

// The diagonal block of large_chol_solve_kernel: ONE wavefront factors the 32 x 32 block whose lower triangle sits in LDS (row stride
// LS), lane r = row r in registers, in place; 1 / L_jj to rs_out.  LDS in, LDS out (round 5): the look-ahead stages the block there and
// the factored block reaches the matrix from another wave.  (As a __noinline__ function — TOA_CHOL_DIAG_CALL — the kernel needed 36
// registers more, not fewer; what took it to the 256-register limit was hipcc hoisting per-lane invariants of all phases in front of
// the block loop, see TOA_CHOL_FRESH_LANE in large_n.hip.)
//   32 unrolled Cholesky columns: column j before its scaling, lane by lane, comes out of the register by v_readlane (lane index =
//   compile-time constant; one LDS round trip per column until late round 4); one division per pivot (A_ij / d against the unscaled
//   column); the scalings 1 / sqrt(d_j) once at the end, lane j takes the root of ITS pivot (one sqrt and one division per lane)
#ifndef TOA_CHOL_DIAG_CALL
#define TOA_CHOL_DIAG_CALL __forceinline__
#endif
// a uniform base pointer + a 32-bit byte offset per lane: hipcc addresses it as global_load v, v_off, s[base] — one register per
// address instead of two, no 64-bit index arithmetic (the kernel holds 32 such addresses at a time where it moves a panel row block)
template <typename T>
__device__ __forceinline__ T ld_at(const T* base, const unsigned elem) { return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + elem * unsigned(sizeof(T))); }
template <typename T>
__device__ __forceinline__ void st_at(T* base, const unsigned elem, const T v) { *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + elem * unsigned(sizeof(T))) = v; }
// TOA_DIAG_VARIANT (tools/ubench/chol_diag.hip, shader cycles per block fp32 / fp64 on one wavefront alone):
//   0  round 4's order: pivot, IEEE 1 / d, the 31 - j updates of the column, each broadcast through s0        15 950 / 19 400
//   1  the next pivot's reciprocal started as soon as column j + 1 has its update                               15 120 / 19 280
//   2  1 + v_rcp and Newton steps instead of the IEEE quotient (not correctly rounded; different bits)           14 760 / 19 110
//   3  1 + eight broadcasts back to back into eight scalar registers, then their eight FMAs (shipped)           13 370 / 16 820
// 0, 1 and 3 give the same bits.  ~50 (fp32) / ~66 (fp64) instructions per pivot at ~8 cycles each: a lone wavefront issues a vector
// instruction every 5 cycles at best (profiles/r04_issue_probe.txt), and half of these wait on the one before.
#ifndef TOA_DIAG_VARIANT
#define TOA_DIAG_VARIANT 3
#endif
// 1 / d for a pivot d > 0: the IEEE quotient (variant 2: v_rcp + Newton steps)
template <typename T>
__device__ __forceinline__ T chol_recip(const T d) {
#if TOA_DIAG_VARIANT == 2
  if constexpr (sizeof(T) == 4) {
    float x = __builtin_amdgcn_rcpf(d);
    x = fmaf(fmaf(-d, x, 1.0f), x, x);
    return x;
  } else {
    double x = __builtin_amdgcn_rcp(d);
    x = fma(fma(-d, x, 1.0), x, x);
    x = fma(fma(-d, x, 1.0), x, x);
    return x;
  }
#else
  return T(1) / d;
#endif
}
template <typename T, int LS>
__device__ TOA_CHOL_DIAG_CALL void chol_diag_block(__attribute__((address_space(3))) T* Ld, __attribute__((address_space(3))) T* rs_out,
                                             __attribute__((address_space(3))) int* fail, const int bs) {
  constexpr int B = 32;
  const int lane = threadIdx.x & 63;
  T r[B];
#pragma unroll
  for (int c = 0; c < B; ++c) r[c] = (lane < bs && c <= lane && c < bs) ? Ld[lane * LS + c] : T(0);   // (columns past the diagonal are never read)
  bool bad = false;
#if TOA_DIAG_VARIANT == 0
#pragma unroll
  for (int j = 0; j < B; ++j) {
    const T d = wave_bcast(r[j], j);                   // the pivot: entry (j, j) after the updates of columns < j
    const bool live = j < bs;
    const bool pos = d > T(0) && d <= NumLimits<T>::max();
    if (live && !pos) bad = true;
    const T dd = (live && pos) ? d : T(1);
    const T t = r[j] * (T(1) / dd);                    // A_ij / d: with the unscaled A_cj this is L_ij L_cj
#pragma unroll
    for (int c = j + 1; c < B; ++c) r[c] = fma(-t, wave_bcast(r[j], c), r[c]);   // (lanes < c hold zeros there and are not stored)
    __builtin_amdgcn_sched_barrier(0);                 // one column's broadcasts at a time
  }
#else
  // round 5: the chain of a pivot — broadcast d, 1 / d, t = column / d, the first update (column j + 1), broadcast the next d — is ~16
  // DEPENDENT vector operations with the IEEE division (v_div_scale, v_rcp, four FMAs, v_div_fmas, v_div_fixup ...) at ~24 cycles
  // each: two thirds of the 506 cycles per pivot.  The next pivot's reciprocal now starts as soon as column j + 1 has its update
  // and runs UNDER the other 30 - j updates of column j (independent of it); the same operations on the same values.
  auto pivot_inv = [&](const T d, const int j) __attribute__((always_inline)) {
    const bool live = j < bs;
    const bool pos = d > T(0) && d <= NumLimits<T>::max();
    if (live && !pos) bad = true;
    const T dd = (live && pos) ? d : T(1);
    return chol_recip(dd);
  };
  T inv = pivot_inv(wave_bcast(r[0], 0), 0);
  static_for<B>([&](auto jc) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    const T t = r[j] * inv;                            // A_ij / d: with the unscaled A_cj this is L_ij L_cj
    if constexpr (j + 1 < B) {
      r[j + 1] = fma(-t, wave_bcast(r[j], j + 1), r[j + 1]);
      inv = pivot_inv(wave_bcast(r[j + 1], j + 1), j + 1);
    }
#if TOA_DIAG_VARIANT == 3
    // eight broadcasts back to back into eight scalar registers, then their eight FMAs: hipcc sent every v_readlane through s0 —
    // readlane, s_nop, fmac, ~29 cycles per update with the vector-to-scalar-to-vector latency exposed each time
    static_for<(B - j - 2 + 7) / 8>([&](auto qc) __attribute__((always_inline)) {
      constexpr int c0 = j + 2 + 8 * decltype(qc)::value;
      T b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) b[u] = c0 + u < B ? wave_bcast(r[j], c0 + u) : T(0);
      asm volatile("" : "+s"(b[0]), "+s"(b[1]), "+s"(b[2]), "+s"(b[3]), "+s"(b[4]), "+s"(b[5]), "+s"(b[6]), "+s"(b[7]));
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (c0 + u < B) r[c0 + u] = fma(-t, b[u], r[c0 + u]);
    });
#else
#pragma unroll
    for (int c = j + 2; c < B; ++c) r[c] = fma(-t, wave_bcast(r[j], c), r[c]);   // (lanes < c hold zeros there and are not stored)
#endif
    asm volatile("" : "+v"(inv));                      // (the next pivot's reciprocal belongs to THIS region: hipcc sank it behind the barrier, in front of the next column)
    __builtin_amdgcn_sched_barrier(0);                 // one column's broadcasts at a time
  });
#endif
  T dj = T(1);
#pragma unroll
  for (int c = 0; c < B; ++c) dj = c == lane ? r[c] : dj;
  const T rs = (lane < bs && dj > T(0) && dj <= NumLimits<T>::max()) ? T(1) / sqrt(dj) : T(1);
#pragma unroll
  for (int c = 0; c < B; ++c) r[c] *= wave_bcast(rs, c);   // lane j, column j: d / sqrt(d) = l; lanes > j: L_ij
  if (lane < bs) {
#pragma unroll
    for (int c = 0; c < B; ++c)
      if (c <= lane && c < bs) Ld[lane * LS + c] = r[c];
    rs_out[lane] = rs;                                 // 1 / L_jj: the panel rows multiply by these (32 divisions per row were 2.4 us of a panel's 11.6)
  }
  if (bad && lane == 0) *fail = 1;
}

}  // namespace toa
