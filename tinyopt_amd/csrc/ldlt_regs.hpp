// K3 fast path — register-resident, unpivoted LDL^T for the positive-definite case.
//
// Same contract as ldlt_lds.hpp (tinyopt::SolveLDLT, include/tinyopt/math.h:232-240) whenever it
// succeeds; when it meets a pivot that is not safely positive it reports failure WITHOUT touching the
// LDS image, and the caller re-runs the pivoted LDS routine, which carries Eigen's exact acceptance
// rule (semi-definite passes, indefinite fails).  For a positive-definite matrix every symmetric
// pivot order yields all-positive pivots, so skipping Eigen's diagonal pivoting changes the result by
// rounding only.  The damped normal matrix of an LM step is positive definite except in degenerate
// problems, so in practice this is the path that runs.
//
// Why registers: on CDNA4 the LDS version is issue-bound (5 wave_syncs and a serial dependent-FMA
// chain per pivot: ~80k SIMD cycles at n = 50, a quarter of the fused kernel).  Here lane i holds row
// i of the matrix in NPAD registers, every index is a compile-time constant (the k and j loops are
// fully unrolled; `n` only gates uniform branches), the pivot column is broadcast with v_readlane and
// the right-looking update is one v_fmac per element: ~2 instructions per updated column, no LDS
// traffic, no synchronisation (~4 µs at n = 50).
//
// Layout in registers after factor():  row[j], j < lane : L[lane][j]
//                                       row[lane]        : d_lane (also kept in `dvec`)
//                                       row[j], j > lane : d_lane * L[j][lane]  (the un-scaled Schur row
//                                                          of step `lane`), which is exactly what the
//                                                          back substitution of  D L^T x = y  needs.
#pragma once
#include <utility>

#include "ldlt_lds.hpp"
#include "wave_utils.hpp"

namespace toa {

// static_for (wave_utils.hpp): every index below must be a compile-time constant for row[] to stay in registers
// (a `#pragma unroll` that hipcc declines — it does at NPAD = 64 — turns row[] into scratch memory).

// Opaque copy of a wave-uniform value.  n is constant for a whole launch, so every `j < n` / `k < n` gate of the
// unrolled code below is loop-invariant with respect to the problem / iteration loops of the fused kernel; LICM
// hoists all ~200 of them and parks their SGPR masks in VGPR lanes for the entire kernel (883 SGPR spills = 14
// VGPRs at n = 50).  Recomputing a scalar compare per call is free next to that.
__device__ __forceinline__ int opaque_uniform(int v) {
  asm volatile("" : "+s"(v));
  return v;
}

template <typename T, int NPAD>
struct LdltRegs {
  T row[NPAD];
  T dvec;

  // Lane i loads row i of the symmetric n×n LDS image (LD-strided); everything beyond n is zero.
  __device__ __forceinline__ void load(const T* __restrict__ M, const int LD, const int n_in, const int lane) {
    const int n = opaque_uniform(n_in);
    const bool in_n = lane < n;
    const T* r = M + (in_n ? lane : 0) * LD;
    static_for<NPAD / 8>([&](auto jbc) __attribute__((always_inline)) {
      constexpr int jb = decltype(jbc)::value * 8;
      if (jb < n) {
        static_for<8>([&](auto jjc) __attribute__((always_inline)) {
          constexpr int j = jb + decltype(jjc)::value;
          row[j] = (in_n && j < n) ? r[j] : T(0);
        });
      } else {
        static_for<8>([&](auto jjc) __attribute__((always_inline)) { row[jb + decltype(jjc)::value] = T(0); });
      }
    });
    dvec = T(1);
  }

  // Returns true iff every pivot was finite and > min_normal (then the factorisation is complete).
  __device__ __forceinline__ bool factor(const int n_in, const int lane) {
    const int n = opaque_uniform(n_in);
    bool ok = true;
    static_for<NPAD>([&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      if (k < n && ok) {  // wave-uniform
        const T d = wave_bcast(row[k], k);
        if (!(d > NumLimits<T>::min_normal()) || !(d < NumLimits<T>::max())) {
          ok = false;
        } else {
          const T c = row[k];              // lane i > k: S[i][k]
          const T inv = T(1) / d;
          const bool below = lane > k;
          const T l = below ? c * inv : T(0);
          row[k] = below ? l : row[k];     // lane k keeps d_k, lanes < k keep their Schur row entry
          dvec = (lane == k) ? d : dvec;
          constexpr int jb0 = ((k + 1) / 8);
          static_for<NPAD / 8 - jb0>([&](auto jbc) __attribute__((always_inline)) {
            constexpr int jb = (jb0 + decltype(jbc)::value) * 8;
            if (jb < n) {  // wave-uniform: skip column chunks that are entirely padding
              static_for<8>([&](auto jjc) __attribute__((always_inline)) {
                constexpr int j = jb + decltype(jjc)::value;
                if constexpr (j > k) {
                  const T cj = wave_bcast(c, j);  // S[j][k] = S[k][j]
                  row[j] = fma(-l, cj, row[j]);
                }
              });
            }
          });
        }
      }
    });
    return ok;
  }

  // x = A^-1 b using the factors above.  b_lane / return: element `lane` (lanes >= n: 0).
  __device__ __forceinline__ T solve(const int n_in, const int lane, const T b_lane) const {
    const int n = opaque_uniform(n_in);
    T y = lane < n ? b_lane : T(0);
    static_for<NPAD - 1>([&](auto kc) __attribute__((always_inline)) {  // L y' = b   (unit lower, column sweep)
      constexpr int k = decltype(kc)::value;
      if (k + 1 < n) {
        const T s = wave_bcast(y, k);
        y = (lane > k) ? fma(-row[k], s, y) : y;
      }
    });
    const T invd = T(1) / dvec;
    // D L^T x = y' with the rows stored un-scaled: x_i = (y_i - sum_{j>i} row_i[j] x_j) / d_i
    static_for<NPAD - 1>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = NPAD - 1 - decltype(jc)::value;
      if (j < n) {
        const T xj = wave_bcast(y * invd, j);
        y = (lane < j) ? fma(-row[j], xj, y) : y;
      }
    });
    return lane < n ? y * invd : T(0);
  }
};

}  // namespace toa
