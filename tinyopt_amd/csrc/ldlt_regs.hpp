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
// the right-looking update is one FMA per element, no LDS traffic, no synchronisation.
//
// Issue-slot economy of the update (tools/ubench/issue_probe.hip, ns per wave-instruction per SIMD):
// v_readlane 1.84, v_fmac 1.97 (SGPR operand), v_pk_fma_f32 with an SGPR-PAIR operand 2.22.  fp32 therefore
// keeps the row as aligned register pairs and updates two columns with one v_pk_fma_f32 whose multiplier is
// the SGPR pair written by two v_readlanes: 3 instructions per 2 columns instead of 4.  All broadcasts of a
// chunk are issued before its FMAs: a v_readlane feeding the very next VALU instruction costs an s_nop.
// (Broadcasting the pivot column through LDS instead — ds_read_b128, same address on all lanes — was measured:
// fewer VALU slots but either +56 live registers when hipcc hoists the reads, or an exposed LDS latency per
// chunk when they are pinned; slower than this version in the fused kernel both ways.)
//
// Layout in registers after factor():  row[j], j < lane : L[lane][j]
//                                       row[lane]        : d_lane (its reciprocal is kept in `dinv`)
//                                       row[j], j > lane : d_lane * L[j][lane]  (the un-scaled Schur row
//                                                          of step `lane`), which is exactly what the
//                                                          back substitution of  D L^T x = y  needs.
#pragma once
#include <utility>

#include "ldlt_lds.hpp"
#include "wave_utils.hpp"

namespace toa {

// Opaque copy of a wave-uniform value.  n is constant for a whole launch, so every `j < n` / `k < n` gate of the
// unrolled code below is loop-invariant with respect to the problem / iteration loops of the fused kernel; LICM
// hoists all ~200 of them and parks their SGPR masks in VGPR lanes for the entire kernel (883 SGPR spills = 14
// VGPRs at n = 50).  Recomputing a scalar compare per call is free next to that.
__device__ __forceinline__ int opaque_uniform(int v) {
  asm volatile("" : "+s"(v));
  return v;
}

// Row storage: element J through get<J>() / set<J>().  fp32 rows are (2p, 2p+1) register pairs.
template <typename T, int NPAD>
struct RowRegs {
  T r[NPAD];
  template <int J> __device__ __forceinline__ T get() const { return r[J]; }
  template <int J> __device__ __forceinline__ void set(T v) { r[J] = v; }
};
template <int NPAD>
struct RowRegs<float, NPAD> {
  using P = float __attribute__((ext_vector_type(2)));
  P r[NPAD / 2];
  template <int J> __device__ __forceinline__ float get() const { return r[J / 2][J & 1]; }
  template <int J> __device__ __forceinline__ void set(float v) { r[J / 2][J & 1] = v; }
};

template <typename T, int NPAD>
struct LdltRegs {
  static_assert(NPAD % 8 == 0, "columns are processed in chunks of 8");
  RowRegs<T, NPAD> row;
  T dinv;  // lane k: 1 / d_k

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
          row.template set<j>((in_n && j < n) ? r[j] : T(0));
        });
      } else {
        static_for<8>([&](auto jjc) __attribute__((always_inline)) { row.template set<jb + decltype(jjc)::value>(T(0)); });
      }
    });
    dinv = T(1);
  }

  // One 8-column chunk of the right-looking update  S[i][j] -= l_i * S[j][k],  j in [JB, JB + 8), j > K.
  template <int K, int JB>
  __device__ __forceinline__ void update_chunk(const T c, const T l) {
    if constexpr (sizeof(T) == 4) {
      using P = float __attribute__((ext_vector_type(2)));
      // pairs (2p, 2p+1) entirely right of the pivot: two broadcasts -> one SGPR pair -> one packed FMA
      P cp[4];
      float cs = 0;  // a leading odd column (j = K + 1 odd) is updated on its own
      static_for<4>([&](auto pc) __attribute__((always_inline)) {
        constexpr int j = JB + 2 * decltype(pc)::value;
        if constexpr (j > K) cp[decltype(pc)::value] = P{wave_bcast(c, j), wave_bcast(c, j + 1)};
        else if constexpr (j + 1 > K) cs = wave_bcast(c, j + 1);
      });
      __builtin_amdgcn_sched_barrier(0);
      const P nl2 = {-l, -l};
      static_for<4>([&](auto pc) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value;
        constexpr int j = JB + 2 * p;
        if constexpr (j > K) {
          asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(row.r[j / 2]) : "v"(nl2), "s"(cp[p]));
        } else if constexpr (j + 1 > K) {
          row.template set<j + 1>(fma(-l, cs, row.template get<j + 1>()));
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    } else {
      T cj[kChunk];
      static_for<kChunk>([&](auto jjc) __attribute__((always_inline)) {
        constexpr int j = JB + decltype(jjc)::value;
        if constexpr (j > K) cj[decltype(jjc)::value] = wave_bcast(c, j);  // S[j][k] = S[k][j]
      });
      __builtin_amdgcn_sched_barrier(0);
      static_for<kChunk>([&](auto jjc) __attribute__((always_inline)) {
        constexpr int j = JB + decltype(jjc)::value;
        if constexpr (j > K) row.template set<j>(fma(-l, cj[decltype(jjc)::value], row.template get<j>()));
      });
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // Columns per update chunk (a chunk that lies entirely beyond n is skipped by a uniform branch).  fp32 works on
  // register PAIRS, 8 columns at a time; fp64 broadcasts cost two v_readlane each, so padding columns are dearer and the
  // chunk is 4: n = 12 updates 12 columns per pivot instead of 16 (C3: a quarter of the factorisation's update work).
  static constexpr int kChunk = sizeof(T) == 8 ? 4 : 8;

  // Pivot gate of the fast path, on the scalar unit: the bit pattern of a positive, normal, not-huge value lies in
  // (lo, hi) as an unsigned integer (negative values and NaN have larger patterns), so one s_sub + s_cmp replaces two
  // vector compares and their s_nops.  The upper bound is below max() so that 1/d is still a NORMAL number for
  // v_rcp; anything outside falls back to the exact LDS routine, which carries the reference's acceptance rule.
  static __device__ __forceinline__ bool pivot_in_range(const T d) {
    if constexpr (sizeof(T) == 4) {
      const unsigned b = __builtin_amdgcn_readfirstlane(__float_as_uint(d));
      return (b - 0x00800001u) < (0x7E000000u - 0x00800001u);
    } else {
      const unsigned b = __builtin_amdgcn_readfirstlane(unsigned(__double_as_longlong(d) >> 32));
      return (b - 0x00100001u) < (0x7FC00000u - 0x00100001u);
    }
  }
  // 1/d for d accepted by pivot_in_range: hardware reciprocal + Newton steps (3 / 5 instructions, error < 1 ulp)
  // instead of the 12- / 22-instruction IEEE division sequence on the critical path of every pivot.
  static __device__ __forceinline__ T recip(const T d) {
    if constexpr (sizeof(T) == 4) {
      float r = __builtin_amdgcn_rcpf(d);
      r = fmaf(r, fmaf(-d, r, 1.0f), r);
      return r;
    } else {
      double r = __builtin_amdgcn_rcp(d);
      r = fma(r, fma(-d, r, 1.0), r);
      r = fma(r, fma(-d, r, 1.0), r);
      return r;
    }
  }

  // Returns true iff every pivot passed pivot_in_range (then the factorisation is complete).
  __device__ __forceinline__ bool factor(const int n_in, const int lane_in) {
    const int n = opaque_uniform(n_in);
    // opaque for the same reason as n: `lane > k` is one v_cmp here, but loop-invariant for the kernel's problem and
    // iteration loops, and LICM would park all 2 x NPAD masks in VGPR lanes (v_writelane / 2 v_readlane per use)
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    bool ok = true;
    static_for<NPAD>([&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      if (k < n && ok) {  // wave-uniform
        const T c = row.template get<k>();  // lane i > k: S[i][k]
        const T d = wave_bcast(c, k);
        if (!pivot_in_range(d)) {
          ok = false;
        } else {
          const T inv = recip(d);
          const bool below = lane > k;
          const T l = below ? c * inv : T(0);
          row.template set<k>(below ? l : c);  // lane k keeps d_k, lanes < k keep their Schur row entry
          dinv = (lane == k) ? inv : dinv;
          constexpr int jb0 = ((k + 1) / kChunk);
          static_for<NPAD / kChunk - jb0>([&](auto jbc) __attribute__((always_inline)) {
            constexpr int jb = (jb0 + decltype(jbc)::value) * kChunk;
            if (jb < n) update_chunk<k, jb>(c, l);  // wave-uniform: skip column chunks that are entirely padding
          });
        }
      }
    });
    return ok;
  }

  // x = A^-1 b using the factors above.  b_lane / return: element `lane` (lanes >= n: 0).
  __device__ __forceinline__ T solve(const int n_in, const int lane_in, const T b_lane) const {
    const int n = opaque_uniform(n_in);
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    T y = lane < n ? b_lane : T(0);
    static_for<NPAD - 1>([&](auto kc) __attribute__((always_inline)) {  // L y' = b   (unit lower, column sweep)
      constexpr int k = decltype(kc)::value;
      if (k + 1 < n) {
        const T s = wave_bcast(y, k);
        y = (lane > k) ? fma(-row.template get<k>(), s, y) : y;
      }
    });
    const T invd = dinv;
    // D L^T x = y' with the rows stored un-scaled: x_i = (y_i - sum_{j>i} row_i[j] x_j) / d_i
    static_for<NPAD - 1>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = NPAD - 1 - decltype(jc)::value;
      if (j < n) {
        const T xj = wave_bcast(y * invd, j);
        y = (lane < j) ? fma(-row.template get<j>(), xj, y) : y;
      }
    });
    return lane < n ? y * invd : T(0);
  }
};

}  // namespace toa
