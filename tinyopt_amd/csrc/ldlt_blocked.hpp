// K3 fast path for 17 <= n <= 63 — blocked, right-looking, unpivoted LDL^T whose trailing updates run on the matrix cores.
//
// Same contract as LdltRegs (ldlt_regs.hpp): the positive-definite fast path of tinyopt::SolveLDLT
// (include/tinyopt/math.h:232-240); it reports failure as soon as a pivot is not safely positive, and the caller then
// re-creates the matrix and runs the pivoted LDS routine (ldlt_lds.hpp), which carries Eigen's exact acceptance rule.
//
// Why blocked.  The register version is one rank-1 update per pivot: n^2/2 broadcast + FMA pairs on the VALU (~3300
// instructions at n = 50), an eighth of a C4 launch.  Here the matrix stays in its LDS image M (n x LD) and is processed
// in 16-column panels:
//   panel      lane i loads its 16 panel entries, the 16 pivots are eliminated INSIDE the panel only (the register
//              scheme of LdltRegs restricted to 16 columns: 120 broadcast + FMA pairs instead of ~600), L written back;
//   trailing   T_{I,I'} -= L_I D L_I'^T for the 16 x 16 tiles right of / below the panel: four v_mfma_*_16x16x4 per tile.
//              Operands come straight out of M: lane l = 16 k' + c supplies L[16 I + c][16 J + 4 s + k'] — rows across the
//              lanes (stride LD, odd: conflict-free), the A-layout of L equals the B-layout of L^T, D folded into B.
//              The tile is loaded from / stored to M in the matrix cores' C/D layout, and mirrored so that the next panel
//              finds its columns.
//   solve      unit-lower forward sweep, D^-1, backward sweep with L read back from M (one column per step).
// n = 50: ~1900 instructions + 40 MFMAs against ~3300.  n <= 16 keeps LdltRegs (a single panel has no trailing update).
#pragma once
#include "dense_row.hpp"
#include "ldlt_regs.hpp"
#include "wave_utils.hpp"

namespace toa {

// t += sum_s a[s] (x) b[s]: the four K-slabs of one 16 x 16 x 16 tile update, inline asm like the Gram steps (dense_row.hpp).
// hipcc does not look inside asm statements, so the wait states are written out: VALU-written operands -> first MFMA, and
// the XDL write -> VALU / LDS read of the accumulator after the last one (19 wait states cover the 16-pass fp64 op; the
// 8-pass fp32 op needs fewer).  The accumulator is a VGPR tuple: it is loaded from and stored to LDS around the update.
__device__ __forceinline__ void mfma_tile_k16(Mfma<float>::Acc& t, const float (&a)[4], const float (&b)[4]) {
  asm volatile(
      "s_nop 1\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %1, %5, %0\n\ts_nop 1\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %2, %6, %0\n\ts_nop 1\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %3, %7, %0\n\ts_nop 1\n\t"
      "v_mfma_f32_16x16x4_f32 %0, %4, %8, %0\n\t"
      "s_nop 15\n\ts_nop 3"
      : "+v"(t)
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
}
__device__ __forceinline__ void mfma_tile_k16(Mfma<double>::Acc& t, const double (&a)[4], const double (&b)[4]) {
  asm volatile(
      "s_nop 1\n\t"
      "v_mfma_f64_16x16x4_f64 %0, %1, %5, %0\n\ts_nop 1\n\t"
      "v_mfma_f64_16x16x4_f64 %0, %2, %6, %0\n\ts_nop 1\n\t"
      "v_mfma_f64_16x16x4_f64 %0, %3, %7, %0\n\ts_nop 1\n\t"
      "v_mfma_f64_16x16x4_f64 %0, %4, %8, %0\n\t"
      "s_nop 15\n\ts_nop 3"
      : "+v"(t)
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
}

template <typename T, int NBLK>
struct LdltBlocked {
  static_assert(NBLK >= 2 && NBLK <= 4, "17 <= n <= 64");
  using Acc = typename Mfma<T>::Acc;
  T dinv;  // lane k: 1 / d_k

  // Factor the LDS image M (n x LD, full symmetric) IN PLACE: strict lower = L, diagonal = D.  Returns true iff every pivot
  // passed the fast path's range test (then the factorisation is complete); on false M is clobbered.
  __device__ __forceinline__ bool factor(T* __restrict__ M, const int LD, const int n_in, const int lane_in) {
    const int n = opaque_uniform(n_in);
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int g = lane >> 4, c = lane & 15;
    bool ok = true;
    dinv = T(1);
    static_for<NBLK>([&](auto jc) __attribute__((always_inline)) {
      constexpr int J = decltype(jc)::value;
      if (16 * J < n && ok) {  // wave-uniform
        // ---- panel: rows >= 16 J, columns [16 J, 16 J + 16)
        const bool in_rows = lane >= 16 * J && lane < n;
        const T* rowp = M + (in_rows ? lane : 0) * LD + 16 * J;
        T P[16];
        static_for<16>([&](auto qc) __attribute__((always_inline)) {
          constexpr int q = decltype(qc)::value;
          P[q] = (in_rows && 16 * J + q < n) ? rowp[q] : T(0);
        });
        static_for<16>([&](auto kc) __attribute__((always_inline)) {
          constexpr int k = decltype(kc)::value;
          constexpr int kg = 16 * J + k;
          if (kg < n && ok) {  // wave-uniform
            const T cv = P[k];  // lane i > kg: S[i][kg]
            const T d = wave_bcast(cv, kg);
            if (!LdltRegs<T, 16>::pivot_in_range(d)) {
              ok = false;
            } else {
              const T inv = LdltRegs<T, 16>::recip(d);
              const bool below = lane > kg;
              const T l = below ? cv * inv : T(0);
              P[k] = below ? l : cv;  // lane kg keeps d, the rows above keep their (unused) Schur entries
              dinv = (lane == kg) ? inv : dinv;
              T cj[15];
              static_for<15 - k>([&](auto jj) __attribute__((always_inline)) {
                constexpr int j = k + 1 + decltype(jj)::value;
                cj[j - 1] = wave_bcast(cv, 16 * J + j);  // S[16 J + j][kg]
              });
              __builtin_amdgcn_sched_barrier(0);
              static_for<15 - k>([&](auto jj) __attribute__((always_inline)) {
                constexpr int j = k + 1 + decltype(jj)::value;
                P[j] = fma(-l, cj[j - 1], P[j]);
              });
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        });
        if (ok) {
          T* wrow = M + (in_rows ? lane : 0) * LD + 16 * J;
          static_for<16>([&](auto qc) __attribute__((always_inline)) {
            constexpr int q = decltype(qc)::value;
            if (in_rows && 16 * J + q < n) wrow[q] = P[q];
          });
          wave_sync();
          // ---- trailing update on the matrix cores (only a FULL panel has anything to its right)
          if constexpr (J + 1 < NBLK) {
            if (16 * (J + 1) < n) {
              T dk[4];  // d of the panel pivot this lane's K index stands for, per slab
#pragma unroll
              for (int s = 0; s < 4; ++s) { const int kk = 16 * J + 4 * s + g; dk[s] = M[kk * LD + kk]; }
              T a[NBLK][4];  // a[I][s] = L[16 I + c][16 J + 4 s + g]
              static_for<NBLK - J - 1>([&](auto ic) __attribute__((always_inline)) {
                constexpr int I = J + 1 + decltype(ic)::value;
                const int r = 16 * I + c;
                const bool valid = r < n;
                const T* src = M + (valid ? r : 0) * LD + 16 * J + g;
#pragma unroll
                for (int s = 0; s < 4; ++s) a[I][s] = valid ? src[4 * s] : T(0);
              });
              static_for<NBLK - J - 1>([&](auto ic) __attribute__((always_inline)) {
                constexpr int I = J + 1 + decltype(ic)::value;
                if (16 * I < n) {
                  static_for<NBLK - I>([&](auto i2c) __attribute__((always_inline)) {
                    constexpr int I2 = I + decltype(i2c)::value;
                    if (16 * I2 < n) {
                      Acc t;
                      const int cc = 16 * I2 + c;
#pragma unroll
                      for (int r = 0; r < 4; ++r) {
                        const int rr = 16 * I + Mfma<T>::out_row(lane, r);
                        t[r] = (rr < n && cc < n) ? M[rr * LD + cc] : T(0);
                      }
                      T na[4], bd[4];
#pragma unroll
                      for (int s = 0; s < 4; ++s) { na[s] = -a[I][s]; bd[s] = a[I2][s] * dk[s]; }
                      mfma_tile_k16(t, na, bd);
#pragma unroll
                      for (int r = 0; r < 4; ++r) {
                        const int rr = 16 * I + Mfma<T>::out_row(lane, r);
                        if (rr < n && cc < n) {
                          M[rr * LD + cc] = t[r];
                          if constexpr (I2 != I) M[cc * LD + rr] = t[r];  // mirror: the next panels read columns
                        }
                      }
                    }
                  });
                }
              });
              wave_sync();
            }
          }
        }
      }
    });
    return ok;
  }

  // x = A^-1 b with the factors left in M by factor().  b_lane / return: element `lane` (lanes >= n: 0).
  __device__ __forceinline__ T solve(const T* __restrict__ M, const int LD, const int n_in, const int lane_in, const T b_lane) const {
    const int n = opaque_uniform(n_in);
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const bool in_n = lane < n;
    T y = in_n ? b_lane : T(0);
    const T* row = M + (in_n ? lane : 0) * LD;
    for (int k0 = 0; k0 < n - 1; k0 += 8) {  // L y' = b  (unit lower, column sweep; 8 columns of L in flight)
      T lk[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) lk[u] = (in_n && k0 + u < n - 1 && lane > k0 + u) ? row[k0 + u] : T(0);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (k0 + u < n - 1) {
          const T s = wave_bcast(y, k0 + u);
          y = fma(-lk[u], s, y);
        }
      }
    }
    y *= dinv;  // D^-1
    for (int j0 = n - 1; j0 > 0; j0 -= 8) {  // L^T x = y''  (column sweep of the transpose: row j of L)
      T lj[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int j = j0 - u; lj[u] = (j > 0 && lane < j) ? M[j * LD + lane] : T(0); }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = j0 - u;
        if (j > 0) {
          const T xj = wave_bcast(y, j);
          y = fma(-lj[u], xj, y);
        }
      }
    }
    return in_n ? y : T(0);
  }
};

// The fast path by size: one register panel up to 16 unknowns, the blocked matrix-core form beyond.  kClobbersM: a failed
// factor() has overwritten the LDS image, which the caller must re-create before the pivoted routine runs.
#ifdef TOA_LDLT_REGS   // A/B switch (tools/variant_build.sh): the register form for every n
template <typename T, int NPAD, bool kBlocked = false>
#else
template <typename T, int NPAD, bool kBlocked = (NPAD > 16)>
#endif
struct LdltFast;
template <typename T, int NPAD>
struct LdltFast<T, NPAD, false> {
  static constexpr bool kClobbersM = false;
  LdltRegs<T, NPAD> F;
  __device__ __forceinline__ bool factor(T* M, int LD, int n, int lane) { F.load(M, LD, n, lane); return F.factor(n, lane); }
  __device__ __forceinline__ T solve(const T*, int, int n, int lane, T b) const { return F.solve(n, lane, b); }
};
template <typename T, int NPAD>
struct LdltFast<T, NPAD, true> {
  static constexpr bool kClobbersM = true;
  LdltBlocked<T, (NPAD + 15) / 16> F;
  __device__ __forceinline__ bool factor(T* M, int LD, int n, int lane) { return F.factor(M, LD, n, lane); }
  __device__ __forceinline__ T solve(const T* M, int LD, int n, int lane, T b) const { return F.solve(M, LD, n, lane, b); }
};

}  // namespace toa
