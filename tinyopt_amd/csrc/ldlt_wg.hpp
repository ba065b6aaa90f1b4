// K3 for 64 <= n <= 128 inside a workgroup-per-problem kernel (large_fused.hip): blocked, right-looking, unpivoted LDL^T of
// the n x n LDS image by the FOUR wavefronts of the workgroup, trailing updates on the matrix cores.
//
// Replaces the same reference code as ldlt_blocked.hpp (tinyopt::SolveLDLT, include/tinyopt/math.h:232-240, behind
// SolverGN::Solve gn.h:150-171) where one wavefront no longer holds a row per lane.  A pivot that is not safely positive
// fails the solve (the damped matrices of an LM run are definite; as on the library path the acceptance differs from
// Eigen's LDLT only for singular positive SEMI-definite matrices).
//
//   panel J    16 columns.  Every wave keeps the 16 rows of the diagonal block in lanes 0..15 (redundantly: the pivots
//              and the block's columns are then 16-lane broadcasts, no inter-wave traffic while the panel is eliminated)
//              and 48 of the rows below it in lanes 16..63: 4 x 48 >= 112 rows.  Register elimination of the 16 pivots
//              restricted to the panel (the scheme of LdltBlocked / LdltRegs), L and D written back.     [2 barriers]
//   trailing   T_{I,I'} -= L_I D L_I'^T for the 16 x 16 tiles behind the panel, dealt round-robin to the four waves:
//              four v_mfma_*_16x16x4 per tile, operands straight from the LDS image (mfma_tile_k16).        [barrier]
//   solve      wave 0: unit-lower forward sweep, D^-1, backward sweep, two unknowns per lane, eight columns of L in
//              flight per step.
// 3 NB barriers instead of the n of the column-by-column Cholesky (large_chol_solve_kernel): n = 128 factors in ~10 us
// instead of ~100 us.
#pragma once
#include "ldlt_blocked.hpp"

namespace toa {

template <typename T, int NB>
struct WgLdlt {
  using Acc = typename Mfma<T>::Acc;
  static __device__ __forceinline__ constexpr int tile_index(int J, int I, int I2) {  // position of (I, I2) behind panel J
    int idx = 0;
    for (int a = J + 1; a < NB; ++a)
      for (int b = a; b < NB; ++b) {
        if (a == I && b == I2) return idx;
        ++idx;
      }
    return idx;
  }

  // Factor M (n x LD, full symmetric, LD odd, followed by >= 16 readable elements) in place: strict lower = L, diagonal = D;
  // dinv[k] = 1 / d_k.  Every thread of
  // the workgroup calls it (it contains barriers) and gets the same answer: true iff every pivot was safely positive.
  static __device__ __forceinline__ bool factor(T* __restrict__ M, const int LD, const int n_in, T* __restrict__ dinv,
                                                const int tid_in) {
    const int n = opaque_uniform(n_in);
    int tid = tid_in;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    bool ok = true;
    static_for<NB>([&](auto jc) __attribute__((always_inline)) {
      constexpr int J = decltype(jc)::value;
      if (16 * J < n && ok) {  // workgroup-uniform
        // ---- panel: lanes 0..15 = the diagonal block's rows, lanes 16..63 = this wave's share of the rows below it
        const int r = lane < 16 ? 16 * J + lane : 16 * (J + 1) + 48 * wave + (lane - 16);
        const bool in_rows = r < n;
        T* rowp = M + (in_rows ? r : 0) * LD + 16 * J;
        T P[16];
        static_for<16>([&](auto qc) __attribute__((always_inline)) {
          constexpr int q = decltype(qc)::value;
          const T v = rowp[q];  // unconditional (the image carries 16 elements of slack), masked afterwards
          P[q] = (in_rows && 16 * J + q < n) ? v : T(0);
        });
        __syncthreads();  // every wave has read the diagonal block before wave 0 overwrites it with L / D
        // The 16 pivots of the panel without a branch per pivot: a pivot outside the safe range only raises `bad` (the
        // arithmetic carries on with inf / nan, harmlessly: the factorisation is abandoned at the end of the panel), and the
        // reciprocals stay in a register until the panel is done.
        bool bad = false;
        T dinv_lane = T(0);  // lane k (< 16): 1 / d of pivot 16 J + k
        static_for<16>([&](auto kc) __attribute__((always_inline)) {
          constexpr int k = decltype(kc)::value;
          constexpr int kg = 16 * J + k;
          if (kg < n) {  // uniform
            const T cv = P[k];
            const T d = wave_bcast(cv, k);  // lane k holds row kg
            bad = bad || !LdltRegs<T, 16>::pivot_in_range(d);
            const T inv = LdltRegs<T, 16>::recip(d);
            const bool below = lane > k;  // lanes >= 16 are rows beyond the diagonal block
            const T l = below ? cv * inv : T(0);
            P[k] = below ? l : cv;
            dinv_lane = (lane == k) ? inv : dinv_lane;
            T cj[15];
            static_for<15 - k>([&](auto jj) __attribute__((always_inline)) {
              constexpr int j = k + 1 + decltype(jj)::value;
              cj[j - 1] = wave_bcast(cv, j);  // S[16 J + j][kg]
            });
            __builtin_amdgcn_sched_barrier(0);
            static_for<15 - k>([&](auto jj) __attribute__((always_inline)) {
              constexpr int j = k + 1 + decltype(jj)::value;
              P[j] = fma(-l, cj[j - 1], P[j]);
            });
            __builtin_amdgcn_sched_barrier(0);
          }
        });
        ok = !bad;
        if (wave == 0 && lane < 16 && 16 * J + lane < n) dinv[16 * J + lane] = dinv_lane;
        if (ok && in_rows && (lane >= 16 || wave == 0)) {
          static_for<16>([&](auto qc) __attribute__((always_inline)) {
            constexpr int q = decltype(qc)::value;
            if (16 * J + q < n) rowp[q] = P[q];
          });
        }
        __syncthreads();
        // ---- trailing update on the matrix cores
        if constexpr (J + 1 < NB) {
          if (ok && 16 * (J + 1) < n) {
            T dk[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) { const int kk = 16 * J + 4 * s + g; dk[s] = M[kk * LD + kk]; }
            T a[NB][4];  // a[I][s] = L[16 I + c][16 J + 4 s + g]
            static_for<NB - J - 1>([&](auto ic) __attribute__((always_inline)) {
              constexpr int I = J + 1 + decltype(ic)::value;
              const int rr = 16 * I + c;
              const bool valid = rr < n;
              const T* src = M + (valid ? rr : 0) * LD + 16 * J + g;
#pragma unroll
              for (int s = 0; s < 4; ++s) { const T v = src[4 * s]; a[I][s] = valid ? v : T(0); }
            });
            static_for<NB - J - 1>([&](auto ic) __attribute__((always_inline)) {
              constexpr int I = J + 1 + decltype(ic)::value;
              static_for<NB - I>([&](auto i2c) __attribute__((always_inline)) {
                constexpr int I2 = I + decltype(i2c)::value;
                constexpr int owner = tile_index(J, I, I2) & 3;
                if (16 * I2 < n && wave == owner) {  // wave-uniform
                  Acc t;
                  const int cc = 16 * I2 + c;
#pragma unroll
                  for (int r2 = 0; r2 < 4; ++r2) {
                    const int rr = 16 * I + Mfma<T>::out_row(lane, r2);
                    const T v = M[(rr < n ? rr : 0) * LD + cc];
                    t[r2] = (rr < n && cc < n) ? v : T(0);
                  }
                  T na[4], bd[4];
#pragma unroll
                  for (int s = 0; s < 4; ++s) { na[s] = -a[I][s]; bd[s] = a[I2][s] * dk[s]; }
                  mfma_tile_k16(t, na, bd);
#pragma unroll
                  for (int r2 = 0; r2 < 4; ++r2) {
                    const int rr = 16 * I + Mfma<T>::out_row(lane, r2);
                    if (rr < n && cc < n) {
                      M[rr * LD + cc] = t[r2];
                      if constexpr (I2 != I) M[cc * LD + rr] = t[r2];  // mirror: the next panels read columns
                    }
                  }
                }
              });
            });
          }
          __syncthreads();
        }
      }
    });
    return ok;
  }

  // x = A^-1 b with the factors left in M by factor(); ONE wavefront (lane = tid < 64) calls it.  b / x: LDS vectors.
  // Branch-free: every LDS read is unconditional on a clamped address and masked by a select afterwards (a predicated
  // read is an exec-mask branch per element), and each sweep is split at unknown 64 so that the register a pivot is
  // broadcast from is known at compile time.
  static __device__ __forceinline__ void solve(const T* __restrict__ M, const int LD, const int n_in, const T* __restrict__ dinv,
                                               T* __restrict__ bx, const int lane_in) {
    const int n = opaque_uniform(n_in);
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const bool in0 = lane < n, in1 = lane + 64 < n;
    const int r0 = in0 ? lane : 0, r1 = in1 ? lane + 64 : 0;
    T y0 = bx[r0], y1 = bx[r1];
    y0 = in0 ? y0 : T(0);
    y1 = in1 ? y1 : T(0);
    const T* row0 = M + r0 * LD;
    const T* row1 = M + r1 * LD;
    const int nm1 = n - 1;
    // ---- L y' = b (unit lower, column sweep; 8 columns of L in flight)
    const int kmid = nm1 < 64 ? nm1 : 64;
    for (int k0 = 0; k0 < kmid; k0 += 8) {  // pivots 0..63: broadcast from y0; rows lane > k of y0, every row of y1
      T l0[8], l1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + u, kc = k < nm1 ? k : nm1;
        const T v0 = row0[kc], v1 = row1[kc];
        l0[u] = (in0 && k < kmid && lane > k) ? v0 : T(0);
        l1[u] = (in1 && k < kmid) ? v1 : T(0);
      }
      __builtin_amdgcn_sched_barrier(0);  // all eight columns in flight before the dependent chain starts
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const T s = wave_bcast(y0, (k0 + u) & 63);
        y0 = fma(-l0[u], s, y0);
        y1 = fma(-l1[u], s, y1);
      }
    }
    for (int k0 = 64; k0 < nm1; k0 += 8) {  // pivots 64..n-2: broadcast from y1; rows lane + 64 > k of y1 only
      T l1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + u, kc = k < nm1 ? k : nm1;
        const T v1 = row1[kc];
        l1[u] = (in1 && k < nm1 && lane + 64 > k) ? v1 : T(0);
      }
      __builtin_amdgcn_sched_barrier(0);  // all eight columns in flight before the dependent chain starts
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const T s = wave_bcast(y1, (k0 + u - 64) & 63);
        y1 = fma(-l1[u], s, y1);
      }
    }
    {  // D^-1
      const T d0 = dinv[r0], d1 = dinv[r1];
      y0 *= in0 ? d0 : T(0);
      y1 *= in1 ? d1 : T(0);
    }
    // ---- L^T x = y'' (row j of L, j descending)
    for (int j0 = nm1; j0 >= 64; j0 -= 8) {  // pivots n-1..64: broadcast from y1; every row of y0, rows lane + 64 < j of y1
      T l0[8], l1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = j0 - u, jc = j > 0 ? j : 0;
        const T v0 = M[jc * LD + r0], v1 = M[jc * LD + r1];
        l0[u] = (in0 && j >= 64) ? v0 : T(0);
        l1[u] = (in1 && j >= 64 && lane + 64 < j) ? v1 : T(0);
      }
      __builtin_amdgcn_sched_barrier(0);  // all eight columns in flight before the dependent chain starts
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const T xj = wave_bcast(y1, (j0 - u - 64) & 63);
        y0 = fma(-l0[u], xj, y0);
        y1 = fma(-l1[u], xj, y1);
      }
    }
    for (int j0 = nm1 < 63 ? nm1 : 63; j0 > 0; j0 -= 8) {  // pivots 63..1: broadcast from y0; rows lane < j of y0 only
      T l0[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = j0 - u, jc = j > 0 ? j : 0;
        const T v0 = M[jc * LD + r0];
        l0[u] = (in0 && j > 0 && lane < j) ? v0 : T(0);
      }
      __builtin_amdgcn_sched_barrier(0);  // all eight columns in flight before the dependent chain starts
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const T xj = wave_bcast(y0, (j0 - u) & 63);
        y0 = fma(-l0[u], xj, y0);
      }
    }
    if (in0) bx[lane] = y0;
    if (in1) bx[lane + 64] = y1;
  }
};

}  // namespace toa
