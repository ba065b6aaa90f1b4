// K1 / K2 for the DenseRow residual family — Jacobian evaluation + J^T J / J^T r / ||r||^2
// accumulation as ONE Gram matrix on the matrix cores.
//
// Replaces, per problem and per LM iteration, the AD closure of the reference
// (include/tinyopt/diff/optimize_autodiff.h:91-166): residuals on Jets (jet.h:304-430), then
// grad = J^T r (:151), H = J^T J (:156), cost = ||r||^2 (:164).
//
// Formulation.  With W = [ J | r ] (m × (n+1)),   G = W^T W = [ J^T J   J^T r ]
//                                                             [ r^T J   r^T r ]
// so H, g and the cost all fall out of one symmetric rank-m update.  It is evaluated with
// v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64: one MFMA consumes 4 residual rows; lane
// l = 16*k + c supplies W[row 4s+k][column block element c] as BOTH the A and the B operand (the
// A-operand layout of W^T equals the B-operand layout of W), so no transpose and no LDS staging is
// needed: the 16-byte global load lands directly in MFMA operand order.
//
// Column blocking.  n+1 <= 16*NB columns are split into NB blocks; block cb holds columns
// {NB*c + cb : c = 0..15}, i.e. lane c owns NB CONTIGUOUS columns — one NB*sizeof(T)-byte load per
// lane per 4 rows (dwordx4 for fp32 n=50).  Only the NB(NB+1)/2 upper block pairs are computed.
//
// Row evaluation.  t = a_i.x is a 16-lane DPP row reduction (rows of the MFMA operand are exactly
// DPP rows); s = 1 + 0.1 cos t scales the lane's a-values into J in registers; the lane holding
// b_i replaces it by r_i = t + 0.1 sin t - b_i.
#pragma once
#include "wave_utils.hpp"

namespace toa {

template <typename T>
struct Mfma;
template <>
struct Mfma<float> {
  using Acc = float __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ Acc fma(float a, float b, Acc c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f32_16x16x4_f32: col = lane&15, row = (lane>>4)*4 + reg
  static __device__ __forceinline__ int out_row(int lane, int reg) { return (lane >> 4) * 4 + reg; }
};
template <>
struct Mfma<double> {
  using Acc = double __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ Acc fma(double a, double b, Acc c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane&15, row = (lane>>4) + 4*reg
  static __device__ __forceinline__ int out_row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};

__device__ __forceinline__ void sincos_t(float t, float* s, float* c) { sincosf(t, s, c); }
__device__ __forceinline__ void sincos_t(double t, double* s, double* c) { sincos(t, s, c); }

// Host+device layout helper.
struct DenseRowLayout {
  int nb;    // 16-column blocks
  int rs;    // row stride in elements = nb * ceil((n+1)/nb)
  int m4;    // rows padded to a multiple of 4
  __host__ __device__ static DenseRowLayout make(int n, int m) {
    DenseRowLayout L;
    L.nb = (n + 1 + 15) / 16;
    L.rs = L.nb * ((n + 1 + L.nb - 1) / L.nb);
    L.m4 = (m + 3) & ~3;
    return L;
  }
  __host__ __device__ size_t elems_per_problem() const { return size_t(m4) * rs; }
};

template <typename T, int NB>
struct DenseRowGram {
  static constexpr int NT = NB * (NB + 1) / 2;  // upper block pairs
  using Acc = typename Mfma<T>::Acc;
  Acc acc[NT];

  static __device__ __forceinline__ constexpr int tile(int i, int j) {  // i <= j
    return i * NB - i * (i - 1) / 2 + (j - i);
  }

  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = Acc{0, 0, 0, 0};
  }

  // One pass over the problem's rows.  WANT_H: full Gram (K1).  !WANT_H: cost only (K2) — returns
  // the wave-reduced sum of squares.  xs: x in LDS.  prob: packed [m4][RS].
  template <bool WANT_H>
  __device__ __forceinline__ T pass(const T* __restrict__ prob, const int m4, const int RS, const int n,
                                    const T* __restrict__ xs, const int lane) {
    const int k = lane >> 4, c = lane & 15;
    const bool active = c * NB < RS;
    const int cB = n / NB, cbB = n % NB;  // where b_i / r_i lives
    T xr[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      const int q = NB * c + cb;
      xr[cb] = (q < n) ? xs[q] : T(0);
    }
    const bool isB_lane = (c == cB);
    if (WANT_H) clear();
    T csum = 0;
    constexpr int kAlign = (NB * sizeof(T)) % 16 == 0 ? 16 : ((NB * sizeof(T)) % 8 == 0 ? 8 : 4);
    const T* rowp = prob + size_t(k) * RS + (active ? c * NB : 0);
    const size_t step_stride = size_t(4) * RS;
    const int steps = m4 >> 2;
    constexpr int U = 4;  // register double-buffer depth (steps in flight)
    T buf[U][NB];
    auto load = [&](int s, T(&w)[NB]) {
      if (active && s < steps) {
        const T* p = (const T*)__builtin_assume_aligned(rowp + size_t(s) * step_stride, kAlign);
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) w[cb] = p[cb];
      } else {
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) w[cb] = T(0);
      }
    };
#pragma unroll
    for (int u = 0; u < U; ++u) load(u, buf[u]);
    for (int s0 = 0; s0 < steps; s0 += U) {
      T cur[U][NB];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) cur[u][cb] = buf[u][cb];
#pragma unroll
      for (int u = 0; u < U; ++u) load(s0 + U + u, buf[u]);  // prefetch the next U steps
#pragma unroll
      for (int u = 0; u < U; ++u) {
        T(&w)[NB] = cur[u];
        T part = 0;
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) part += w[cb] * xr[cb];
        const T t = row16_allreduce_sum(part);
        T sn, cs;
        sincos_t(t, &sn, &cs);
        const T sc = T(1) + T(0.1) * cs;
        const T rbase = t + T(0.1) * sn;
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
          const bool isb = isB_lane && (cb == cbB);
          w[cb] = isb ? (rbase - w[cb]) : w[cb] * sc;
        }
        if (WANT_H) {
#pragma unroll
          for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = i; j < NB; ++j) acc[tile(i, j)] = Mfma<T>::fma(w[i], w[j], acc[tile(i, j)]);
        } else {
          T r = 0;
#pragma unroll
          for (int cb = 0; cb < NB; ++cb) r = (cb == cbB) ? w[cb] : r;
          csum += isB_lane ? r * r : T(0);
        }
      }
    }
    if (WANT_H) return T(0);
    return wave_allreduce_sum(csum);
  }

  // Scatter the Gram tiles: g[q] (q<n), undamped diagonal hd[q], cost = G[n][n].
  // Returns the cost (wave-uniform).  g / hd are LDS (or global) arrays of n.
  __device__ __forceinline__ T extract_g_diag_cost(T* __restrict__ g, T* __restrict__ hd, const int n,
                                                   const int lane, T* __restrict__ cost_slot) const {
    const int cj = lane & 15;
#pragma unroll
    for (int bi = 0; bi < NB; ++bi)
#pragma unroll
      for (int bj = bi; bj < NB; ++bj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qi = NB * Mfma<T>::out_row(lane, r) + bi;
          const int qj = NB * cj + bj;
          const T v = acc[tile(bi, bj)][r];
          if (qi < n && qj == n) g[qi] = v;
          if (qj < n && qi == n) g[qj] = v;
          if (qi == qj && qi < n) hd[qi] = v;
          if (qi == n && qj == n) *cost_slot = v;
        }
    wave_sync();
    return *cost_slot;
  }

  // Write the full symmetric n×n matrix (off-diagonals undamped, diagonal from `diag`) with row
  // stride LD.  Used to build the LDLT workspace and to export H.
  template <typename O>
  __device__ __forceinline__ void write_sym(O* __restrict__ M, const int LD, const int n, const int lane) const {
    const int cj = lane & 15;
#pragma unroll
    for (int bi = 0; bi < NB; ++bi)
#pragma unroll
      for (int bj = bi; bj < NB; ++bj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qi = NB * Mfma<T>::out_row(lane, r) + bi;
          const int qj = NB * cj + bj;
          if (qi < n && qj < n) {
            const O v = O(acc[tile(bi, bj)][r]);
            M[qi * LD + qj] = v;
            if (bi != bj) M[qj * LD + qi] = v;
          }
        }
  }
};

}  // namespace toa
