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

// sin/cos for the DenseRow functor.  Branch-free Cody-Waite reduction by pi/2 (FMA, 3 / 2 constants)
// + Cephes minimax polynomials on [-pi/4, pi/4]: <= ~1.5 ulp for |t| < ~1e4, which covers the
// model's domain (|a_i.x| <= n * |x|_inf).  libm's sincos carries a Payne-Hanek path whose register
// footprint would halve the occupancy of the whole fused kernel for arguments that never occur.
__device__ __forceinline__ void sincos_t(float t, float* s, float* c) {
  const float j = rintf(t * 0.636619772367581343f);
  float y = fmaf(-j, 1.5707963705062866f, t);
  y = fmaf(-j, -4.371138828673793e-08f, y);
  y = fmaf(-j, -1.7151245100058819e-15f, y);
  const float z = y * y;
  const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * y, y);
  const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z,
                        fmaf(-0.5f, z, 1.0f));
  const int q = int(j) & 3;
  const float sv = (q & 1) ? pc : ps;
  const float cv = (q & 1) ? ps : pc;
  *s = (q & 2) ? -sv : sv;
  *c = ((q + 1) & 2) ? -cv : cv;
}
__device__ __forceinline__ void sincos_t(double t, double* s, double* c) {
  const double j = rint(t * 0.63661977236758134308);
  double y = fma(-j, 1.5707963267948966, t);
  y = fma(-j, 6.123233995736766e-17, y);
  const double z = y * y;
  double ps = 1.58962301576546568060e-10;
  ps = fma(ps, z, -2.50507477628578072866e-8);
  ps = fma(ps, z, 2.75573136213857245213e-6);
  ps = fma(ps, z, -1.98412698295895385996e-4);
  ps = fma(ps, z, 8.33333333332211858878e-3);
  ps = fma(ps, z, -1.66666666666666307295e-1);
  ps = fma(ps, z * y, y);
  double pc = -1.13585365213876817300e-11;
  pc = fma(pc, z, 2.08757008419747316778e-9);
  pc = fma(pc, z, -2.75573141792967388112e-7);
  pc = fma(pc, z, 2.48015872888517045348e-5);
  pc = fma(pc, z, -1.38888888888730564116e-3);
  pc = fma(pc, z, 4.16666666666665929218e-2);
  pc = fma(pc, z * z, fma(-0.5, z, 1.0));
  const int q = int(j) & 3;
  const double sv = (q & 1) ? pc : ps;
  const double cv = (q & 1) ? ps : pc;
  *s = (q & 2) ? -sv : sv;
  *c = ((q + 1) & 2) ? -cv : cv;
}

// ---- bounds-checked buffer loads (SRSRC path), software-pipelined by hand ------------------------
// A buffer descriptor (V#) spanning exactly one problem lets every lane issue its load
// UNCONDITIONALLY: lanes past the row's last column and steps past the last row use an offset
// >= num_records and the hardware returns 0 — no exec-masked branches in the hot loop.
//
// The loads are inline asm on purpose (cdna_hip_programming.md §5.7 form (ii)): with compiler-visible
// loads hipcc either sinks the prefetch below the MFMAs or waits vmcnt(0) right after issuing it
// (seen in the .s of three different formulations), which serialises HBM latency with the matrix
// pipe.  Here the issue point is pinned at the top of the batch and the single wait sits after the
// batch's MFMAs, naming every destination register "+v" so no consumer can be scheduled above it.
using u32x2 = unsigned __attribute__((ext_vector_type(2)));
using u32x3 = unsigned __attribute__((ext_vector_type(3)));
using u32x4 = unsigned __attribute__((ext_vector_type(4)));
using i32x4 = int __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane(int(unsigned(a)));
  r[1] = __builtin_amdgcn_readfirstlane(int(unsigned(a >> 32) & 0xffffu));  // stride = 0
  r[2] = __builtin_amdgcn_readfirstlane(int(bytes));                         // num_records (bytes)
  r[3] = 0x00020000;  // DATA_FORMAT = 32 (raw dwords), no swizzle: gfx9 / CDNA encoding
  return r;
}

template <int kDwords>
struct RawVec;
template <>
struct RawVec<1> {
  unsigned a;
  __device__ __forceinline__ void issue(i32x4 r, unsigned voff, unsigned soff) {
    asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, %3 offen" : "=v"(a) : "v"(voff), "s"(r), "s"(soff) : "memory");
  }
  __device__ __forceinline__ void get(unsigned* o) const { o[0] = a; }
};
template <>
struct RawVec<2> {
  u32x2 a;
  __device__ __forceinline__ void issue(i32x4 r, unsigned voff, unsigned soff) {
    asm volatile("s_nop 4\n\tbuffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(a) : "v"(voff), "s"(r), "s"(soff) : "memory");
  }
  __device__ __forceinline__ void get(unsigned* o) const { o[0] = a[0]; o[1] = a[1]; }
};
template <>
struct RawVec<3> {
  u32x3 a;
  __device__ __forceinline__ void issue(i32x4 r, unsigned voff, unsigned soff) {
    asm volatile("s_nop 4\n\tbuffer_load_dwordx3 %0, %1, %2, %3 offen" : "=v"(a) : "v"(voff), "s"(r), "s"(soff) : "memory");
  }
  __device__ __forceinline__ void get(unsigned* o) const { o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; }
};
template <>
struct RawVec<4> {
  u32x4 a;
  __device__ __forceinline__ void issue(i32x4 r, unsigned voff, unsigned soff) {
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(a) : "v"(voff), "s"(r), "s"(soff) : "memory");
  }
  __device__ __forceinline__ void get(unsigned* o) const { o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; }
};
template <>
struct RawVec<6> {
  u32x4 a;
  u32x2 b;
  __device__ __forceinline__ void issue(i32x4 r, unsigned voff, unsigned soff) {
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %3, %4 offen\n\tbuffer_load_dwordx2 %1, %2, %3, %4 offen offset:16"
                 : "=&v"(a), "=&v"(b) : "v"(voff), "s"(r), "s"(soff) : "memory");
  }
  __device__ __forceinline__ void get(unsigned* o) const {
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1];
  }
};
template <>
struct RawVec<8> {
  u32x4 a, b;
  __device__ __forceinline__ void issue(i32x4 r, unsigned voff, unsigned soff) {
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %3, %4 offen\n\tbuffer_load_dwordx4 %1, %2, %3, %4 offen offset:16"
                 : "=&v"(a), "=&v"(b) : "v"(voff), "s"(r), "s"(soff) : "memory");
  }
  __device__ __forceinline__ void get(unsigned* o) const {
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
  }
};
// Wait for every outstanding VMEM load of this wave; the operands make the four batch slots
// data-dependent on the wait so that no consumer is hoisted above it.
template <int kDwords>
__device__ __forceinline__ void wait_batch(RawVec<kDwords>& v0, RawVec<kDwords>& v1, RawVec<kDwords>& v2, RawVec<kDwords>& v3) {
  if constexpr (kDwords <= 4) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v0.a), "+v"(v1.a), "+v"(v2.a), "+v"(v3.a) : : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(v0.a), "+v"(v1.a), "+v"(v2.a), "+v"(v3.a), "+v"(v0.b), "+v"(v1.b), "+v"(v2.b), "+v"(v3.b)
                 : : "memory");
  }
}

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
    const int steps = m4 >> 2;
    // one descriptor per problem: offsets >= num_records read as 0 (inactive lanes, tail steps)
    const unsigned prob_bytes = unsigned(m4) * unsigned(RS) * unsigned(sizeof(T));
    const i32x4 rsrc = make_rsrc(prob, prob_bytes);
    const unsigned voff = active ? unsigned((k * RS + c * NB) * int(sizeof(T))) : 0x80000000u;
    const unsigned step_bytes = unsigned(4 * RS) * unsigned(sizeof(T));
    constexpr int U = 4;  // steps per batch; the next batch's loads are in flight while this one computes
    constexpr int kDw = NB * int(sizeof(T)) / 4;
    RawVec<kDw> nxt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) nxt[u].issue(rsrc, voff, unsigned(u) * step_bytes);
    wait_batch<kDw>(nxt[0], nxt[1], nxt[2], nxt[3]);
    for (int s0 = 0; s0 < steps; s0 += U) {
      T cur[U][NB];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        unsigned raw[kDw];
        nxt[u].get(raw);
        __builtin_memcpy(&cur[u][0], &raw[0], sizeof(raw));
      }
      // prefetch the next U steps (past the end: reads 0); pinned here, ahead of this batch's math
      const unsigned soff0 = unsigned(s0 + U) * step_bytes;
#pragma unroll
      for (int u = 0; u < U; ++u) nxt[u].issue(rsrc, voff, soff0 + unsigned(u) * step_bytes);
      __builtin_amdgcn_sched_barrier(0);
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
      __builtin_amdgcn_sched_barrier(0);
      wait_batch<kDw>(nxt[0], nxt[1], nxt[2], nxt[3]);
    }
    if (WANT_H) return T(0);
    return wave_allreduce_sum(csum);
  }

  // Scatter the Gram tiles: g[q] (q<n), undamped diagonal hd[q], cost = G[n][n].
  // Returns the cost (wave-uniform).  g / hd are LDS (or global) arrays of n.
  __device__ __forceinline__ T extract_g_diag_cost(T* __restrict__ g, T* __restrict__ hd, const int n,
                                                   const int lane_in, T* __restrict__ cost_slot) const {
    // Opaque copy of the lane id: the ~40 per-element indices / predicates below are loop-invariant
    // across LM iterations, and LICM would otherwise hoist them out of the problem loop and pin
    // ~100 VGPRs + ~300 SGPRs across the hot accumulate loop.  Recomputing them per call is free.
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
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
  __device__ __forceinline__ void write_sym(O* __restrict__ M, const int LD, const int n, const int lane_in) const {
    int lane = lane_in;
    asm volatile("" : "+v"(lane));  // see extract_g_diag_cost: keep the index math out of LICM's reach
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
