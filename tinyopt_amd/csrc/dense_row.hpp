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
// Register class of the Gram accumulators in the inline-asm MFMAs: AGPRs ("+a").  (-DTOA_ACC_VGPR, "+v", served the team form of
// round 5 and a three-waves-per-SIMD experiment of round 6 — profiles/r06_ab_log.md: under a launch bound below 256 registers
// hipcc splits the budget of a kernel whose inline asm names AGPRs half and half between the two files.)
#ifdef TOA_ACC_VGPR
#define TOA_ACC "+v"
#else
#define TOA_ACC "+a"
#endif
#include <type_traits>
#include "robust.hpp"
#include "wave_utils.hpp"

namespace toa {

#define TOA_C ,
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
#define TOA_MF32(D, A, B) "v_mfma_f32_16x16x4_f32 %" #D ", %" #A ", %" #B ", %" #D "\n\t"
};
template <>
struct Mfma<double> {
  using Acc = double __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ Acc fma(double a, double b, Acc c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane&15, row = (lane>>4) + 4*reg
  static __device__ __forceinline__ int out_row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
#define TOA_MF64(D, A, B) "v_mfma_f64_16x16x4_f64 %" #D ", %" #A ", %" #B ", %" #D "\n\t"
};

// One Gram step = the NB(NB+1)/2 upper block-pair MFMAs of 4 residual rows, accumulating IN PLACE.
// Inline asm with "+a" (AGPR, tied in/out) operands: with the builtin, hipcc's register allocator
// rotates the accumulator tuples between the unrolled steps and pays ~100 v_accvgpr_mov/read/write
// per 40 MFMAs to undo it.  `s_nop 1` covers the VALU-write -> MFMA-operand wait states that hipcc
// does not insert inside an asm statement (cdna_hip_programming.md §5.7 item 2).
// Primary template: any NB, one asm statement per tile (used for NB > 4, the workgroup-per-problem kernel of
// large_fused.hip; NB <= 4 has the hand-ordered single-statement specialisations below).  The accumulator tuples stay tied
// in place by the "+a" constraints; the s_nop covers VALU write -> MFMA read on the first tile and is hidden behind the
// matrix pipe on the others.
template <typename T, int NB>
struct GramStep {
  using Acc = typename Mfma<T>::Acc;
  static __device__ __forceinline__ constexpr int tile(int i, int j) { return i * NB - i * (i - 1) / 2 + (j - i); }
  template <int I, int J>
  static __device__ __forceinline__ void from(Acc* acc, const T* w) {
    if constexpr (I < NB) {
      if constexpr (sizeof(T) == 4)
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : TOA_ACC(acc[tile(I, J)]) : "v"(w[I]), "v"(w[J]));
      else
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : TOA_ACC(acc[tile(I, J)]) : "v"(w[I]), "v"(w[J]));
      if constexpr (J + 1 < NB) from<I, J + 1>(acc, w);
      else from<I + 1, I + 1>(acc, w);
    }
  }
  // ONE s_nop in front of the step, in a statement that names every operand of the step as an input: whatever VALU
  // instruction produces w[j] is then ordered before it, and no VALU write can sit directly in front of a later MFMA that
  // reads it.  (A nop in front of every MFMA costs an issue slot each: 36 x 8 cycles per step at NB = 8, measured as
  // 175 us instead of ~147 us per accumulate pass of the n = 128 benchmark.)
  static __device__ __forceinline__ void run(Acc* acc, const T* w) {
    static_assert(NB <= 8, "operand list below");
    constexpr int L = NB - 1;  // blocks beyond NB repeat the last one
    asm volatile("s_nop 1" : : "v"(w[0]), "v"(w[1 < L ? 1 : L]), "v"(w[2 < L ? 2 : L]), "v"(w[3 < L ? 3 : L]), "v"(w[4 < L ? 4 : L]),
                 "v"(w[5 < L ? 5 : L]), "v"(w[6 < L ? 6 : L]), "v"(w[L]));
    from<0, 0>(acc, w);
  }
  static __device__ __forceinline__ void run_tail(Acc* acc, const T* w, int last_in) {
    run(acc, w);
    const int last = __builtin_amdgcn_readfirstlane(last_in);
    asm volatile("s_cmp_eq_u32 %[last], 0\n\ts_cbranch_scc1 2\n\ts_nop 15\n\ts_nop 15" : : [last] "s"(last) : "scc", "memory");
  }
  // Only the tiles T0 <= tile(I, J) < T1 (row-major order over the upper block triangle): for accumulator sets that do not fit
  // the register file in one go (fp64, NB >= 7: large_fused.hip makes two passes over the rows, half of the tiles each).
  template <int T0, int T1, int I, int J>
  static __device__ __forceinline__ void from_range(Acc* acc, const T* w) {
    if constexpr (I < NB) {
      if constexpr (tile(I, J) >= T0 && tile(I, J) < T1) {
        if constexpr (sizeof(T) == 4)
          asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : TOA_ACC(acc[tile(I, J)]) : "v"(w[I]), "v"(w[J]));
        else
          asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : TOA_ACC(acc[tile(I, J)]) : "v"(w[I]), "v"(w[J]));
      }
      if constexpr (J + 1 < NB) from_range<T0, T1, I, J + 1>(acc, w);
      else from_range<T0, T1, I + 1, I + 1>(acc, w);
    }
  }
  template <int T0, int T1>
  static __device__ __forceinline__ void run_range(Acc* acc, const T* w, int last_in) {
    static_assert(NB <= 8, "operand list below");
    constexpr int L = NB - 1;
    asm volatile("s_nop 1" : : "v"(w[0]), "v"(w[1 < L ? 1 : L]), "v"(w[2 < L ? 2 : L]), "v"(w[3 < L ? 3 : L]), "v"(w[4 < L ? 4 : L]),
                 "v"(w[5 < L ? 5 : L]), "v"(w[6 < L ? 6 : L]), "v"(w[L]));
    from_range<T0, T1, 0, 0>(acc, w);
    const int last = __builtin_amdgcn_readfirstlane(last_in);
    asm volatile("s_cmp_eq_u32 %[last], 0\n\ts_cbranch_scc1 2\n\ts_nop 15\n\ts_nop 15" : : [last] "s"(last) : "scc", "memory");
  }
};
#define TOA_GRAM_STEP(T, TAG, NBV, BODY, ACCS, WS)                                                      \
  template <>                                                                                           \
  struct GramStep<T, NBV> {                                                                             \
    using Acc = typename Mfma<T>::Acc;                                                                  \
    static __device__ __forceinline__ void run(Acc* acc, const T* w) {                                  \
      asm volatile("s_nop 1\n\t" BODY : ACCS : WS);                                                    \
    }                                                                                                   \
    /* Same, and when `last` != 0 (wave-uniform SGPR) wait out the MFMA pipeline INSIDE the statement:  \
       hipcc cannot see the XDL-write -> accvgpr-read hazard of an asm MFMA, and any accumulator copy   \
       it places after the loop's final step must not overtake the matrix core. */                      \
    static __device__ __forceinline__ void run_tail(Acc* acc, const T* w, int last_in) {                \
      const int last = __builtin_amdgcn_readfirstlane(last_in); /* "s" needs a provably uniform value */ \
      asm volatile("s_nop 1\n\t" BODY                                                                  \
                   "s_cmp_eq_u32 %[last], 0\n\ts_cbranch_scc1 2\n\ts_nop 15\n\ts_nop 15\n\t"            \
                   : ACCS : WS TOA_C [last] "s"(last) : "scc");                                         \
    }                                                                                                   \
  };
// tile order matches DenseRowGram::tile(i, j): (0,0),(0,1)..(0,NB-1),(1,1)...
TOA_GRAM_STEP(float, F32, 1, TOA_MF32(0, 1, 1), TOA_ACC(acc[0]), "v"(w[0]))
TOA_GRAM_STEP(float, F32, 2, TOA_MF32(0, 3, 3) TOA_MF32(1, 3, 4) TOA_MF32(2, 4, 4),
              TOA_ACC(acc[0]) TOA_C TOA_ACC(acc[1]) TOA_C TOA_ACC(acc[2]), "v"(w[0]) TOA_C "v"(w[1]))
TOA_GRAM_STEP(float, F32, 3,
              TOA_MF32(0, 6, 6) TOA_MF32(1, 6, 7) TOA_MF32(2, 6, 8) TOA_MF32(3, 7, 7) TOA_MF32(4, 7, 8) TOA_MF32(5, 8, 8),
              TOA_ACC(acc[0]) TOA_C TOA_ACC(acc[1]) TOA_C TOA_ACC(acc[2]) TOA_C TOA_ACC(acc[3]) TOA_C TOA_ACC(acc[4]) TOA_C TOA_ACC(acc[5]),
              "v"(w[0]) TOA_C "v"(w[1]) TOA_C "v"(w[2]))
TOA_GRAM_STEP(float, F32, 4,
              TOA_MF32(0, 10, 10) TOA_MF32(1, 10, 11) TOA_MF32(2, 10, 12) TOA_MF32(3, 10, 13) TOA_MF32(4, 11, 11)
              TOA_MF32(5, 11, 12) TOA_MF32(6, 11, 13) TOA_MF32(7, 12, 12) TOA_MF32(8, 12, 13) TOA_MF32(9, 13, 13),
              TOA_ACC(acc[0]) TOA_C TOA_ACC(acc[1]) TOA_C TOA_ACC(acc[2]) TOA_C TOA_ACC(acc[3]) TOA_C TOA_ACC(acc[4]) TOA_C TOA_ACC(acc[5]) TOA_C
              TOA_ACC(acc[6]) TOA_C TOA_ACC(acc[7]) TOA_C TOA_ACC(acc[8]) TOA_C TOA_ACC(acc[9]),
              "v"(w[0]) TOA_C "v"(w[1]) TOA_C "v"(w[2]) TOA_C "v"(w[3]))
TOA_GRAM_STEP(double, F64, 1, TOA_MF64(0, 1, 1), TOA_ACC(acc[0]), "v"(w[0]))
TOA_GRAM_STEP(double, F64, 2, TOA_MF64(0, 3, 3) TOA_MF64(1, 3, 4) TOA_MF64(2, 4, 4),
              TOA_ACC(acc[0]) TOA_C TOA_ACC(acc[1]) TOA_C TOA_ACC(acc[2]), "v"(w[0]) TOA_C "v"(w[1]))
TOA_GRAM_STEP(double, F64, 3,
              TOA_MF64(0, 6, 6) TOA_MF64(1, 6, 7) TOA_MF64(2, 6, 8) TOA_MF64(3, 7, 7) TOA_MF64(4, 7, 8) TOA_MF64(5, 8, 8),
              TOA_ACC(acc[0]) TOA_C TOA_ACC(acc[1]) TOA_C TOA_ACC(acc[2]) TOA_C TOA_ACC(acc[3]) TOA_C TOA_ACC(acc[4]) TOA_C TOA_ACC(acc[5]),
              "v"(w[0]) TOA_C "v"(w[1]) TOA_C "v"(w[2]))
TOA_GRAM_STEP(double, F64, 4,
              TOA_MF64(0, 10, 10) TOA_MF64(1, 10, 11) TOA_MF64(2, 10, 12) TOA_MF64(3, 10, 13) TOA_MF64(4, 11, 11)
              TOA_MF64(5, 11, 12) TOA_MF64(6, 11, 13) TOA_MF64(7, 12, 12) TOA_MF64(8, 12, 13) TOA_MF64(9, 13, 13),
              TOA_ACC(acc[0]) TOA_C TOA_ACC(acc[1]) TOA_C TOA_ACC(acc[2]) TOA_C TOA_ACC(acc[3]) TOA_C TOA_ACC(acc[4]) TOA_C TOA_ACC(acc[5]) TOA_C
              TOA_ACC(acc[6]) TOA_C TOA_ACC(acc[7]) TOA_C TOA_ACC(acc[8]) TOA_C TOA_ACC(acc[9]),
              "v"(w[0]) TOA_C "v"(w[1]) TOA_C "v"(w[2]) TOA_C "v"(w[3]))
#undef TOA_GRAM_STEP

// sin/cos for the DenseRow functor.  Branch-free Cody-Waite reduction by pi/2 (FMA, 3 / 2 constants)
// + Cephes minimax polynomials on [-pi/4, pi/4]: <= ~1.5 ulp for |t| < ~1e4, which covers the
// model's domain (|a_i.x| <= n * |x|_inf).  libm's sincos carries a Payne-Hanek path whose register
// footprint would halve the occupancy of the whole fused kernel for arguments that never occur.
#if !defined(TOA_POLY_SINCOS)
// fp32: v_sin_f32 / v_cos_f32 (inputs in revolutions) — 3 VALU instructions instead of ~28.  Measured on
// MI355X against an fp64 reference (tests/tools/accuracy_probe.py, n=50, m=2000): g / H / cost errors are
// indistinguishable from the polynomial version and smaller than the fp32 CPU oracle's own error.
// The accumulate pass is VALU-issue-bound (f32 MFMA shares the VALU's issue slots), so this is
// worth ~15 % of the pass.  -DTOA_POLY_SINCOS selects the ~1.5 ulp polynomial instead.
__device__ __forceinline__ void sincos_t(float t, float* s, float* c) {
  *s = __sinf(t);
  *c = __cosf(t);
}
#else
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
#endif
// Horner step acc * z + c as a three-address v_fma_f64: hipcc otherwise emits v_fmac_f64 into a COPY of the
// coefficient (one v_mov_b64 per step: the coefficients are loop-invariant registers it must not clobber).
__device__ __forceinline__ double horner_f64(double acc, double z, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(acc), "v"(z), "v"(c));
  return d;
}
// The same polynomial with every coefficient as the (one) scalar operand of its v_fma_f64: for a loop whose registers are
// the bound (pass16s) — twelve coefficient pairs are 24 vector registers otherwise.  Same operations in the same order as
// sincos_t: same bits.
__device__ __forceinline__ double horner_f64_s(double acc, double z, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(acc), "v"(z), "s"(c));
  return d;
}
__device__ __forceinline__ void sincos_t_s(double t, double* s, double* c) {
  const double j = rint(t * 0.63661977236758134308);
  double y = fma(-j, 1.5707963267948966, t);
  y = fma(-j, 6.123233995736766e-17, y);
  const double z = y * y;
  double ps = horner_f64_s(z, 1.58962301576546568060e-10, -2.50507477628578072866e-8);
  ps = horner_f64_s(ps, z, 2.75573136213857245213e-6);
  ps = horner_f64_s(ps, z, -1.98412698295895385996e-4);
  ps = horner_f64_s(ps, z, 8.33333333332211858878e-3);
  ps = horner_f64_s(ps, z, -1.66666666666666307295e-1);
  ps = fma(ps, z * y, y);
  double pc = horner_f64_s(z, -1.13585365213876817300e-11, 2.08757008419747316778e-9);
  pc = horner_f64_s(pc, z, -2.75573141792967388112e-7);
  pc = horner_f64_s(pc, z, 2.48015872888517045348e-5);
  pc = horner_f64_s(pc, z, -1.38888888888730564116e-3);
  pc = horner_f64_s(pc, z, 4.16666666666665929218e-2);
  pc = fma(pc, z * z, fma(-0.5, z, 1.0));
  const int q = int(j) & 3;
  const double sv = (q & 1) ? pc : ps;
  const double cv = (q & 1) ? ps : pc;
  *s = (q & 2) ? -sv : sv;
  *c = ((q + 1) & 2) ? -cv : cv;
}
__device__ __forceinline__ void sincos_t(double t, double* s, double* c) {
  const double j = rint(t * 0.63661977236758134308);
  double y = fma(-j, 1.5707963267948966, t);
  y = fma(-j, 6.123233995736766e-17, y);
  const double z = y * y;
  double ps = 1.58962301576546568060e-10;
  ps = horner_f64(ps, z, -2.50507477628578072866e-8);
  ps = horner_f64(ps, z, 2.75573136213857245213e-6);
  ps = horner_f64(ps, z, -1.98412698295895385996e-4);
  ps = horner_f64(ps, z, 8.33333333332211858878e-3);
  ps = horner_f64(ps, z, -1.66666666666666307295e-1);
  ps = fma(ps, z * y, y);
  double pc = -1.13585365213876817300e-11;
  pc = horner_f64(pc, z, 2.08757008419747316778e-9);
  pc = horner_f64(pc, z, -2.75573141792967388112e-7);
  pc = horner_f64(pc, z, 2.48015872888517045348e-5);
  pc = horner_f64(pc, z, -1.38888888888730564116e-3);
  pc = horner_f64(pc, z, 4.16666666666665929218e-2);
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
  // soff must be wave-uniform (callers readfirstlane it once per batch).  NOP = false: this load directly follows
  // another asm load that already covered the SALU-write -> VMEM-read wait states for the same soff.
  template <bool NOP = true, int IMM = 0>   // IMM: immediate byte offset (12 bits) added to voff + soff
  __device__ __forceinline__ void issue(i32x4 r, unsigned voff, unsigned soff) {
    if constexpr (NOP) {
      asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, %3 offen offset:%4" : "=v"(a) : "v"(voff), "s"(r), "s"(soff), "n"(IMM) : "memory");
    } else {
      asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:%4" : "=v"(a) : "v"(voff), "s"(r), "s"(soff), "n"(IMM) : "memory");
    }
  }
  __device__ __forceinline__ void get(unsigned* o) const { o[0] = a; }
};
template <>
struct RawVec<2> {
  u32x2 a;
  // soff must be wave-uniform (callers readfirstlane it once per batch).  NOP = false: this load directly follows
  // another asm load that already covered the SALU-write -> VMEM-read wait states for the same soff.
  template <bool NOP = true, int IMM = 0>   // IMM: immediate byte offset (12 bits) added to voff + soff
  __device__ __forceinline__ void issue(i32x4 r, unsigned voff, unsigned soff) {
    if constexpr (NOP) {
      asm volatile("s_nop 4\n\tbuffer_load_dwordx2 %0, %1, %2, %3 offen offset:%4" : "=v"(a) : "v"(voff), "s"(r), "s"(soff), "n"(IMM) : "memory");
    } else {
      asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen offset:%4" : "=v"(a) : "v"(voff), "s"(r), "s"(soff), "n"(IMM) : "memory");
    }
  }
  __device__ __forceinline__ void get(unsigned* o) const { o[0] = a[0]; o[1] = a[1]; }
};
template <>
struct RawVec<3> {
  u32x3 a;
  // soff must be wave-uniform (callers readfirstlane it once per batch).  NOP = false: this load directly follows
  // another asm load that already covered the SALU-write -> VMEM-read wait states for the same soff.
  template <bool NOP = true, int IMM = 0>   // IMM: immediate byte offset (12 bits) added to voff + soff
  __device__ __forceinline__ void issue(i32x4 r, unsigned voff, unsigned soff) {
    if constexpr (NOP) {
      asm volatile("s_nop 4\n\tbuffer_load_dwordx3 %0, %1, %2, %3 offen offset:%4" : "=v"(a) : "v"(voff), "s"(r), "s"(soff), "n"(IMM) : "memory");
    } else {
      asm volatile("buffer_load_dwordx3 %0, %1, %2, %3 offen offset:%4" : "=v"(a) : "v"(voff), "s"(r), "s"(soff), "n"(IMM) : "memory");
    }
  }
  __device__ __forceinline__ void get(unsigned* o) const { o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; }
};
template <>
struct RawVec<4> {
  u32x4 a;
  // soff must be wave-uniform (callers readfirstlane it once per batch).  NOP = false: this load directly follows
  // another asm load that already covered the SALU-write -> VMEM-read wait states for the same soff.
  template <bool NOP = true, int IMM = 0>   // IMM: immediate byte offset (12 bits) added to voff + soff
  __device__ __forceinline__ void issue(i32x4 r, unsigned voff, unsigned soff) {
    if constexpr (NOP) {
      asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(a) : "v"(voff), "s"(r), "s"(soff), "n"(IMM) : "memory");
    } else {
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(a) : "v"(voff), "s"(r), "s"(soff), "n"(IMM) : "memory");
    }
  }
  __device__ __forceinline__ void get(unsigned* o) const { o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; }
};
template <>
struct RawVec<6> {
  u32x4 a;
  u32x2 b;
  // soff must be wave-uniform (callers readfirstlane it once per batch).  NOP = false: this load directly follows
  // another asm load that already covered the SALU-write -> VMEM-read wait states for the same soff.
  template <bool NOP = true, int IMM = 0>
  __device__ __forceinline__ void issue(i32x4 r, unsigned voff, unsigned soff) {
    if constexpr (NOP) {
      asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %3, %4 offen offset:%5\n\tbuffer_load_dwordx2 %1, %2, %3, %4 offen offset:%6"
                 : "=&v"(a), "=&v"(b) : "v"(voff), "s"(r), "s"(soff), "n"(IMM), "n"(IMM + 16) : "memory");
    } else {
      asm volatile("buffer_load_dwordx4 %0, %2, %3, %4 offen offset:%5\n\tbuffer_load_dwordx2 %1, %2, %3, %4 offen offset:%6"
                 : "=&v"(a), "=&v"(b) : "v"(voff), "s"(r), "s"(soff), "n"(IMM), "n"(IMM + 16) : "memory");
    }
  }
  __device__ __forceinline__ void get(unsigned* o) const {
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1];
  }
};
template <>
struct RawVec<8> {
  u32x4 a, b;
  // soff must be wave-uniform (callers readfirstlane it once per batch).  NOP = false: this load directly follows
  // another asm load that already covered the SALU-write -> VMEM-read wait states for the same soff.
  template <bool NOP = true, int IMM = 0>
  __device__ __forceinline__ void issue(i32x4 r, unsigned voff, unsigned soff) {
    if constexpr (NOP) {
      asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %3, %4 offen offset:%5\n\tbuffer_load_dwordx4 %1, %2, %3, %4 offen offset:%6"
                 : "=&v"(a), "=&v"(b) : "v"(voff), "s"(r), "s"(soff), "n"(IMM), "n"(IMM + 16) : "memory");
    } else {
      asm volatile("buffer_load_dwordx4 %0, %2, %3, %4 offen offset:%5\n\tbuffer_load_dwordx4 %1, %2, %3, %4 offen offset:%6"
                 : "=&v"(a), "=&v"(b) : "v"(voff), "s"(r), "s"(soff), "n"(IMM), "n"(IMM + 16) : "memory");
    }
  }
  __device__ __forceinline__ void get(unsigned* o) const {
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
  }
};
// 5 / 7 / 10 / 12 / 14 dwords (NB = 5, 7 in fp32; NB = 5, 6, 7 in fp64): the natural-layout rows of large_fused.hip
#define TOA_RAWVEC_ISSUE(ASM, OUTS, INS)                                                                   \
  template <bool NOP = true, int IMM = 0>                                                                  \
  __device__ __forceinline__ void issue(i32x4 r, unsigned voff, unsigned soff) {                           \
    if constexpr (NOP) asm volatile("s_nop 4\n\t" ASM : OUTS : INS : "memory");                           \
    else asm volatile(ASM : OUTS : INS : "memory");                                                        \
  }
template <>
struct RawVec<5> {
  u32x4 a;
  unsigned b;
  TOA_RAWVEC_ISSUE("buffer_load_dwordx4 %0, %2, %3, %4 offen offset:%5\n\tbuffer_load_dword %1, %2, %3, %4 offen offset:%6",
                   "=&v"(a) TOA_C "=&v"(b), "v"(voff) TOA_C "s"(r) TOA_C "s"(soff) TOA_C "n"(IMM) TOA_C "n"(IMM + 16))
  __device__ __forceinline__ void get(unsigned* o) const { o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b; }
};
template <>
struct RawVec<7> {
  u32x4 a;
  u32x3 b;
  TOA_RAWVEC_ISSUE("buffer_load_dwordx4 %0, %2, %3, %4 offen offset:%5\n\tbuffer_load_dwordx3 %1, %2, %3, %4 offen offset:%6",
                   "=&v"(a) TOA_C "=&v"(b), "v"(voff) TOA_C "s"(r) TOA_C "s"(soff) TOA_C "n"(IMM) TOA_C "n"(IMM + 16))
  __device__ __forceinline__ void get(unsigned* o) const {
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2];
  }
};
template <>
struct RawVec<10> {
  u32x4 a, b;
  u32x2 c;
  TOA_RAWVEC_ISSUE("buffer_load_dwordx4 %0, %3, %4, %5 offen offset:%6\n\tbuffer_load_dwordx4 %1, %3, %4, %5 offen offset:%7\n\t"
                   "buffer_load_dwordx2 %2, %3, %4, %5 offen offset:%8",
                   "=&v"(a) TOA_C "=&v"(b) TOA_C "=&v"(c),
                   "v"(voff) TOA_C "s"(r) TOA_C "s"(soff) TOA_C "n"(IMM) TOA_C "n"(IMM + 16) TOA_C "n"(IMM + 32))
  __device__ __forceinline__ void get(unsigned* o) const {
    for (int i = 0; i < 4; ++i) { o[i] = a[i]; o[4 + i] = b[i]; }
    o[8] = c[0]; o[9] = c[1];
  }
};
template <>
struct RawVec<12> {
  u32x4 a, b, c;
  TOA_RAWVEC_ISSUE("buffer_load_dwordx4 %0, %3, %4, %5 offen offset:%6\n\tbuffer_load_dwordx4 %1, %3, %4, %5 offen offset:%7\n\t"
                   "buffer_load_dwordx4 %2, %3, %4, %5 offen offset:%8",
                   "=&v"(a) TOA_C "=&v"(b) TOA_C "=&v"(c),
                   "v"(voff) TOA_C "s"(r) TOA_C "s"(soff) TOA_C "n"(IMM) TOA_C "n"(IMM + 16) TOA_C "n"(IMM + 32))
  __device__ __forceinline__ void get(unsigned* o) const {
    for (int i = 0; i < 4; ++i) { o[i] = a[i]; o[4 + i] = b[i]; o[8 + i] = c[i]; }
  }
};
template <>
struct RawVec<14> {
  u32x4 a, b, c;
  u32x2 d;
  TOA_RAWVEC_ISSUE("buffer_load_dwordx4 %0, %4, %5, %6 offen offset:%7\n\tbuffer_load_dwordx4 %1, %4, %5, %6 offen offset:%8\n\t"
                   "buffer_load_dwordx4 %2, %4, %5, %6 offen offset:%9\n\tbuffer_load_dwordx2 %3, %4, %5, %6 offen offset:%10",
                   "=&v"(a) TOA_C "=&v"(b) TOA_C "=&v"(c) TOA_C "=&v"(d),
                   "v"(voff) TOA_C "s"(r) TOA_C "s"(soff) TOA_C "n"(IMM) TOA_C "n"(IMM + 16) TOA_C "n"(IMM + 32) TOA_C "n"(IMM + 48))
  __device__ __forceinline__ void get(unsigned* o) const {
    for (int i = 0; i < 4; ++i) { o[i] = a[i]; o[4 + i] = b[i]; o[8 + i] = c[i]; }
    o[12] = d[0]; o[13] = d[1];
  }
};
template <>
struct RawVec<16> {
  u32x4 a, b, c, d;
  TOA_RAWVEC_ISSUE("buffer_load_dwordx4 %0, %4, %5, %6 offen offset:%7\n\tbuffer_load_dwordx4 %1, %4, %5, %6 offen offset:%8\n\t"
                   "buffer_load_dwordx4 %2, %4, %5, %6 offen offset:%9\n\tbuffer_load_dwordx4 %3, %4, %5, %6 offen offset:%10",
                   "=&v"(a) TOA_C "=&v"(b) TOA_C "=&v"(c) TOA_C "=&v"(d),
                   "v"(voff) TOA_C "s"(r) TOA_C "s"(soff) TOA_C "n"(IMM) TOA_C "n"(IMM + 16) TOA_C "n"(IMM + 32) TOA_C "n"(IMM + 48))
  __device__ __forceinline__ void get(unsigned* o) const {
    for (int i = 0; i < 4; ++i) { o[i] = a[i]; o[4 + i] = b[i]; o[8 + i] = c[i]; o[12 + i] = d[i]; }
  }
};
#undef TOA_RAWVEC_ISSUE
// Wait for every outstanding VMEM load of this wave; the operands make the four batch slots
// data-dependent on the wait so that no consumer is hoisted above it.
// kLeave: how many YOUNGER vector-memory accesses may stay outstanding (loads retire in issue order).
template <int kLeave, int kDwords>
__device__ __forceinline__ void wait_batch(RawVec<kDwords>& v0, RawVec<kDwords>& v1, RawVec<kDwords>& v2, RawVec<kDwords>& v3) {
  static_assert(kLeave >= 0 && kLeave < 64, "vmcnt is a 6-bit counter");
  if constexpr (kDwords <= 4) {
    asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(v0.a), "+v"(v1.a), "+v"(v2.a), "+v"(v3.a) : [n] "n"(kLeave) : "memory");
  } else if constexpr (kDwords <= 8) {
    asm volatile("s_waitcnt vmcnt(%[n])"
                 : "+v"(v0.a), "+v"(v1.a), "+v"(v2.a), "+v"(v3.a), "+v"(v0.b), "+v"(v1.b), "+v"(v2.b), "+v"(v3.b)
                 : [n] "n"(kLeave) : "memory");
  } else if constexpr (kDwords <= 12) {
    asm volatile("s_waitcnt vmcnt(%[n])"
                 : "+v"(v0.a), "+v"(v1.a), "+v"(v2.a), "+v"(v3.a), "+v"(v0.b), "+v"(v1.b), "+v"(v2.b), "+v"(v3.b),
                   "+v"(v0.c), "+v"(v1.c), "+v"(v2.c), "+v"(v3.c)
                 : [n] "n"(kLeave) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(%[n])"
                 : "+v"(v0.a), "+v"(v1.a), "+v"(v2.a), "+v"(v3.a), "+v"(v0.b), "+v"(v1.b), "+v"(v2.b), "+v"(v3.b),
                   "+v"(v0.c), "+v"(v1.c), "+v"(v2.c), "+v"(v3.c), "+v"(v0.d), "+v"(v1.d), "+v"(v2.d), "+v"(v3.d)
                 : [n] "n"(kLeave) : "memory");
  }
}
template <int kDwords>
constexpr int rawvec_loads() { return (kDwords + 3) / 4; }

// Host+device layout helper.
//
// A packed row is  [ main : rsm elements ][ thin : `thin` elements ].
//   main  — the columns that go through the matrix cores, NBM blocks of 16; lane c of a 16-lane row
//           group owns the NBM contiguous elements [NBM*c, NBM*c+NBM).
//   thin  — when n = 16*NBM + (0..3) the last <= 3 Jacobian columns plus b do not justify a whole
//           extra block column of MFMA tiles (for n = 50 that would be 4 of 10 tiles for 3 useful
//           columns); they are broadcast-loaded by every lane of the row group and their products are
//           accumulated on the VALU instead (`thin` = those columns + b, 1..4).  thin == 0: b lives in
//           the LAST main element (static position -> no per-element select in the hot loop).
// Row stride = rsm + thin (n = 50 fp32: 48 + 3 = 51 floats = exactly the algorithmic m(n+1) bytes).
struct DenseRowLayout {
  int nbm;   // MFMA column blocks (1..4)
  int thin;  // VALU tail elements incl. b (0..4)
  int nmr;   // real Jacobian columns in the main part
  int rsm;   // physical main length
  int rs;    // row stride in elements
  int m4;    // rows padded to a multiple of 4
  __host__ __device__ static DenseRowLayout make(int n, int m) {
    DenseRowLayout L;
    const int rem = n & 15;
    if (n >= 16 && rem + 1 <= 4) {
      L.nbm = n >> 4;
      L.thin = rem + 1;
      L.nmr = 16 * L.nbm;
      L.rsm = L.nmr;
    } else {
      L.nbm = (n + 1 + 15) / 16;
      L.thin = 0;
      L.nmr = n;
      L.rsm = L.nbm * ((n + 1 + L.nbm - 1) / L.nbm);
    }
    L.rs = L.rsm + L.thin;
    L.m4 = (m + 3) & ~3;
    return L;
  }
  __host__ __device__ size_t elems_per_problem() const { return size_t(m4) * rs; }
  // physical position of Jacobian column j (0 <= j < n) and of b inside a packed row
  __host__ __device__ int pos_col(int j) const { return j < nmr ? j : rsm + (j - nmr); }
  __host__ __device__ int pos_b() const { return thin ? rs - 1 : rsm - 1; }
};

// all-reduce over the 4 row groups of a wave (lanes with equal lane&15)
template <typename T>
__device__ __forceinline__ T kgroup_allreduce_sum(T v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

template <typename T, int NBM, int THIN>
struct DenseRowGram {
  static constexpr int NT = NBM * (NBM + 1) / 2;                // upper block pairs on the matrix cores
  static constexpr int NTT = THIN * (THIN + 1) / 2;             // thin x thin products
  static constexpr int NTM = THIN ? NBM * THIN : 1;             // main x thin products per lane
  using Acc = typename Mfma<T>::Acc;
  Acc acc[NT];
  T accT[NTM];               // accT[ti(cb, j)] = sum_rows W[row][NBM*c+cb] * thin_j   (this lane's c)
  T accTT[NTT ? NTT : 1];    // thin_j * thin_j' (j <= j'), identical on every lane after the pass

  static __device__ __forceinline__ constexpr int tile(int i, int j) { return i * NBM - i * (i - 1) / 2 + (j - i); }
  // thin x thin product (j <= j2): stored column by column so that (j, j2), (j+1, j2) are adjacent (packed FMA)
  static __device__ __forceinline__ constexpr int tt(int j, int j2) { return j2 * (j2 + 1) / 2 + j; }
  // accT index of (column block cb, thin element j): pairs of blocks adjacent (packed-FMA friendly), odd block last
  static __device__ __forceinline__ constexpr int ti(int cb, int j) {
    return cb < 2 * (NBM / 2) ? ((cb / 2) * THIN + j) * 2 + (cb & 1) : 2 * (NBM / 2) * THIN + j;
  }

  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = Acc{0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < NTM; ++t) accT[t] = T(0);
#pragma unroll
    for (int t = 0; t < (NTT ? NTT : 1); ++t) accTT[t] = T(0);
  }

#ifndef TOA_U
#define TOA_U 4
#endif
  static constexpr int U = TOA_U;  // steps (of 4 rows) per batch; one batch is in flight while the other computes
  static_assert(U == 4, "wait_batch ties exactly four slots");
  static constexpr int kDw = NBM * int(sizeof(T)) / 4;
  static constexpr int kDwT = THIN ? THIN * int(sizeof(T)) / 4 : 1;
  using Slots = RawVec<kDw>[U];
  using SlotsT = RawVec<kDwT>[U];

  // Loop-invariant per-lane operands of a pass.
  struct PassCtx {
    T xr[NBM];
    T xt[THIN > 1 ? THIN - 1 : 1];  // x of the thin columns on lane c == 0, zero elsewhere (they join the row reduction once)
    bool isB_lane;
    T mA, mB;                        // (1, 0) on ordinary lanes, (0, 1) on the lane whose last main element is b
    bool q0, q1;                     // bits of the lane's quad position: the batch step whose a_i.x this lane finishes
    int c;
    // M-estimator on each residual (ROBUST passes only; losses/robust_norms.h:20-26: cost += l, the row's J^T J and J^T r
    // scaled by s = dl/dn2 — here as sqrt(s) on the row [J | r] in front of the Gram)
    int loss;                        // TOA_LOSS_* (wave-uniform)
    T th2;
    int k;                           // row group of this lane (lane >> 4)
    int rows_real;                   // rows of the problem / chunk that exist (padding rows are zero and must not count)
    bool owner;                      // the one lane per row that books cost and inliers
    T inl;                           // inlier rows booked by this lane
  };

  // A thin-tail layout serves exactly one n (= 16 NBM + THIN - 1), so its row stride is a compile-time constant and the
  // four steps of a batch are reached through the loads' IMMEDIATE offsets from one scalar offset per batch: no s_add +
  // s_nop 4 (SALU write -> VMEM read wait states) in front of every load, one in front of the first.
  static constexpr int kStepBytesImm = THIN > 0 ? 4 * (16 * NBM + THIN) * int(sizeof(T)) : 0;
  static constexpr bool kImmOffsets = THIN > 0 && (U - 1) * kStepBytesImm + 32 <= 4095;
  static __device__ __forceinline__ void issue_batch(Slots& m, SlotsT& t, const i32x4 rsrc, const unsigned voff,
                                                     const unsigned vofft, const unsigned soff0, const unsigned step_bytes_u) {
    if constexpr (kImmOffsets) {
      static_for<U>([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        m[u].template issue<u == 0, u * kStepBytesImm>(rsrc, voff, soff0);
        t[u].template issue<false, u * kStepBytesImm>(rsrc, vofft, soff0);
      });
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        m[u].issue(rsrc, voff, soff0 + unsigned(u) * step_bytes_u);
        if (THIN) t[u].issue(rsrc, vofft, soff0 + unsigned(u) * step_bytes_u);
      }
    }
  }
#ifndef TOA_DEPTH
#define TOA_DEPTH 2
#endif
  static constexpr int kDepth = TOA_DEPTH;  // slot sets in the ring: kDepth - 1 batches are in flight while one computes
  static_assert(kDepth >= 2 && kDepth <= 4, "ring depth");
  static constexpr int kLoadsPerBatch = U * (rawvec_loads<kDw>() + (THIN ? rawvec_loads<kDwT>() : 0));
  static constexpr int kLeave = (kDepth - 2) * kLoadsPerBatch;  // younger batches that may stay outstanding at a wait
  static __device__ __forceinline__ void wait_slots(Slots& m, SlotsT& t) {
    wait_batch<kLeave, kDw>(m[0], m[1], m[2], m[3]);
    if (THIN) wait_batch<kLeave, kDwT>(t[0], t[1], t[2], t[3]);
  }

  // Unpack the load registers of one batch and form each lane's partial a_i.x of its U steps.
  __device__ __forceinline__ void batch_dots(const Slots& m, const SlotsT& tv, const PassCtx& pc, T (&wa)[U][NBM],
                                             T (&va)[U][THIN ? THIN : 1], T (&part)[U]) const {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      unsigned raw[kDw];
      m[u].get(raw);
      __builtin_memcpy(&wa[u][0], &raw[0], sizeof(raw));
      if (THIN) {
        unsigned rawT[kDwT];
        tv[u].get(rawT);
        __builtin_memcpy(&va[u][0], &rawT[0], sizeof(T) * THIN);
      }
      T pu = 0;
#pragma unroll
      for (int cb = 0; cb < NBM; ++cb) pu += wa[u][cb] * pc.xr[cb];
#pragma unroll
      for (int j = 0; j + 1 < THIN; ++j) pu += va[u][j] * pc.xt[j];
      part[u] = pu;
    }
  }

  // One step of the batch once the scale sc = 1 + 0.1 cos(a_i.x) and rbase = a_i.x + 0.1 sin(a_i.x) of its four rows are
  // known: J = sc a in registers, r = rbase - b, the step's MFMAs and thin products (or, cost only, r^2).  w / v: the
  // step's main / thin operands; row_u: index of the step's first row (ROBUST bookkeeping).
  // T0, T1: tile range of this pass (all of them by default); THINP: form the thin products too.
  template <bool WANT_H, bool TAIL_STEP, bool ROBUST, int T0 = 0, int T1 = NT, bool THINP = true>
  __device__ __forceinline__ void apply_step(T (&w)[NBM], T (&v)[THIN ? THIN : 1], T sc, const T rbase, PassCtx& pc, T& csum,
                                             const int last, const int row_u) {
    T rsq = T(0);  // ROBUST: sqrt(s) * r of this row
    if constexpr (ROBUST) {
      // the residual of this row on every lane of its row group: the thin tail is broadcast-loaded (all 16 lanes hold
      // b); with b in the main block only its owner has it, and a 16-lane reduction hands r to the others
      T r;
      if constexpr (THIN == 0) r = row16_allreduce_sum(pc.isB_lane ? rbase - w[NBM - 1] : T(0));
      else r = rbase - v[THIN - 1];
      const T n2 = r * r;
      T l, sw;
      robust_norm(pc.loss, n2, pc.th2, l, sw);
      const bool valid = row_u + pc.k < pc.rows_real;
      const bool book = pc.owner && valid;
      if constexpr (THINP) {                           // (a second half-tile pass over the same rows books nothing)
        csum += book ? l : T(0);                       // Cost += l
        pc.inl += (book && n2 <= pc.th2) ? T(1) : T(0);   // cost.h:84-95: inliers are the residuals inside the threshold
      }
      const T sq = r_sqrt(sw);
      sc *= sq;
      rsq = r * sq;
    }
    if constexpr (sizeof(T) == 4) {  // J = sc * a, two columns per v_pk_mul_f32
      using f2 = float __attribute__((ext_vector_type(2)));
      constexpr int kScaled = THIN == 0 ? NBM - 1 : NBM;  // THIN == 0: the last main element may be b
#pragma unroll
      for (int cb = 0; cb + 1 < kScaled; cb += 2) {
        const f2 wp = f2{w[cb], w[cb + 1]} * f2{sc, sc};
        w[cb] = wp[0];
        w[cb + 1] = wp[1];
      }
      if constexpr (kScaled & 1) w[kScaled - 1] *= sc;
#pragma unroll
      for (int j = 0; j + 2 < THIN; j += 2) {
        const f2 vp = f2{v[j], v[j + 1]} * f2{sc, sc};
        v[j] = vp[0];
        v[j + 1] = vp[1];
      }
      if constexpr (THIN > 1 && ((THIN - 1) & 1)) v[THIN - 2] *= sc;
    } else {
#pragma unroll
      for (int cb = 0; cb + 1 < NBM; ++cb) w[cb] *= sc;
      if (THIN != 0) w[NBM - 1] *= sc;
#pragma unroll
      for (int j = 0; j + 1 < THIN; ++j) v[j] *= sc;
    }
    // THIN == 0: the lane holding b turns it into r = rbase - b, every other lane scales its column by sc; written as
    // one FMA with per-lane constants (mA, mB) = (1, 0) / (0, 1) instead of a select (fp64: no exec-mask branch)
    if constexpr (ROBUST) {
      if constexpr (THIN == 0) w[NBM - 1] = pc.isB_lane ? rsq : w[NBM - 1] * sc;
      else v[THIN - 1] = rsq;
    } else {
      if constexpr (THIN == 0) w[NBM - 1] = fma(w[NBM - 1], fma(sc, pc.mA, -pc.mB), rbase * pc.mB);
      else v[THIN - 1] = rbase - v[THIN - 1];
    }
    if (WANT_H) {
#if defined(TOA_SPLIT_MFMA)
#pragma unroll
      for (int i = 0; i < NBM; ++i)
#pragma unroll
        for (int j = i; j < NBM; ++j) {
          if constexpr (sizeof(T) == 4)
            asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : TOA_ACC(acc[tile(i, j)]) : "v"(w[i]), "v"(w[j]));
          else
            asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : TOA_ACC(acc[tile(i, j)]) : "v"(w[i]), "v"(w[j]));
        }
#elif !defined(TOA_ABL_NOMFMA)
      if constexpr (T0 != 0 || T1 != NT) GramStep<T, NBM>::template run_range<T0, T1>(acc, w, TAIL_STEP ? last : 0);
      else if (TAIL_STEP) GramStep<T, NBM>::run_tail(acc, w, last);
      else GramStep<T, NBM>::run(acc, w);
#else
#pragma unroll
      for (int cb = 0; cb < NBM; ++cb) asm volatile("" ::"v"(w[cb]));
#endif
#ifndef TOA_ABL_NOTHIN
      if constexpr (!THINP) {
      } else if constexpr (sizeof(T) == 4 && THIN > 0) {
        // v_pk_fma_f32: two column blocks per instruction; accT is laid out so that the pair (cb, cb+1) of a
        // given j sits in adjacent elements (see ti()), w[cb], w[cb+1] are adjacent lanes of the load tuple
        using f2 = float __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int pb = 0; pb < NBM / 2; ++pb) {
          const f2 wp = {w[2 * pb], w[2 * pb + 1]};
#pragma unroll
          for (int j = 0; j < THIN; ++j) {
            f2 a = {accT[ti(2 * pb, j)], accT[ti(2 * pb + 1, j)]};
            a += wp * f2{v[j], v[j]};
            accT[ti(2 * pb, j)] = a[0];
            accT[ti(2 * pb + 1, j)] = a[1];
          }
        }
        if constexpr (NBM & 1) {
#ifndef TOA_NO_THIN_PAIR_LAST
          // the odd block against the thin columns two at a time: (v[j], v[j+1]) is an aligned pair of the load tuple
#pragma unroll
          for (int j = 0; j + 1 < THIN; j += 2) {
            f2 a = {accT[ti(NBM - 1, j)], accT[ti(NBM - 1, j + 1)]};
            a += f2{w[NBM - 1], w[NBM - 1]} * f2{v[j], v[j + 1]};
            accT[ti(NBM - 1, j)] = a[0];
            accT[ti(NBM - 1, j + 1)] = a[1];
          }
          if constexpr (THIN & 1) accT[ti(NBM - 1, THIN - 1)] += w[NBM - 1] * v[THIN - 1];
#else
#pragma unroll
          for (int j = 0; j < THIN; ++j) accT[ti(NBM - 1, j)] += w[NBM - 1] * v[j];
#endif
        }
        // thin x thin corner, column j2 at a time: (j, j+1) pairs against a splat of v[j2] in one packed FMA
#pragma unroll
        for (int j2 = 0; j2 < THIN; ++j2) {
#pragma unroll
          for (int j = 0; j + 1 <= j2; j += 2) {
            f2 a = {accTT[tt(j, j2)], accTT[tt(j + 1, j2)]};
            a += f2{v[j], v[j + 1]} * f2{v[j2], v[j2]};
            accTT[tt(j, j2)] = a[0];
            accTT[tt(j + 1, j2)] = a[1];
          }
          if ((j2 & 1) == 0) accTT[tt(j2, j2)] += v[j2] * v[j2];
        }
      } else {
#pragma unroll
        for (int cb = 0; cb < NBM; ++cb)
#pragma unroll
          for (int j = 0; j < THIN; ++j) accT[ti(cb, j)] += w[cb] * v[j];
#pragma unroll
        for (int j = 0; j < THIN; ++j)
#pragma unroll
          for (int j2 = j; j2 < THIN; ++j2) accTT[tt(j, j2)] += v[j] * v[j2];
      }
#else
#pragma unroll
      for (int j = 0; j < THIN; ++j) asm volatile("" ::"v"(v[j]));
#endif
    } else if constexpr (!ROBUST) {
      if constexpr (THIN == 0) csum += pc.isB_lane ? w[NBM - 1] * w[NBM - 1] : T(0);
      else csum += (pc.c == 0) ? v[THIN - 1] * v[THIN - 1] : T(0);
    }
  }

  // The arithmetic of one batch (U steps of 4 rows), reading the operands straight out of the load registers
  // of `m` / `tv` (no staging copies: the other slot set is the one being refilled meanwhile).
  template <bool WANT_H, bool TAIL = false, bool ROBUST = false, int T0 = 0, int T1 = NT, bool THINP = true>
  __device__ __forceinline__ void compute_batch(const Slots& m, const SlotsT& tv, PassCtx& pc, T& csum,
                                                const int last = 0, const int row0 = 0) {
    T wa[U][NBM];
    T va[U][THIN ? THIN : 1];
    T part[U];
    batch_dots(m, tv, pc, wa, va, part);
    // a_i.x of the four steps in one transposed reduction; quad lane q then owns step q: ONE sin/cos per batch
    const T tsel = quad_transpose_reduce(part, pc.q0, pc.q1);
    T sn, cs;
#ifndef TOA_ABL_NOSINCOS
    sincos_t(tsel, &sn, &cs);
#else
    sn = tsel; cs = tsel;
#endif
    const T sc_sel = T(1) + T(0.1) * cs;
    const T rb_sel = tsel + T(0.1) * sn;
    static_for<U>([&](auto uc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value;
      apply_step<WANT_H, (TAIL && u == U - 1), ROBUST, T0, T1, THINP>(wa[u], va[u], quad_bcast<u>(sc_sel), quad_bcast<u>(rb_sel), pc,
                                                                       csum, last, row0 + 4 * u);
    });
  }


  // One step (4 residual rows, one per row group) handed in as operands instead of being loaded: w = this lane's NBM main
  // columns of W = [J | r], v = the thin columns (every lane of the row group holds the same values).  For the models
  // that PRODUCE their rows (RowModel, row_model.hpp: a user's functor) rather than stream them; the hot streaming
  // loop above keeps its own hand-scheduled copy of this arithmetic.  last != 0: final step of the pass (see run_tail).
  // TAIL: this may be the final step of the pass (then `last` != 0 waits the matrix pipe out, see run_tail).
  template <bool TAIL = true>
  __device__ __forceinline__ void add_step(T (&w)[NBM], T (&v)[THIN ? THIN : 1], const int last) {
    if constexpr (TAIL) GramStep<T, NBM>::run_tail(acc, w, last);
    else GramStep<T, NBM>::run(acc, w);
    if constexpr (sizeof(T) == 4 && THIN > 0) {   // the packed-FMA forms of apply_step (two products per v_pk_fma_f32)
      using f2 = float __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int pb = 0; pb < NBM / 2; ++pb) {
        const f2 wp = {w[2 * pb], w[2 * pb + 1]};
#pragma unroll
        for (int j = 0; j < THIN; ++j) {
          f2 a = {accT[ti(2 * pb, j)], accT[ti(2 * pb + 1, j)]};
          a += wp * f2{v[j], v[j]};
          accT[ti(2 * pb, j)] = a[0];
          accT[ti(2 * pb + 1, j)] = a[1];
        }
      }
      if constexpr (NBM & 1) {
#pragma unroll
        for (int j = 0; j + 1 < THIN; j += 2) {
          f2 a = {accT[ti(NBM - 1, j)], accT[ti(NBM - 1, j + 1)]};
          a += f2{w[NBM - 1], w[NBM - 1]} * f2{v[j], v[j + 1]};
          accT[ti(NBM - 1, j)] = a[0];
          accT[ti(NBM - 1, j + 1)] = a[1];
        }
        if constexpr (THIN & 1) accT[ti(NBM - 1, THIN - 1)] += w[NBM - 1] * v[THIN - 1];
      }
#pragma unroll
      for (int j2 = 0; j2 < THIN; ++j2) {
#pragma unroll
        for (int j = 0; j + 1 <= j2; j += 2) {
          f2 a = {accTT[tt(j, j2)], accTT[tt(j + 1, j2)]};
          a += f2{v[j], v[j + 1]} * f2{v[j2], v[j2]};
          accTT[tt(j, j2)] = a[0];
          accTT[tt(j + 1, j2)] = a[1];
        }
        if ((j2 & 1) == 0) accTT[tt(j2, j2)] += v[j2] * v[j2];
      }
    } else if constexpr (THIN > 0) {
#pragma unroll
      for (int cb = 0; cb < NBM; ++cb)
#pragma unroll
        for (int j = 0; j < THIN; ++j) accT[ti(cb, j)] += w[cb] * v[j];
#pragma unroll
      for (int j = 0; j < THIN; ++j)
#pragma unroll
        for (int j2 = j; j2 < THIN; ++j2) accTT[tt(j, j2)] += v[j] * v[j2];
    }
  }
  // End of a pass made of add_step calls: retire the matrix pipe and fold the four row groups of the thin products.
  __device__ __forceinline__ void finish_steps() {
    mfma_retire();
    if (THIN) {
#pragma unroll
      for (int t = 0; t < NTM; ++t) accT[t] = kgroup_allreduce_sum(accT[t]);
#pragma unroll
      for (int t = 0; t < NTT; ++t) accTT[t] = kgroup_allreduce_sum(accTT[t]);
    }
  }

  // Belt and braces after the loop (the final step already waited inside its own asm statement, see
  // GramStep::run_tail); tools/isa_lint.py checks the built objects for accumulator reads that could overtake an MFMA.
  __device__ __forceinline__ void mfma_retire() {
    asm volatile("s_nop 7" ::: "memory");
#pragma unroll
    for (int t = 0; t < NT; ++t) asm volatile("" : TOA_ACC(acc[t]));
  }

  // One pass over the problem's rows.  WANT_H: full Gram (K1).  !WANT_H: cost only (K2) — returns
  // the wave-reduced sum of squares.  xs: x in LDS.  prob: packed [m4][RS].
  // Software pipeline: two sets of load registers (A, B) alternate between "being refilled by buffer loads" and
  // "being consumed by the arithmetic", so no register-to-register staging copies are needed (24 v_mov per batch
  // in the single-set version = 13 % of the VALU issue slots of this issue-bound loop).
  // ROBUST: every residual goes through the M-estimator `loss` (a separate instantiation of the loop, compiled only into
  // the kernels of DenseRowModel<.., ROBUST = true>: the plain kernels' instruction stream and register budget are
  // untouched).  Returns the cost (K2, and K1 with a loss); *ninl = inlier rows, or -1 when every residual is one.
  // STAGED (fp64 n <= 15 only): the row-per-lane pass through the wave's LDS stage (pass16s) instead of pass16.
  template <bool WANT_H, bool ROBUST = false, bool STAGED = false>
  __device__ __forceinline__ T pass(const T* __restrict__ prob, const DenseRowLayout& lay, const int n,
                                    const T* __restrict__ xs, const int lane, const int loss = 0, const T th2 = T(0),
                                    const int rows_real = 0, int* ninl = nullptr, unsigned char* stage = nullptr) {
    if constexpr (!ROBUST) {
      if (ninl) *ninl = -1;
      if constexpr (kSuper16 && STAGED) return pass16s<WANT_H>(prob, lay, n, xs, lane, stage);
      else if constexpr (kSuper16) return pass16<WANT_H>(prob, lay, n, xs, lane);
    }
    const int k = lane >> 4, c = lane & 15;
    const int RS = lay.rs, rsm = lay.rsm;
    const bool active = c * NBM < rsm;
    PassCtx pc;
    pc.c = c;
#pragma unroll
    for (int cb = 0; cb < NBM; ++cb) {
      const int q = NBM * c + cb;
      pc.xr[cb] = (q < lay.nmr) ? xs[q] : T(0);  // b slot and padding contribute nothing to a_i.x
    }
#pragma unroll
    for (int j = 0; j + 1 < THIN; ++j) pc.xt[j] = (c == 0) ? xs[lay.nmr + j] : T(0);
    pc.q0 = (lane & 1) != 0;
    pc.q1 = (lane & 2) != 0;
    pc.isB_lane = (THIN == 0) && ((c + 1) * NBM == rsm);  // b = last main element
    pc.mA = pc.isB_lane ? T(0) : T(1);
    pc.mB = pc.isB_lane ? T(1) : T(0);
    pc.loss = loss;
    pc.th2 = th2;
    pc.k = k;
    pc.rows_real = rows_real;
    pc.owner = THIN == 0 ? pc.isB_lane : c == 0;
    pc.inl = T(0);
    if (WANT_H) clear();
    T csum = 0;
    const int steps = lay.m4 >> 2;
    // one descriptor per problem: offsets >= num_records read as 0 (inactive lanes, tail steps)
    const unsigned prob_bytes = unsigned(lay.m4) * unsigned(RS) * unsigned(sizeof(T));
    const i32x4 rsrc = make_rsrc(prob, prob_bytes);
    const unsigned voff = active ? unsigned((k * RS + c * NBM) * int(sizeof(T))) : 0x80000000u;
    const unsigned vofft = unsigned((k * RS + rsm) * int(sizeof(T)));  // same address for the 16 lanes of a row group
    const unsigned step_bytes = unsigned(4 * RS) * unsigned(sizeof(T));
    const unsigned step_bytes_u = unsigned(__builtin_amdgcn_readfirstlane(int(step_bytes)));
    // Ring of kDepth slot sets: while set i is consumed, the kDepth - 1 younger batches are in flight.  One exit at the
    // bottom keeps a single loop-carried copy of the accumulators (a mid-loop exit makes hipcc keep one AGPR set per
    // segment and shuffle 24 registers between them); a batch count that is not a multiple of kDepth runs up to
    // kDepth - 1 batches of zero rows (loads past the end return 0, and an all-zero row adds nothing).
    // With kDepth > 2 one set is in flight across the back-edge: tools/isa_lint.py proves no instruction touches a
    // register whose load is still outstanding (a compiler-inserted copy there would read stale data).
    Slots S[kDepth];
    SlotsT St[kDepth];
    static_for<kDepth - 1>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      issue_batch(S[i], St[i], rsrc, voff, vofft, unsigned(i * U) * step_bytes_u, step_bytes_u);
    });
    wait_slots(S[0], St[0]);
    for (int s0 = 0; s0 < steps; s0 += kDepth * U) {
      static_for<kDepth>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        constexpr int refill = (i + kDepth - 1) % kDepth;  // consumed in the previous segment
        const unsigned soff = unsigned(__builtin_amdgcn_readfirstlane(int(unsigned(s0 + (i + kDepth - 1) * U) * step_bytes_u)));
        issue_batch(S[refill], St[refill], rsrc, voff, vofft, soff, step_bytes_u);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ROBUST) {
          // The M-estimators (exp / log / atan2 / divisions, in fp64 too) need far more registers than the plain loop:
          // with loads in flight across them hipcc splits the live ranges of the load DESTINATION registers and copies
          // them before the data has landed (tools/isa_lint.py flags exactly that).  So this variant lets every load
          // land before it computes: the latency is hidden by the other waves of the SIMD only.
          wait_batch<0, kDw>(S[refill][0], S[refill][1], S[refill][2], S[refill][3]);
          if (THIN) wait_batch<0, kDwT>(St[refill][0], St[refill][1], St[refill][2], St[refill][3]);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (i == kDepth - 1)
          compute_batch<WANT_H, true, ROBUST>(S[i], St[i], pc, csum, __builtin_amdgcn_readfirstlane(int(s0 + kDepth * U >= steps)),
                                               4 * (s0 + i * U));
        else
          compute_batch<WANT_H, false, ROBUST>(S[i], St[i], pc, csum, 0, 4 * (s0 + i * U));
        __builtin_amdgcn_sched_barrier(0);
        wait_slots(S[(i + 1) % kDepth], St[(i + 1) % kDepth]);
      });
    }
    if constexpr (kDepth > 2) {  // drain the prefetches past the end before their registers are reused
      static_for<kDepth>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        wait_batch<0, kDw>(S[i][0], S[i][1], S[i][2], S[i][3]);
        if (THIN) wait_batch<0, kDwT>(St[i][0], St[i][1], St[i][2], St[i][3]);
      });
    }
    if (WANT_H) mfma_retire();
    if constexpr (ROBUST) {
      if (ninl) *ninl = int(wave_allreduce_sum(pc.inl));   // exact in T: one count per row
    }
    if (WANT_H) {
      if (THIN) {  // fold the 4 row groups: every lane ends with the totals for its column set
#pragma unroll
        for (int t = 0; t < NTM; ++t) accT[t] = kgroup_allreduce_sum(accT[t]);
#pragma unroll
        for (int t = 0; t < NTT; ++t) accTT[t] = kgroup_allreduce_sum(accTT[t]);
      }
      if constexpr (ROBUST) return wave_allreduce_sum(csum);   // sum of l; the Gram's (r, r) entry holds sum of s r^2
      return T(0);
    }
    return wave_allreduce_sum(csum);
  }

  // ---- one CHUNK of a pass (cooperative tail, kernels.hpp DenseRowModel::coop_*): the steps [step0, step1) of the problem,
  // accumulated FROM ZERO.  A pass that is the fixed-order fold of such chunk partials gives the same bits whether one wave
  // computes every chunk or the four waves of a workgroup share them — which is what lets an idle wave take rows of a
  // sibling's problem once the work queue is dry without giving up run-to-run and batch-independent results.
  // step0 is a multiple of kDepth * U (so the ring never straddles a chunk boundary); the buffer descriptor ends at the
  // chunk's last row: the prefetches past it return 0 without touching memory.  No thin-product fold here (fold_thin(),
  // once, after the chunk partials have been summed).  !WANT_H: returns the chunk's wave-reduced sum of squares.
  template <bool WANT_H>
  __device__ __forceinline__ T pass_chunk(const T* __restrict__ prob, const DenseRowLayout& lay, const int n,
                                          const T* __restrict__ xs, const int lane_in, const int step0, const int step1) {
    if constexpr (kSuper16) return pass16<WANT_H>(prob, lay, n, xs, lane_in, step0, step1);   // (step0 a multiple of 16 there)
    // Opaque copy of the lane id: everything below that depends only on the lane and the layout is invariant across chunks,
    // passes and problems, and LICM would hoist it out of all three loops and pin it across the LDL^T and the state machine.
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int k = lane >> 4, c = lane & 15;
    const int RS = lay.rs, rsm = lay.rsm;
    const bool active = c * NBM < rsm;
    PassCtx pc;
    pc.c = c;
#pragma unroll
    for (int cb = 0; cb < NBM; ++cb) {
      const int q = NBM * c + cb;
      pc.xr[cb] = (q < lay.nmr) ? xs[q] : T(0);
    }
#pragma unroll
    for (int j = 0; j + 1 < THIN; ++j) pc.xt[j] = (c == 0) ? xs[lay.nmr + j] : T(0);
    pc.q0 = (lane & 1) != 0;
    pc.q1 = (lane & 2) != 0;
    pc.isB_lane = (THIN == 0) && ((c + 1) * NBM == rsm);
    pc.mA = pc.isB_lane ? T(0) : T(1);
    pc.mB = pc.isB_lane ? T(1) : T(0);
    pc.loss = 0;
    pc.th2 = T(0);
    pc.k = k;
    pc.rows_real = 0;
    pc.owner = THIN == 0 ? pc.isB_lane : c == 0;
    pc.inl = T(0);
    if (WANT_H) clear();
    T csum = 0;
    const int steps = step1 - step0;
    const unsigned step_bytes = unsigned(4 * RS) * unsigned(sizeof(T));
    const unsigned step_bytes_u = unsigned(__builtin_amdgcn_readfirstlane(int(step_bytes)));
    const i32x4 rsrc = make_rsrc(prob, unsigned(step1) * step_bytes);   // ends with the chunk
    const unsigned voff = active ? unsigned((k * RS + c * NBM) * int(sizeof(T))) : 0x80000000u;
    const unsigned vofft = unsigned((k * RS + rsm) * int(sizeof(T)));
    const unsigned base = unsigned(__builtin_amdgcn_readfirstlane(int(unsigned(step0) * step_bytes)));
    Slots S[kDepth];
    SlotsT St[kDepth];
    static_for<kDepth - 1>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      issue_batch(S[i], St[i], rsrc, voff, vofft, base + unsigned(i * U) * step_bytes_u, step_bytes_u);
    });
    wait_slots(S[0], St[0]);
    for (int s0 = 0; s0 < steps; s0 += kDepth * U) {
      static_for<kDepth>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        constexpr int refill = (i + kDepth - 1) % kDepth;
        const unsigned soff = unsigned(__builtin_amdgcn_readfirstlane(int(base + unsigned(s0 + (i + kDepth - 1) * U) * step_bytes_u)));
        issue_batch(S[refill], St[refill], rsrc, voff, vofft, soff, step_bytes_u);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (i == kDepth - 1)
          compute_batch<WANT_H, true, false>(S[i], St[i], pc, csum, __builtin_amdgcn_readfirstlane(int(s0 + kDepth * U >= steps)), 0);
        else
          compute_batch<WANT_H, false, false>(S[i], St[i], pc, csum, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        wait_slots(S[(i + 1) % kDepth], St[(i + 1) % kDepth]);
      });
    }
    if constexpr (kDepth > 2) {
      static_for<kDepth>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        wait_batch<0, kDw>(S[i][0], S[i][1], S[i][2], S[i][3]);
        if (THIN) wait_batch<0, kDwT>(St[i][0], St[i][1], St[i][2], St[i][3]);
      });
    }
    if (WANT_H) {
      mfma_retire();
      return T(0);
    }
    return wave_allreduce_sum(csum);
  }
  // the thin products' fold over the 4 row groups (what pass() does at its end), after the chunk partials were summed
  __device__ __forceinline__ void fold_thin() {
    if (THIN) {
#pragma unroll
      for (int t = 0; t < NTM; ++t) accT[t] = kgroup_allreduce_sum(accT[t]);
#pragma unroll
      for (int t = 0; t < NTT; ++t) accTT[t] = kgroup_allreduce_sum(accTT[t]);
    }
  }

  // ---- fp64, n <= 15: SIXTEEN steps (64 rows) per sin / cos ---------------------------------------------------------------
  // The quad-transposed reduction above leaves each a_i.x on FOUR lanes (16 row totals per batch on 64 lanes), so the
  // polynomial sin / cos — 20 half-rate fp64 operations + ~20 selects / integer operations, a third of the fp64 pass's issue
  // cycles — is evaluated four times per row.  Here four batches are reduced together: a 16-way transposed reduction
  // (butterfly over the lane bits 0..3 of a row group: quad swaps, then row rotations by 4 / 12 and by 8) leaves the total
  // of step s on lane c == s of its row group — 64 distinct rows on 64 lanes, ONE sin / cos per 64 rows — and each step
  // fetches its (scale, residual base) back with a row broadcast (DPP row_newbcast).  Costs 32 load registers instead of 16
  // (NBM = 1); the loads of the next super-batch are issued batch by batch into the registers the MFMAs have just freed.
#ifndef TOA_NO_SUPER16
  // NBM == 1, THIN == 0 only (n <= 15: the shapes of BASELINE's fp64 configs C2 / C3).  Measured on one MI355X, same-box
  // interleaved: C3 (n = 12, m = 500, 10 000 problems) 0.653 -> 0.586 ms per launch although the kernel goes from 120 to
  // 156 registers (4 -> 3 waves / SIMD); NBM == 2 (n = 24, 31): 200 registers, 2 waves / SIMD, 1-2 % SLOWER — off there;
  // with a thin tail the four batches' thin registers and products take the kernel past 300 registers.
  static constexpr bool kSuper16 = sizeof(T) == 8 && NBM == 1 && THIN == 0;
#else
  static constexpr bool kSuper16 = false;
#endif
  template <int CTRL, int BANK_MASK, bool BOUND>
  static __device__ __forceinline__ T dpp_move(const T old, const T v) {
    static_assert(sizeof(T) == 8, "fp64 helper");
    const long long b = __double_as_longlong(v), o = __double_as_longlong(old);
    const int lo = __builtin_amdgcn_update_dpp(int(o & 0xffffffffll), int(b & 0xffffffffll), CTRL, 0xf, BANK_MASK, BOUND);
    const int hi = __builtin_amdgcn_update_dpp(int(o >> 32), int(b >> 32), CTRL, 0xf, BANK_MASK, BOUND);
    return __longlong_as_double((long long)(unsigned)lo | ((long long)hi << 32));
  }
  // value of lane c ^ 4 of the same 16-lane row: rotation by 12 (source c - 4) everywhere, then rotation by 4 (source c + 4)
  // written over the banks whose lanes have bit 2 clear (banks 0 and 2 = lanes 0-3, 8-11)
  static __device__ __forceinline__ T row_xor4(const T v) {
    const T a = dpp_move<0x120 + 12, 0xf, true>(T(0), v);
    return dpp_move<0x120 + 4, 0x5, false>(a, v);
  }
  template <int S>
  static __device__ __forceinline__ T row_bcast16(const T v) { return dpp_move<0x150 + S, 0xf, true>(T(0), v); }  // row_newbcast:S
  static __device__ __forceinline__ T reduce16(const T (&p)[16], const int lane) {
    const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0, b2 = (lane & 4) != 0, b3 = (lane & 8) != 0;
    T r1[8], r2[4], r3[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const T keep = b0 ? p[2 * i + 1] : p[2 * i], send = b0 ? p[2 * i] : p[2 * i + 1];
      r1[i] = keep + dpp_quad<kQuadSwap1>(send);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const T keep = b1 ? r1[2 * i + 1] : r1[2 * i], send = b1 ? r1[2 * i] : r1[2 * i + 1];
      r2[i] = keep + dpp_quad<kQuadSwap2>(send);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const T keep = b2 ? r2[2 * i + 1] : r2[2 * i], send = b2 ? r2[2 * i] : r2[2 * i + 1];
      r3[i] = keep + row_xor4(send);
    }
    const T keep = b3 ? r3[1] : r3[0], send = b3 ? r3[0] : r3[1];
    return keep + dpp_row_ror<8>(send);
  }

  // step0 / step1: the steps [step0, step1) only (a chunk of the cooperative tail, see pass_chunk: step0 a multiple of the
  // 16-step super-batch, the buffer descriptor ends with the chunk); step1 < 0: the whole problem.
  template <bool WANT_H>
  __device__ __forceinline__ T pass16(const T* __restrict__ prob, const DenseRowLayout& lay, const int n, const T* __restrict__ xs,
                                      const int lane_in, const int step0 = 0, const int step1 = -1) {
    int lane = lane_in;
    if (step1 >= 0) asm volatile("" : "+v"(lane));   // (chunk form: keep the lane-only values out of the caller's loops, as pass_chunk does)
    const int k = lane >> 4, c = lane & 15;
    const int RS = lay.rs, rsm = lay.rsm;
    const bool active = c * NBM < rsm;
    PassCtx pc;
    pc.c = c;
#pragma unroll
    for (int cb = 0; cb < NBM; ++cb) {
      const int q = NBM * c + cb;
      pc.xr[cb] = (q < lay.nmr) ? xs[q] : T(0);
    }
#pragma unroll
    for (int j = 0; j + 1 < THIN; ++j) pc.xt[j] = (c == 0) ? xs[lay.nmr + j] : T(0);
    pc.q0 = (lane & 1) != 0;
    pc.q1 = (lane & 2) != 0;
    pc.isB_lane = (THIN == 0) && ((c + 1) * NBM == rsm);
    pc.mA = pc.isB_lane ? T(0) : T(1);
    pc.mB = pc.isB_lane ? T(1) : T(0);
    pc.loss = 0;
    pc.th2 = T(0);
    pc.k = k;
    pc.rows_real = 0;
    pc.owner = THIN == 0 ? pc.isB_lane : c == 0;
    pc.inl = T(0);
    if (WANT_H) clear();
    T csum = 0;
    const int last_step = step1 >= 0 ? step1 : (lay.m4 >> 2);
    const int steps = last_step - step0;
    const unsigned step_bytes_u = unsigned(__builtin_amdgcn_readfirstlane(int(unsigned(4 * RS) * unsigned(sizeof(T)))));
    const i32x4 rsrc = make_rsrc(prob, unsigned(last_step) * step_bytes_u);
    const unsigned base = unsigned(__builtin_amdgcn_readfirstlane(int(unsigned(step0) * step_bytes_u)));
    const unsigned voff = active ? unsigned((k * RS + c * NBM) * int(sizeof(T))) : 0x80000000u;
    const unsigned vofft = unsigned((k * RS + rsm) * int(sizeof(T)));
    constexpr int NBATCH = 4;   // batches of U steps per super-batch
    Slots S[NBATCH];
    SlotsT St[NBATCH];
    T wa[NBATCH][U][NBM];
    T va[NBATCH][U][THIN ? THIN : 1];
    T part[NBATCH * U];
    // wait for batch j (the younger batches of the super-batch may stay outstanding) and form its partial dot products
    auto land = [&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      constexpr int leave = (NBATCH - 1 - j) * kLoadsPerBatch;
      wait_batch<leave, kDw>(S[j][0], S[j][1], S[j][2], S[j][3]);
      if (THIN) wait_batch<leave, kDwT>(St[j][0], St[j][1], St[j][2], St[j][3]);
      T pj[U];
      batch_dots(S[j], St[j], pc, wa[j], va[j], pj);
#pragma unroll
      for (int u = 0; u < U; ++u) part[j * U + u] = pj[u];
    };
    static_for<NBATCH>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      issue_batch(S[j], St[j], rsrc, voff, vofft, base + unsigned(j * U) * step_bytes_u, step_bytes_u);
    });
    static_for<NBATCH>(land);
    // Loop invariant at the back-edge: NO load is in flight (every batch of the next super-batch has landed and been folded
    // into part[] at the bottom of the body), so the register copies hipcc places on the back-edge only ever touch data that
    // has arrived (tools/isa_lint.py checks exactly that).
    for (int s0 = 0; s0 < steps; s0 += NBATCH * U) {
      const T t16 = reduce16(part, lane);
      T sn, cs;
      sincos_t(t16, &sn, &cs);
      const T sc16 = T(1) + T(0.1) * cs;
      const T rb16 = t16 + T(0.1) * sn;
      const int last = __builtin_amdgcn_readfirstlane(int(s0 + NBATCH * U >= steps));
      static_for<NBATCH>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        static_for<U>([&](auto uc) __attribute__((always_inline)) {
          constexpr int u = decltype(uc)::value;
          apply_step<WANT_H, (j == NBATCH - 1 && u == U - 1), false>(wa[j][u], va[j][u], row_bcast16<j * U + u>(sc16),
                                                                       row_bcast16<j * U + u>(rb16), pc, csum, last, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
        // refill the registers this batch's MFMAs have just consumed with the same batch of the NEXT super-batch
        const unsigned soff = unsigned(__builtin_amdgcn_readfirstlane(int(base + unsigned(s0 + NBATCH * U + j * U) * step_bytes_u)));
        issue_batch(S[j], St[j], rsrc, voff, vofft, soff, step_bytes_u);
        __builtin_amdgcn_sched_barrier(0);
      });
      static_for<NBATCH>(land);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (WANT_H) {
      mfma_retire();
      if (THIN) {
#pragma unroll
        for (int t = 0; t < NTM; ++t) accT[t] = kgroup_allreduce_sum(accT[t]);
#pragma unroll
        for (int t = 0; t < NTT; ++t) accTT[t] = kgroup_allreduce_sum(accTT[t]);
      }
      return T(0);
    }
    return wave_allreduce_sum(csum);
  }

  // ---- fp64, n <= 15, ROW PER LANE through LDS (round 3; lm_fused_kernel only: DenseRowModel<.., COOP = true> of these
  // layouts) ---------------------------------------------------------------------------------------------------------------
  // PMC of the pass above at C3 (profiles/r03_ab_log.md): 26 VALU instructions per 4-row step against ONE 64-cycle MFMA —
  // the cross-lane reductions of a_i.x (a lane holds one element of a row) and the broadcasts back dominate the issue
  // slots, and VALU work does not overlap the matrix pipe.  Here a super-batch of 64 rows goes
  //   HBM -> registers (coalesced, element l + 64 i on lane l) -> LDS, linear
  //   LDS -> lane r reads ROW r (RS doubles; conflict-free for odd RS): a_r.x is RS - 1 in-lane FMAs, ONE sin / cos per
  //          row, J_r = s a_r in-lane — no cross-lane instruction at all
  //   [J_r | r_r | 0] -> LDS, 128-byte rows, 16-byte granules XOR-swizzled by the row (the writer is a row, the reader a column)
  //   LDS -> MFMA operand of step q straight from ds_read_b64 (lane (i, k): column i of row 4 q + k; the step enters the
  //          address as an immediate)
  // ~5 VALU instructions per step instead of 26.  The 8 KB stage of the wave is one region used three times per super-batch
  // (raw rows, then the scaled rows over them): the LDS operations of a wave execute in order.
  // The summation order of a_i.x changes (in-lane chain instead of a lane tree): results differ from pass16's in the last
  // bits; every comparison with the oracle is by tolerance, and the launch-per-iteration forms keep pass16.
  static constexpr int kStageBytes = 64 * 128;
  // stage image: row r at byte r 128, its 16-byte granule g (columns 2 g, 2 g + 1) at position g ^ (r & 7) — for the raw rows
  // and for the scaled rows alike, so [J | r] simply overwrites the row it was made from.  Columns >= RS are zero (the loads
  // of lanes beyond the row's end return 0), which makes the whole pass independent of n: 16 columns, no branches.
  // Round 5, measured and NOT the default (-DTOA_S16_TWO_SETS; profiles/r05_ab_log.md section 3): TWO sets of load registers, the loads
  // of super-batch k + 2 issued while k is computed.  The idea: what bounds a launch of 10 000 problems is its tail — a problem
  // started into an emptying chip takes as long as one started into a full one — and a lone wave with two round trips in flight
  // should be faster.  It is not by much (3 072 problems, one round: 0.258 -> 0.249 ms), the second set costs the third wave per
  // SIMD (192 registers), which the steady state does use (40 960 problems: 1.689 -> 1.792 ms), and at the BASELINE's 10 000 the
  // two cancel (0.541 -> 0.547 ms).  A problem's latency at low occupancy is not its load latency.
#ifdef TOA_S16_TWO_SETS
#define TOA_S16_SETS 2
#else
#define TOA_S16_SETS 1
#endif
  template <bool WANT_H>
  __device__ __forceinline__ void s16_compute(unsigned char* __restrict__ stage, const int myrow, const int myb, const int wr0, const int wr1,
                                              const T (&xu)[16], T& csum) {
    typedef T T2 __attribute__((ext_vector_type(2)));
    T a[16];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const T2 v = *reinterpret_cast<const T2*>(stage + (myrow ^ (g << 4)));
      a[2 * g] = v[0];
      a[2 * g + 1] = v[1];
    }
    const T bi = *reinterpret_cast<const T*>(stage + myb);
    T t = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) t = fma(a[j], xu[j], t);
    T sn, cs;
    sincos_t_s(t, &sn, &cs);
    const T sc = T(1) + T(0.1) * cs;
    const T res = (t + T(0.1) * sn) - bi;
    if constexpr (!WANT_H) {
      csum = fma(res, res, csum);
    } else {
      // ---- [J | r | 0] over the row it came from
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        *reinterpret_cast<T2*>(stage + (myrow ^ (g << 4))) = T2{sc * a[2 * g], sc * a[2 * g + 1]};
      }
      *reinterpret_cast<T*>(stage + myb) = res;   // (after the granule that holds column n: same lane, in order)
      __builtin_amdgcn_wave_barrier();
      // ---- sixteen steps on the matrix core
      T op[16];                          // every operand read is issued before the first MFMA waits for its own
#pragma unroll
      for (int q = 0; q < 16; ++q) op[q] = *reinterpret_cast<const T*>(stage + ((q & 1) ? wr1 : wr0) + q * 512);
#pragma unroll
      for (int q = 0; q < 16; ++q) asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %1, %0" : TOA_ACC(acc[0]) : "v"(op[q]));
#if TOA_S16_SETS == 2
      // two sites of this block in one kernel (one per register set): hipcc moves the accumulator between them, right behind the
      // last MFMA, where it cannot see the XDL-write -> read hazard of an asm MFMA.  The matrix pipe is waited out inside the
      // statement instead (18 wait states of a 16-pass DGEMM MFMA: 20 cycles per super-batch of 16 x 64).
      asm volatile("s_nop 15\n\ts_nop 3" : TOA_ACC(acc[0]));
#endif
      __builtin_amdgcn_wave_barrier();
    }
  }
  template <bool WANT_H>
  __device__ __forceinline__ T pass16s(const T* __restrict__ prob, const DenseRowLayout& lay, const int n, const T* __restrict__ xs,
                                       const int lane, unsigned char* __restrict__ stage) {
    static_assert(sizeof(T) == 8 && NBM == 1 && THIN == 0, "fp64, n <= 15");
    const int RS = lay.rs;                      // n + 1 <= 16: the row [a_i | b_i]
    const int k = lane >> 4, c = lane & 15;
    const int steps = lay.m4 >> 2;
    const unsigned step_bytes_u = unsigned(__builtin_amdgcn_readfirstlane(int(unsigned(4 * RS) * 8u)));
    const i32x4 rsrc = make_rsrc(prob, unsigned(steps) * step_bytes_u);
    const unsigned voff = c < RS ? unsigned((k * RS + c) * 8) : 0x80000000u;
    if (WANT_H) clear();
    T csum = 0;
    // loop-invariant LDS addresses (bytes from the wave's stage)
    const int wr0 = k * 128 + (((c >> 1) ^ k) << 4) + (c & 1) * 8;   // element (row 4 i + k, column c): wr0 + i 512 for even i, (wr0 ^ 64) + i 512 for odd
    const int wr1 = wr0 ^ 64;                                        // (also the MFMA operand of step i: lane (c, k) reads column c of row 4 i + k)
    const int myrow = lane * 128 + ((lane & 7) << 4);                // granule g of row `lane`: myrow ^ (g << 4)
    const int myb = lane * 128 + ((((n >> 1) ^ (lane & 7))) << 4) + (n & 1) * 8;   // column n of row `lane`: b_i, later r_i
    // x is wave-uniform: sixteen SGPR pairs (x_j = 0 for j >= n: b and the padding add nothing to a_i.x).  As vector registers
    // they would cost the kernel its third wave per SIMD.
    T xu[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const long long b = __double_as_longlong(xs[j]);
      const unsigned lo = unsigned(__builtin_amdgcn_readfirstlane(int(unsigned(b)))), hi = unsigned(__builtin_amdgcn_readfirstlane(int(unsigned(b >> 32))));
      xu[j] = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
#if TOA_S16_SETS == 1
    RawVec<2> st[16];
    s16_issue<0>(st, rsrc, voff, 0u, step_bytes_u);
    for (int s0 = 0; s0 < steps; s0 += 16) {
      // ---- raw rows -> LDS, then the next super-batch's loads into the same registers
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      s16_to_lds<0>(st, stage + wr0, stage + wr1);
      __builtin_amdgcn_wave_barrier();
      s16_issue<0>(st, rsrc, voff, unsigned(__builtin_amdgcn_readfirstlane(int(unsigned(s0 + 16) * step_bytes_u))), step_bytes_u);
      s16_compute<WANT_H>(stage, myrow, myb, wr0, wr1, xu, csum);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the prefetch past the end (zeros) before its registers die
    s16_touch<0>(st);
#else
    // super-batches alternate between the two register sets; the loads of k + 2 go out as soon as k has been copied to LDS, so two
    // super-batches are in flight while one is computed.  vmcnt(16): the sixteen loads of the YOUNGER set may stay outstanding.
    RawVec<2> sa[16], sb[16];
    s16_issue<0>(sa, rsrc, voff, 0u, step_bytes_u);
    s16_issue<0>(sb, rsrc, voff, unsigned(__builtin_amdgcn_readfirstlane(int(16u * step_bytes_u))), step_bytes_u);
    for (int s0 = 0; s0 < steps; s0 += 32) {
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      s16_to_lds<0>(sa, stage + wr0, stage + wr1);
      __builtin_amdgcn_wave_barrier();
      s16_issue<0>(sa, rsrc, voff, unsigned(__builtin_amdgcn_readfirstlane(int(unsigned(s0 + 32) * step_bytes_u))), step_bytes_u);
      s16_compute<WANT_H>(stage, myrow, myb, wr0, wr1, xu, csum);
      // the second set: wait / copy / re-issue UNCONDITIONALLY — the load pipeline has one shape on every path (tools/isa_lint.py
      // follows both sides of every branch) — and only the arithmetic is skipped when an odd super-batch count leaves it no rows
      // (its loads came back as zeros without touching memory: past the descriptor's end)
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      s16_to_lds<0>(sb, stage + wr0, stage + wr1);
      __builtin_amdgcn_wave_barrier();
      s16_issue<0>(sb, rsrc, voff, unsigned(__builtin_amdgcn_readfirstlane(int(unsigned(s0 + 48) * step_bytes_u))), step_bytes_u);
      if (s0 + 16 < steps) s16_compute<WANT_H>(stage, myrow, myb, wr0, wr1, xu, csum);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the prefetches past the end (zeros) before their registers die
    s16_touch<0>(sa);
    s16_touch<0>(sb);
#endif
    if (WANT_H) {
      mfma_retire();
      return T(0);
    }
    return wave_allreduce_sum(csum);
  }
  // (the asm operands below cannot be named from inside a lambda: plain recursive templates)
  template <int I>
  static __device__ __forceinline__ void s16_issue(RawVec<2> (&st)[16], const i32x4 rsrc, const unsigned voff, const unsigned soff,
                                                   const unsigned step_bytes) {
    if constexpr (I < 16) {
      st[I].template issue<I == 0, 0>(rsrc, voff, soff);
      s16_issue<I + 1>(st, rsrc, voff, soff + step_bytes, step_bytes);
    }
  }
  template <int I>
  static __device__ __forceinline__ void s16_to_lds(RawVec<2> (&st)[16], unsigned char* even, unsigned char* odd) {
    if constexpr (I < 16) {
      asm volatile("" : "+v"(st[I].a));
      *reinterpret_cast<u32x2*>(((I & 1) ? odd : even) + I * 512) = st[I].a;
      s16_to_lds<I + 1>(st, even, odd);
    }
  }
  template <int I>
  static __device__ __forceinline__ void s16_touch(RawVec<2> (&st)[16]) {
    if constexpr (I < 16) {
      asm volatile("" : "+v"(st[I].a));
      s16_touch<I + 1>(st);
    }
  }

  // The same pass over rows in the NATURAL layout (large_fused.hip, 64 <= n <= 128): `A` = nrows rows of n elements,
  // `bv` = their nrows right-hand sides (THIN == 1: the thin tail is b alone, fetched through its own descriptor).  A lane
  // whose NBM columns straddle the end of a row reads the head of the next one: those columns q >= n meet x = 0 in a_i.x,
  // and the Gram rows / columns they pollute are never read back (extract_g_diag_cost / the caller's fold stop at n).
  // T0, T1, THINP: this pass forms the tiles T0 <= t < T1 only, and the thin products (g, cost) only if THINP.
  // ROBUST: every residual goes through the M-estimator (loss, th2); returns the sum of the losses (also for WANT_H) and
  // the number of inlier rows through *ninl.
  template <bool WANT_H, int D = 3, int T0 = 0, int T1 = NT, bool THINP = true, bool ROBUST = false>   // D: slot sets in the ring
  __device__ __forceinline__ T pass_natural(const T* __restrict__ A, const T* __restrict__ bv, const int n, const int nrows,
                                            const T* __restrict__ xs, const int lane, const int loss = 0, const T th2 = T(0),
                                            int* ninl = nullptr) {
    static_assert(THIN == 1 && D >= 2 && D <= 4, "natural layout: b is the whole thin tail");
    const int k = lane >> 4, c = lane & 15;
    PassCtx pc;
    pc.c = c;
#pragma unroll
    for (int cb = 0; cb < NBM; ++cb) {
      const int q = NBM * c + cb;
      pc.xr[cb] = (q < n) ? xs[q] : T(0);
    }
    pc.xt[0] = T(0);
    pc.q0 = (lane & 1) != 0;
    pc.q1 = (lane & 2) != 0;
    pc.isB_lane = false;
    pc.mA = T(1);
    pc.mB = T(0);
    pc.loss = loss;
    pc.th2 = th2;
    pc.k = k;
    pc.rows_real = nrows;
    pc.owner = c == 0;
    pc.inl = T(0);
    if (WANT_H) {
#pragma unroll
      for (int t = T0; t < T1; ++t) acc[t] = Acc{0, 0, 0, 0};
      if (THINP) {
#pragma unroll
        for (int t = 0; t < NTM; ++t) accT[t] = T(0);
#pragma unroll
        for (int t = 0; t < (NTT ? NTT : 1); ++t) accTT[t] = T(0);
      }
    }
    T csum = 0;
    const int steps = (nrows + 3) >> 2;
    const i32x4 rsA = make_rsrc(A, unsigned(nrows) * unsigned(n) * unsigned(sizeof(T)));
    const i32x4 rsB = make_rsrc(bv, unsigned(nrows) * unsigned(sizeof(T)));
    const unsigned voff = (NBM * c < n) ? unsigned((k * n + NBM * c) * int(sizeof(T))) : 0x80000000u;
    const unsigned voffb = unsigned(k * int(sizeof(T)));
    const unsigned stepA = unsigned(__builtin_amdgcn_readfirstlane(int(4u * unsigned(n) * unsigned(sizeof(T)))));
    constexpr unsigned stepB = 4u * unsigned(sizeof(T));
    auto issue_nat = [&](Slots& m, SlotsT& t, const int step0) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const unsigned sa = unsigned(__builtin_amdgcn_readfirstlane(int(unsigned(step0 + u) * stepA)));
        const unsigned sb = unsigned(__builtin_amdgcn_readfirstlane(int(unsigned(step0 + u) * stepB)));
        m[u].issue(rsA, voff, sa);
        t[u].issue(rsB, voffb, sb);
      }
    };
    constexpr int kLeaveN = (D - 2) * kLoadsPerBatch;  // younger batches that may stay outstanding at a wait
    static_assert(kLeaveN < 64, "vmcnt is a 6-bit counter");
    auto wait_nat = [&](Slots& m, SlotsT& t) __attribute__((always_inline)) {
      wait_batch<kLeaveN, kDw>(m[0], m[1], m[2], m[3]);
      wait_batch<kLeaveN, kDwT>(t[0], t[1], t[2], t[3]);
    };
    Slots S[D];
    SlotsT St[D];
    static_for<D - 1>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      issue_nat(S[i], St[i], i * U);
    });
    wait_nat(S[0], St[0]);
    for (int s0 = 0; s0 < steps; s0 += D * U) {
      static_for<D>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        constexpr int refill = (i + D - 1) % D;  // consumed in the previous segment
        issue_nat(S[refill], St[refill], s0 + (i + D - 1) * U);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (i == D - 1)
          compute_batch<WANT_H, true, ROBUST, T0, T1, THINP>(S[i], St[i], pc, csum, __builtin_amdgcn_readfirstlane(int(s0 + D * U >= steps)), 4 * (s0 + i * U));
        else
          compute_batch<WANT_H, false, ROBUST, T0, T1, THINP>(S[i], St[i], pc, csum, 0, 4 * (s0 + i * U));
        __builtin_amdgcn_sched_barrier(0);
        wait_nat(S[(i + 1) % D], St[(i + 1) % D]);
      });
    }
    if constexpr (D > 2) {  // drain the prefetches past the end before their registers are reused
      static_for<D>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        wait_batch<0, kDw>(S[i][0], S[i][1], S[i][2], S[i][3]);
        wait_batch<0, kDwT>(St[i][0], St[i][1], St[i][2], St[i][3]);
      });
    }
    if constexpr (ROBUST) {
      if (ninl) *ninl = int(wave_allreduce_sum(pc.inl));   // exact in T: one count per row
    }
    if (WANT_H) {
      asm volatile("s_nop 7" ::: "memory");   // mfma_retire() over the tiles of this pass
#pragma unroll
      for (int t = T0; t < T1; ++t) asm volatile("" : TOA_ACC(acc[t]));
      if (THINP) {
#pragma unroll
        for (int t = 0; t < NTM; ++t) accT[t] = kgroup_allreduce_sum(accT[t]);
#pragma unroll
        for (int t = 0; t < NTT; ++t) accTT[t] = kgroup_allreduce_sum(accTT[t]);
      }
      if constexpr (ROBUST) return wave_allreduce_sum(csum);   // sum of l; the Gram's (r, r) entry holds sum of s r^2
      return T(0);
    }
    return wave_allreduce_sum(csum);
  }

  // ---- memo of a finished Gram (lm_device.hpp: lm_memo): the accumulators exactly as the pass left them (after the thin
  // products' row-group fold), element r of lane l at slot[r * 64 + l] — every store / load is one contiguous 256-byte
  // (1 KB for the tiles) line per wave.  save -> load returns the same bits, so everything derived from the registers
  // afterwards (g, diagonal, cost, the LDL^T image) equals what a second pass over the rows would have produced.
  static constexpr int kMemoElems = (NT * 4 + NTM + NTT) * 64;
  // (One tile at a time, with scheduling barriers in between: left alone hipcc issues every load of a fold first — 24 tile +
  //  15 thin registers for the loads, as many for the accumulator copies, as many for the sums — and those ~120 transient
  //  registers on top of the kernel's long-lived ones became the register count of the WHOLE fused kernel.)
  __device__ __forceinline__ void memo_save(T* __restrict__ slot, const int lane_in) const {
    int lane = lane_in;
    asm volatile("" : "+v"(lane));   // keep the per-lane addresses out of LICM's reach (see extract_g_diag_cost)
    Acc* __restrict__ st = reinterpret_cast<Acc*>(slot);
    static_for<NT>([&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value;
      st[t * 64 + lane] = acc[t];
      __builtin_amdgcn_sched_barrier(0);
    });
    T* __restrict__ s2 = slot + NT * 4 * 64;
    if (THIN) {
#pragma unroll
      for (int t = 0; t < NTM; ++t) s2[t * 64 + lane] = accT[t];
#pragma unroll
      for (int t = 0; t < NTT; ++t) s2[(NTM + t) * 64 + lane] = accTT[t];
    }
  }
  // slot += the registers, element by element (the fold of one chunk partial into the running total of a pass)
  __device__ __forceinline__ void memo_add(T* __restrict__ slot, const int lane_in) const {
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    Acc* __restrict__ st = reinterpret_cast<Acc*>(slot);
    static_for<NT>([&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value;
      const Acc o = st[t * 64 + lane];
      st[t * 64 + lane] = Acc{o[0] + acc[t][0], o[1] + acc[t][1], o[2] + acc[t][2], o[3] + acc[t][3]};
      __builtin_amdgcn_sched_barrier(0);
    });
    T* __restrict__ s2 = slot + NT * 4 * 64;
    if (THIN) {
      static_for<NTM>([&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
        s2[t * 64 + lane] = s2[t * 64 + lane] + accT[t];
        if constexpr ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      });
      __builtin_amdgcn_sched_barrier(0);
      static_for<NTT>([&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
        s2[(NTM + t) * 64 + lane] = s2[(NTM + t) * 64 + lane] + accTT[t];
        if constexpr ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      });
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // memo_load for the END of a chunked pass: the same values, but every register is written as a function of its OLD content
  // (old & 0 | loaded, with a zero the compiler cannot see through).  A plain load starts new live ranges for the 24 + 15
  // accumulators next to the ones the chunk loop just used, and hipcc then kept both sets: 156 -> 180 registers for the
  // whole fused kernel (3 -> 2 waves / SIMD at n = 50).
  template <int t>
  __device__ __forceinline__ void lds_read_tile(unsigned a0) {
    if constexpr (sizeof(T) == 4) {
      asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : TOA_ACC(acc[t]) : "v"(a0), "n"(t * 64 * 16) : "memory");
    } else {
      // fp64: a tile is 8 dwords, read as two halves through a cast of the register variable
      asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : TOA_ACC(reinterpret_cast<u32x4*>(&acc[t])[0]) : "v"(a0), "n"(t * 64 * 32) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : TOA_ACC(reinterpret_cast<u32x4*>(&acc[t])[1]) : "v"(a0), "n"(t * 64 * 32 + 16) : "memory");
    }
    if constexpr (t + 1 < NT) lds_read_tile<t + 1>(a0);
  }
  template <int t, int N, int BASE>
  static __device__ __forceinline__ void lds_read_thin(T* dst, unsigned a1) {
    if constexpr (t < N) {
      if constexpr (sizeof(T) == 4) asm volatile("ds_read_b32 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "+v"(dst[t]) : "v"(a1), "n"((BASE + t) * 64 * 4) : "memory");
      else asm volatile("ds_read_b64 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "+v"(dst[t]) : "v"(a1), "n"((BASE + t) * 64 * 8) : "memory");
      lds_read_thin<t + 1, N, BASE>(dst, a1);
    }
  }
  __device__ __forceinline__ void memo_load_inplace(const T* __restrict__ slot, const int lane_in) {
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    // LDS byte address of this lane's first element; the ds_read destinations are IN-OUT operands (tied to the registers the
    // chunk loop used).  Every statement waits for its own data: an LDS read completes asynchronously, hipcc does not know
    // that an asm statement's output is still in flight, and any register copy it places between two statements would read
    // stale data (seen in fp64, whose tiles are read as two halves through a cast: run-to-run differences).  The ~20 serial
    // LDS round trips per pass are hidden by the other waves of the SIMD.
    const unsigned a0 = unsigned(reinterpret_cast<size_t>((__attribute__((address_space(3))) const char*)(slot))) + unsigned(lane) * unsigned(sizeof(Acc));
    lds_read_tile<0>(a0);
    if (THIN) {
      const unsigned a1 = unsigned(reinterpret_cast<size_t>((__attribute__((address_space(3))) const char*)(slot + NT * 4 * 64))) + unsigned(lane) * unsigned(sizeof(T));
      lds_read_thin<0, NTM, 0>(accT, a1);
      lds_read_thin<0, NTT, NTM>(accTT, a1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < NT; ++t) asm volatile("" : TOA_ACC(acc[t]));
    if (THIN) {
#pragma unroll
      for (int t = 0; t < NTM; ++t) asm volatile("" : "+v"(accT[t]));
#pragma unroll
      for (int t = 0; t < NTT; ++t) asm volatile("" : "+v"(accTT[t]));
    }
  }
  __device__ __forceinline__ void memo_load(const T* __restrict__ slot, const int lane_in) {
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const Acc* __restrict__ st = reinterpret_cast<const Acc*>(slot);
    static_for<NT>([&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value;
      acc[t] = st[t * 64 + lane];
      __builtin_amdgcn_sched_barrier(0);
    });
    const T* __restrict__ s2 = slot + NT * 4 * 64;
    if (THIN) {
#pragma unroll
      for (int t = 0; t < NTM; ++t) accT[t] = s2[t * 64 + lane];
#pragma unroll
      for (int t = 0; t < NTT; ++t) accTT[t] = s2[(NTM + t) * 64 + lane];
    }
  }

  // Scatter: g[q] (q<n), undamped diagonal hd[q], and the cost.  Returns the cost (wave-uniform).
  __device__ __forceinline__ T extract_g_diag_cost(T* __restrict__ g, T* __restrict__ hd, const DenseRowLayout& lay,
                                                   const int n, const int lane_in, T* __restrict__ cost_slot) const {
    // Opaque copy of the lane id: the ~40 per-element indices / predicates below are loop-invariant
    // across LM iterations, and LICM would otherwise hoist them out of the problem loop and pin
    // ~100 VGPRs + ~300 SGPRs across the hot accumulate loop.  Recomputing them per call is free.
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int cj = lane & 15;
    const int nmr = lay.nmr;
    const int qB = lay.rsm - 1;  // THIN == 0: index of the r column inside the main Gram
#pragma unroll
    for (int bi = 0; bi < NBM; ++bi)
#pragma unroll
      for (int bj = bi; bj < NBM; ++bj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qi = NBM * Mfma<T>::out_row(lane, r) + bi;
          const int qj = NBM * cj + bj;
          const T v = acc[tile(bi, bj)][r];
          if (THIN == 0) {
            if (qi < nmr && qj == qB) g[qi] = v;
            if (qj < nmr && qi == qB) g[qj] = v;
            if (qi == qB && qj == qB) *cost_slot = v;
          }
          if (qi == qj && qi < nmr) hd[qi] = v;
        }
    if (THIN) {
      if (lane < 16) {
#pragma unroll
        for (int cb = 0; cb < NBM; ++cb) g[NBM * cj + cb] = accT[ti(cb, THIN - 1)];
      }
      if (lane == 0) {
#pragma unroll
        for (int j = 0; j + 1 < THIN; ++j) {
          g[nmr + j] = accTT[tt(j, THIN - 1)];
          hd[nmr + j] = accTT[tt(j, j)];
        }
        *cost_slot = accTT[tt(THIN - 1, THIN - 1)];
      }
    }
    (void)n;
    wave_sync();
    return *cost_slot;
  }

  // Write the full symmetric n×n matrix (undamped) with row stride LD.  Used to build the LDLT
  // workspace and to export H.
  template <typename O>
  __device__ __forceinline__ void write_sym(O* __restrict__ M, const int LD, const DenseRowLayout& lay, const int n,
                                            const int lane_in) const {
    int lane = lane_in;
    asm volatile("" : "+v"(lane));  // see extract_g_diag_cost: keep the index math out of LICM's reach
    const int cj = lane & 15;
    const int nmr = lay.nmr;
#pragma unroll
    for (int bi = 0; bi < NBM; ++bi)
#pragma unroll
      for (int bj = bi; bj < NBM; ++bj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qi = NBM * Mfma<T>::out_row(lane, r) + bi;
          const int qj = NBM * cj + bj;
          if (qi < nmr && qj < nmr) {
            const O v = O(acc[tile(bi, bj)][r]);
            M[qi * LD + qj] = v;
            if (bi != bj) M[qj * LD + qi] = v;
          }
        }
    if (THIN > 1) {
      if (lane < 16) {
#pragma unroll
        for (int cb = 0; cb < NBM; ++cb)
#pragma unroll
          for (int j = 0; j + 1 < THIN; ++j) {
            const int q = NBM * cj + cb;
            const O v = O(accT[ti(cb, j)]);
            M[q * LD + (nmr + j)] = v;
            M[(nmr + j) * LD + q] = v;
          }
      }
      if (lane == 0) {
#pragma unroll
        for (int j = 0; j + 1 < THIN; ++j)
#pragma unroll
          for (int j2 = j; j2 + 1 < THIN; ++j2) {
            const O v = O(accTT[tt(j, j2)]);
            M[(nmr + j) * LD + (nmr + j2)] = v;
            M[(nmr + j2) * LD + (nmr + j)] = v;
          }
      }
    }
    (void)n;
  }
};

}  // namespace toa
