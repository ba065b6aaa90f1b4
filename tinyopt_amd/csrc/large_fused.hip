// The LM / GN loop for 64 <= n <= 128 (TOA_MODEL_DENSE_ROW_NATURAL) as ONE persistent kernel: a workgroup owns a problem
// from its first residual evaluation to its StopReason, nothing returns to the host in between.
//
// Replaces, for this range of n, the launch-per-phase pipeline of large_n.hip (rows kernel writing J, rocBLAS
// gemm_batched J^T J, pre / Cholesky / post kernels, two integers read back per pass).  Same reference code
// (OptimizeAcc optimizer.h:242-327, Step :331-539, SolverLM::Build lm.h:59-120, SolverGN::Solve gn.h:150-171), same
// state machine pieces (LmState, lm_judge_core, lm_good_step / lm_bad_step from lm_device.hpp).
//
//   data pass   the m rows are split into 4 contiguous ranges, one per wavefront; each wave streams its rows ONCE from
//               HBM in matrix-core operand order and accumulates the Gram of [J | r] for them: J^T J as NB (NB + 1) / 2
//               tiles of v_mfma_{f32,f64}_16x16x4 in AGPRs (NB = ceil(n / 16) <= 8: 36 tiles = 144 registers), J^T r and
//               ||r||^2 on the VALU (DenseRowGram<T, NB, 1>::pass_natural).  J is never written, only the lower
//               block triangle is computed.  The four partial Grams are folded in fixed order through an L2-resident
//               scratch block (deterministic: the result does not depend on timing).
//   Build       clip, diagonal check, damping of the diagonal copy in LDS                                    (lm.h:59-120)
//   Solve       blocked LDL^T of the LDS image by the four waves, trailing updates on the matrix cores, substitutions in
//               wave 0 (ldlt_wg.hpp).  Acceptance differs from Eigen's LDLT only for singular positive SEMI-definite
//               matrices, exactly like the library path.
//   Step / loop thread 0 runs lm_judge_core and the loop bookkeeping; x, g, dx, last_dx live in LDS.
//
// HBM traffic per LM iteration: m (n + 1) sizeof(T) — the algorithmic minimum (the library path: 3x that for J alone).
#include <string>

#include "kernels.hpp"
#include "ldlt_wg.hpp"

namespace toa {
namespace {

template <typename T>
struct LfArgs {
  const T* data;  // per problem: A row-major [m][n], then b [m]
  T* x;           // [P][n] in / out
  int n, m;
  long long P;
  toa_options opt;
  toa_results res;
  int* queue;                    // [0] pop counter, [16] workgroups that have left (self-cleaning, as lm_fused_kernel)
  char* scratch;                 // per resident workgroup: 4 partial Grams + the folded H
  size_t scratch_per_wg;
  unsigned long long* counters;  // [4] or null
  int loss;                      // TOA_LOSS_* of the handle (toa_set_loss): the ROBUST instantiation applies it to every residual
  double loss_th2;
  int memo;                      // != 0: the linearisation of the last accepted point is kept (second H slot) and read back after a
                                 // bit-exact roll-back; a failed solve re-entering Build re-uses the one at hand (toa_tuning::memo_off)
};

template <typename T>
__device__ __forceinline__ double wg_sum(const double v, double* red) {  // fixed-order tree: deterministic
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (int(threadIdx.x) < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const double out = red[0];
  __syncthreads();
  return out;
}

// -DTOA_LF_TIMING: workgroup 0 prints where its time went (100 MHz ticks -> us), phase by phase
#ifndef TOA_LF_DEPTH
#define TOA_LF_DEPTH 2   // ring depth of the data pass: 3 measured no faster (4.27 vs 4.25 ms, one MFMA-bound wave per SIMD) and costs 36 registers
#endif
#ifdef TOA_LF_TIMING
#define LF_TICK_START unsigned long long tk_prev = wall_clock64(); const long long ck0 = clock64();
#define LF_TICK(i) { const unsigned long long now_ = wall_clock64(); tk[i] += now_ - tk_prev; tk_prev = now_; }
#else
#define LF_TICK_START
#define LF_TICK(i)
#endif

// ROBUST: every residual through the handle's M-estimator (robust_norms.h:20-26: cost += l, the row's J^T J and J^T r scaled by
// s) — a separate instantiation, so the plain kernel's instruction stream and registers are untouched.
// ---- TS = true (round 3, fp32, rows a multiple of four columns): the TILE-SPLIT data pass ---------------------------------
// The row-split pass above gives every wave the whole Gram (36 tiles = 144 accumulator registers at n = 128; the kernel
// needs 404 registers: one workgroup per CU, one wave per SIMD, every wait exposed, and 65 us of fold + LDL^T + step per
// iteration with the matrix cores idle).  Here the rows of a stage (64 at a time) are prepared ONCE by the workgroup —
// four threads per row: a_i.x, one sin / cos, J_i = s_i a_i — and written to LDS as one [64][16] panel per 16-column block;
// then every wave runs ITS quarter of the tiles (tile t belongs to wave t mod 4) over all 64 rows, operands straight from
// the panels (the step enters the ds_read as an immediate) — 9 accumulator tiles per wave, no fold at all, under 256
// registers: two workgroups share a CU and one's Build / LDL^T / step hides behind the other's Gram.  The two stage buffers
// live in the LDS image of the factorisation, which is only filled after the pass.  J^T r: the wave that owns block b (b mod 4)
// forms it on the VALU from the same operands and the stage's residuals; ||r||^2 by the row workers.
template <int NB>
struct TsTiles {   // tile t = (bi, bj), bi <= bj, row-major over the upper block triangle
  static constexpr int NT = NB * (NB + 1) / 2;
  static constexpr int bi_of(int t) { int bi = 0; while (t >= NB - bi) { t -= NB - bi; ++bi; } return bi; }
  static constexpr int bj_of(int t) { int bi = 0; while (t >= NB - bi) { t -= NB - bi; ++bi; } return bi + t; }
  static constexpr int kSlots = (NT + 3) / 4;
};
template <int NB, int W, int S = 0>
__device__ __forceinline__ void ts_step(float __attribute__((ext_vector_type(4))) (&acc)[TsTiles<NB>::kSlots], const float (&w)[NB]) {
  constexpr int t = W + 4 * S;
  if constexpr (S < TsTiles<NB>::kSlots && t < TsTiles<NB>::NT) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[S]) : "v"(w[TsTiles<NB>::bi_of(t)]), "v"(w[TsTiles<NB>::bj_of(t)]));
    ts_step<NB, W, S + 1>(acc, w);
  }
}



template <int I>
__device__ __forceinline__ void ts_issue(RawVec<4> (&pre)[8], const i32x4 rs, const unsigned voff, const unsigned soff) {
  if constexpr (I < 8) {
    pre[I].template issue<I == 0, I * 64>(rs, voff, soff);
    ts_issue<I + 1>(pre, rs, voff, soff);
  }
}
template <int I>
__device__ __forceinline__ void ts_touch(RawVec<4> (&pre)[8]) {
  if constexpr (I < 8) {
    asm volatile("" : "+v"(pre[I].a));
    ts_touch<I + 1>(pre);
  }
}
// The tile-split data pass as a function of its own (NOT inlined): inside the kernel body hipcc's allocation of the whole
// persistent loop (state machine, blocked LDL^T, cost-only pass) pushed these accumulators around — copies right behind an
// MFMA, which tools/isa_lint.py rejects — and past 256 registers.  As a callee it has its own allocation.  Leaves: H (full,
// symmetric, undamped) in Hs, the raw J^T r in gout (LDS), this wave's part of ||r||^2 in costw[wave].
// WANT_H = false: the cost-only pass — the SAME row arithmetic (so that the cost of a point is the same number whether it
// comes from an accumulate or an evaluate-only pass: with pass_natural's sums for the one and these for the other, last-bit
// differences at the noise floor cost 1.2 extra iterations per problem), no stage, no barrier, no matrix-core work.
template <int NB, bool WANT_H>
__device__ __noinline__ void ts_data_pass(const float* __restrict__ A_in, const float* __restrict__ bv_in, const int n_in, const int m_in,
                                          const float* xs, float* __restrict__ Hs_in, float* gout, float* costw, float* hd_out) {
  typedef float T;
  // the arguments of a non-inlined function arrive in VECTOR registers: make the uniform ones scalar again (a buffer
  // descriptor built from a "divergent" pointer costs a waterfall loop around every load)
  auto uni = [](const void* q) __attribute__((always_inline)) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(q);
    const unsigned lo = unsigned(__builtin_amdgcn_readfirstlane(int(unsigned(b)))), hi = unsigned(__builtin_amdgcn_readfirstlane(int(unsigned(b >> 32))));
    return reinterpret_cast<void*>((static_cast<unsigned long long>(hi) << 32) | lo);
  };
  const float* A = static_cast<const float*>(uni(A_in));
  const float* bv = static_cast<const float*>(uni(bv_in));
  float* Hs = static_cast<float*>(uni(Hs_in));
  const int n = __builtin_amdgcn_readfirstlane(n_in), m = __builtin_amdgcn_readfirstlane(m_in);
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef float Acc __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  constexpr int NT = TsTiles<NB>::NT, kTsSlots = TsTiles<NB>::kSlots;
  constexpr int R = 64, PANEL = R * 64, STAGE = NB * PANEL + R * 4;   // bytes: NB panels of [64][16] floats + the stage's residuals
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rrow = tid >> 2, rj = tid & 3, n4 = n >> 2;
  const int k = lane >> 4, ci = lane & 15;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  static_assert(NB == 8, "the rotation scheme below is for eight column blocks");
  // Tile ownership by ROTATION: wave W works in virtual blocks v = physical block (v + 2 W) mod 8 and every wave runs the same
  // nine virtual tiles — (0,0) (1,1) (0,1) (1,2) (0,2) (1,3) (0,3) (1,4) and (x, x + 4) — whose four rotations are exactly the
  // 36 unordered block pairs (the pairs at distance 4 have only two distinct rotations: waves 0, 1 take x = 0, waves 2, 3 x = 1).
  constexpr int kVa[9] = {0, 1, 0, 1, 0, 1, 0, 1, -1}, kVb[9] = {0, 1, 1, 2, 2, 3, 3, 4, -1};
  const int xsp = wave_u < 2 ? 0 : 1;
  int voff[5];
#pragma unroll
  for (int v = 0; v < 5; ++v) voff[v] = ((v + 2 * wave_u) & 7) * PANEL;
  const int soffa = ((xsp + 2 * wave_u) & 7) * PANEL, soffb = ((xsp + 4 + 2 * wave_u) & 7) * PANEL;
  Acc ts_acc[kTsSlots];
  T ts_g[2] = {T(0), T(0)};
  typedef float f2 __attribute__((ext_vector_type(2)));
  // rows through bounds-checked buffer loads (a row past the end returns zeros: no branches), 16 bytes per lane, the eight
  // loads of a thread as immediates off ONE offset; issued by hand (dense_row.hpp: hipcc does not count asm loads)
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, int(unsigned(m) * unsigned(n) * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bv), 0, int(unsigned(m) * 4u), 0x00020000);
  u4 pre[8];
  T pre_b = 0;
  const unsigned row_bytes = unsigned(n) * 4u;
  const unsigned vofs0 = unsigned(rrow) * row_bytes + unsigned(rj) * 16u;
  auto fetch = [&](const int r0) __attribute__((always_inline)) {
    const int soff = __builtin_amdgcn_readfirstlane(int(unsigned(r0) * row_bytes));
#pragma unroll
    for (int i = 0; i < 8; ++i) pre[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, int(vofs0) + i * 64, soff, 0);
    pre_b = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsB, (r0 + rrow) * 4, 0, 0));   // (the builtin returns the raw dword)
  };
  T csum = 0;
  const char* xl = lds_raw + 2 * STAGE;            // x, copied here by the caller (an LDS address this function can name)
  auto rowwork = [&](char* st) __attribute__((always_inline)) {
    f2 t2 = {0, 0};
    f4 av[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_memcpy(&av[i], &pre[i], 16);
      if (i == 7 && 4 * (rj + 28) >= n) av[i] = f4{0, 0, 0, 0};   // (n < 128, n >= 116: only the last group can run into the next row)
      const f4 xv = *reinterpret_cast<const f4*>(xl + (rj + 4 * i) * 16);
      t2 = f2{av[i][0], av[i][1]} * f2{xv[0], xv[1]} + t2;    // v_pk_fma_f32
      t2 = f2{av[i][2], av[i][3]} * f2{xv[2], xv[3]} + t2;
    }
    T t = t2[0] + t2[1];
    t += __shfl_xor(t, 1);
    t += __shfl_xor(t, 2);
    T sn, cs;
    sincos_t(t, &sn, &cs);
    const T sc = T(1) + T(0.1f) * cs;
    const T resv = (t + T(0.1f) * sn) - pre_b;   // (a row past the end: a = 0, b = 0 -> 0)
    if constexpr (WANT_H) {
      char* wp = st + rrow * 64 + rj * 16;       // float4 idx = rj + 4 i -> block i, position rj of the row's 64 bytes
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<f4*>(wp + i * PANEL) = av[i] * sc;
    }
    if (rj == 0) {
      if constexpr (WANT_H) reinterpret_cast<T*>(st + NB * PANEL)[rrow] = resv;
      csum = fmaf(resv, resv, csum);
    }
  };
#pragma unroll
  for (int sl = 0; sl < kTsSlots; ++sl) ts_acc[sl] = Acc{0, 0, 0, 0};
  fetch(0);
  int cur = 0;
  for (int r0 = 0; r0 < m; r0 += R) {
    char* st = lds_raw + cur * STAGE;
    rowwork(st);                       // stage r0 -> buffer cur (free since the barrier that ended stage r0 - R)
    if (r0 + R < m) fetch(r0 + R);     // the next stage's rows are in flight during this stage's MFMAs
    if constexpr (!WANT_H) continue;
    __syncthreads();
    const char* op = st + (k * 64 + ci * 4);
    const T* rs = reinterpret_cast<const T*>(st + NB * PANEL) + k;
    // operands of the VIRTUAL blocks 0..4 (physical block (v + 2 wave) mod 8), of this wave's ninth tile, and the step's
    // residuals; the NEXT step's are in flight during this step's MFMAs (two register sets, alternating: no copies)
    auto load_ops = [&](const int q, T (&w)[5], T& sa, T& sb, T& rr) __attribute__((always_inline)) {
#pragma unroll
      for (int v = 0; v < 5; ++v) w[v] = *reinterpret_cast<const T*>(op + voff[v] + q * 256);
      sa = *reinterpret_cast<const T*>(op + soffa + q * 256);
      sb = *reinterpret_cast<const T*>(op + soffb + q * 256);
      rr = rs[4 * q];
    };
    // ONE MFMA site per register set for all four waves (hipcc keeps a separate accumulator set per site it finds over the same
    // accumulators only when their tile lists differ — these two are textually identical).  s_nop: VALU write -> MFMA read wait
    // states for operands hipcc may have moved through the VALU.
    auto mfma9 = [&](const T (&w)[5], const T sa, const T sb, const T rr) __attribute__((always_inline)) {
      asm volatile("s_nop 1\n\t"
                   "v_mfma_f32_16x16x4_f32 %0, %9, %9, %0\n\t"     // (0, 0)
                   "v_mfma_f32_16x16x4_f32 %1, %10, %10, %1\n\t"   // (1, 1)
                   "v_mfma_f32_16x16x4_f32 %2, %9, %10, %2\n\t"    // (0, 1)
                   "v_mfma_f32_16x16x4_f32 %3, %10, %11, %3\n\t"   // (1, 2)
                   "v_mfma_f32_16x16x4_f32 %4, %9, %11, %4\n\t"    // (0, 2)
                   "v_mfma_f32_16x16x4_f32 %5, %10, %12, %5\n\t"   // (1, 3)
                   "v_mfma_f32_16x16x4_f32 %6, %9, %12, %6\n\t"    // (0, 3)
                   "v_mfma_f32_16x16x4_f32 %7, %10, %13, %7\n\t"   // (1, 4)
                   "v_mfma_f32_16x16x4_f32 %8, %14, %15, %8"         // (x, x + 4): x = 0 for waves 0, 1; 1 for waves 2, 3
                   : "+a"(ts_acc[0]), "+a"(ts_acc[1]), "+a"(ts_acc[2]), "+a"(ts_acc[3]), "+a"(ts_acc[4]), "+a"(ts_acc[5]), "+a"(ts_acc[6]),
                     "+a"(ts_acc[7]), "+a"(ts_acc[8])
                   : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(sa), "v"(sb));
      ts_g[0] = fmaf(w[0], rr, ts_g[0]);   // J^T r of the physical blocks 2 wave and 2 wave + 1
      ts_g[1] = fmaf(w[1], rr, ts_g[1]);
    };
    T wA[5], saA, sbA, rrA, wB[5], saB, sbB, rrB;
    load_ops(0, wA, saA, sbA, rrA);
    for (int q = 0; q < R / 4; q += 2) {
      load_ops(q + 1, wB, saB, sbB, rrB);
      mfma9(wA, saA, sbA, rrA);
      load_ops(q + 2 < R / 4 ? q + 2 : 0, wA, saA, sbA, rrA);   // (the last one re-reads step 0: harmless)
      mfma9(wB, saB, sbB, rrB);
    }
    cur ^= 1;
  }
  // ||r||^2: the row workers' partial sums, one per wave (summed in fixed order by the caller)
  csum += __shfl_xor(csum, 4); csum += __shfl_xor(csum, 8); csum += __shfl_xor(csum, 16); csum += __shfl_xor(csum, 32);
  if (lane == 0) costw[wave] = csum;
  if constexpr (!WANT_H) return;
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the matrix pipe has drained before an accumulator is read
#pragma unroll
  for (int sl = 0; sl < kTsSlots; ++sl) asm volatile("" : "+a"(ts_acc[sl]));
  // every tile is final in the registers of the wave that owns it: straight to the L2-resident H (kept undamped for eval-only
  // iterations and the export) AND to the LDS image of the factorisation, which may overwrite the stage buffers once every
  // wave is past its last MFMA (the barrier); the undamped diagonal to hd (lm.h:108-117 acts on that copy)
  __syncthreads();
  T* Aimg = reinterpret_cast<T*>(lds_raw);
  const int LD = n | 1;
#pragma unroll
  for (int sl = 0; sl < 9; ++sl) {
    const int va = sl < 8 ? kVa[sl] : xsp, vb = sl < 8 ? kVb[sl] : xsp + 4;
    const int pa = (va + 2 * wave_u) & 7, pb = (vb + 2 * wave_u) & 7;
    const int qj = 16 * pb + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qi = 16 * pa + Mfma<T>::out_row(lane, r);
      const T v = ts_acc[sl][r];
      if (qi < n && qj < n) {
        Hs[qi * n + qj] = v;
        Aimg[qi * LD + qj] = v;
        if (pa != pb) { Hs[qj * n + qi] = v; Aimg[qj * LD + qi] = v; }
        else if (qi == qj) hd_out[qi] = v;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {   // J^T r of the physical block b = 2 wave + u: the four row groups' partials, then lanes 0..15
    const int b = 2 * wave_u + u;
    T gi = ts_g[u];
    gi += __shfl_xor(gi, 16);
    gi += __shfl_xor(gi, 32);
    const int q = 16 * b + lane;
    if (b < NB && lane < 16 && q < n) gout[q] = gi;
  }
}
template <typename T, int NB, bool ROBUST, bool TS>
__device__ __forceinline__ void large_fused_body(const LfArgs<T>& a);
template <typename T, int NB, bool ROBUST = false, bool TS = false>
__global__ void __launch_bounds__(256) large_fused_kernel(const LfArgs<T> a) {
  large_fused_body<T, NB, ROBUST, false>(a);
}
// the tile-split form must stay under 256 registers (two waves per SIMD = two workgroups per CU): ask for it
template <typename T, int NB>
__global__ void __launch_bounds__(256) large_fused_ts_kernel(const LfArgs<T> a) {
  large_fused_body<T, NB, false, true>(a);
}
template <typename T, int NB, bool ROBUST, bool TS>
__device__ __forceinline__ void large_fused_body(const LfArgs<T>& a) {
  static_assert(!TS || (sizeof(T) == 4 && !ROBUST), "tile-split pass: fp32, no M-estimator");
#ifdef TOA_LF_TIMING
  unsigned long long tk[6] = {0, 0, 0, 0, 0, 0};
  long long ck_pass = 0;
  unsigned long long tk_eval = 0;
#endif
  using Gram = DenseRowGram<T, NB, 1>;
  using Acc = typename Mfma<T>::Acc;
  constexpr int NT = Gram::NT;
  constexpr int NV = 16 * NB;  // padded vector length (>= n)
  // ring depth of the cost-only pass (see there).  fp64 with NB >= 5: the deeper ring's registers do not fit beside the rest
  // of the kernel, and hipcc then parks in-flight load destinations in AGPRs (tools/isa_lint.py rejects that code)
  constexpr int kEvalDepth = TS ? 2 : (sizeof(T) == 4 ? 4 : (NB <= 4 ? 3 : 2));   // (TS: the kernel lives under 256 registers)
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  T* Aimg = reinterpret_cast<T*>(lds_raw);  // n x (n | 1) image of the damped matrix, factored in place
  __shared__ __attribute__((aligned(16))) T xs[NV];
  __shared__ T g[NV], hd[NV], dx[NV], ldx[NV], rhs[NV], diag[NV];
  __shared__ T gfold[4][NV];
  __shared__ T costw[4];
  __shared__ int ninlw[4];
  __shared__ double red[256];
  __shared__ LmState<T> S;
  __shared__ int sh_p, sh_action, sh_cont;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = a.n, m = a.m;
  const toa_options& opt = a.opt;
  const toa_results& res = a.res;
  const bool is_lm = opt.solver_type == 0;
  char* my_scratch = a.scratch + size_t(blockIdx.x) * a.scratch_per_wg;
  Acc* part = reinterpret_cast<Acc*>(my_scratch);                                        // [4][NT][64]
  // Two slots for the undamped H [n][n] — the one the current linearisation lives in (sh_cur) and, while the memo is valid, the
  // one that holds the linearisation of the last ACCEPTED point (sh_memo_idx): an accumulate pass never writes over the parked
  // slot, a memo hit just makes it the current one again (no n x n copy either way) — and four vectors behind them.
  T* const Hs_base = reinterpret_cast<T*>(my_scratch + size_t(4) * NT * 64 * sizeof(Acc));
  T* const hdu = Hs_base + size_t(2) * n * n;   // undamped diagonal of the current linearisation (hd is damped in place, lm.h:108-117)
  T* const g_m = hdu + NV;                       // memo: J^T r, undamped diagonal, the x it was taken at
  T* const hdu_m = g_m + NV;
  T* const xs_m = hdu_m + NV;
  __shared__ int sh_cur, sh_memo_idx, sh_park, sh_check;
  __shared__ double lin_cost, memo_cost;         // normalised cost / inliers of the current and of the parked linearisation
  __shared__ int lin_ninl, memo_ninl;
  unsigned long long n_reused = 0;               // Builds served without streaming the rows (thread 0)
  // this wave's rows
  const int rows_per_wave = (((m + 3) / 4 + 3) / 4) * 4;
  const int row0 = wave * rows_per_wave;
  const int nrows = row0 >= m ? 0 : (m - row0 < rows_per_wave ? m - row0 : rows_per_wave);
  unsigned long long n_acc = 0, n_eval = 0, n_solves = 0, n_problems = 0;  // thread 0 only

  bool first = true;
  for (;;) {
    if (tid == 0) {
      int p = first ? int(blockIdx.x) : atomicAdd(a.queue, 1) + int(gridDim.x);
      sh_p = p;
    }
    first = false;
    __syncthreads();
    const long long p = sh_p;
    if (p >= a.P) break;
    const T* A = a.data + size_t(p) * m * (size_t(n) + 1);
    const T* bv = A + size_t(m) * n;
    if (tid == 0) {  // lm_init (lm.h:46-52, output.h:104-117, optimizer.h:248-250)
      S.lambda = opt.damping_init; S.prev_lambda = 0; S.bad_factor = opt.bad_factor; S.rebuild = 1;
      S.final_cost = kDblMax; S.final_nres = 0; S.final_ninl = 0; S.cost_ninl = 0; S.final_rerr = kDblMax;
      S.stop = TOA_STOP_NONE; S.num_iters = 0; S.num_failures = 0; S.num_consec = 0;
      S.cost_val = 0; S.cost_nres = 0;
      S.max_iters = opt.max_iters + 1 + (opt.check_final_cost ? 1 : 0);
      S.has_last_dx = 0; S.last_was_success = 1; S.iter = 0;
      S.acc_passes = S.eval_passes = S.solves = S.problems = 0;
      S.acc_at_x = 0; S.memo_valid = 0; S.memo_hit = 0;
      sh_cur = 0; sh_memo_idx = 0; sh_park = 0; sh_check = 0;
    }
    for (int i = tid; i < NV; i += 256) {
      xs[i] = i < n ? a.x[p * n + i] : T(0);
      dx[i] = T(0); ldx[i] = T(0); g[i] = T(0); hd[i] = T(0);
    }
    __syncthreads();

    for (;;) {  // one Build + Solve attempt per trip (a failed solve re-damps and retries, optimizer.h:358)
      const bool do_acc = !is_lm || S.rebuild;
      // The callback is a pure function of x (lm_device.hpp, lm_build_and_solve): when the linearisation at THIS x, bit for bit, is
      // still at hand (a failed solve re-entering Build, optimizer.h:358-393: skip = 1) or parked (the re-accumulation after a
      // roll-back that restored the accepted point exactly, optimizer.h:283-287 + :266: skip = 2) the rows are not streamed again.
      const int skip = (do_acc && a.memo) ? (S.acc_at_x ? 1 : (S.memo_hit ? 2 : 0)) : 0;
      if (tid == 0) {
        if (skip == 2) sh_cur = sh_memo_idx;
        else if (do_acc && !skip && S.memo_valid && sh_cur == sh_memo_idx) sh_cur ^= 1;   // never accumulate over the parked slot
      }
      __syncthreads();
      T* const Hs = Hs_base + size_t(sh_cur) * n * n;
      LF_TICK_START
      // ---------------- data pass: this wave's rows ----------------
      if (skip) {
        for (int i = tid; i < n; i += 256) {
          if (skip == 2) { g[i] = g_m[i]; hdu[i] = hdu_m[i]; }
          hd[i] = skip == 2 ? hdu_m[i] : hdu[i];
        }
        if (tid == 0 && skip == 2) { lin_cost = memo_cost; lin_ninl = memo_ninl; }
      } else {
        Gram gram;
        if constexpr (TS) {
          if (do_acc) {
            for (int i = tid; i < NV; i += 256) reinterpret_cast<T*>(lds_raw + 2 * (NB * 64 * 64 + 64 * 4))[i] = xs[i];   // x where the callee can name it
            __syncthreads();
            ts_data_pass<NB, true>(A, bv, n, m, xs, Hs, g, costw, hd);
            if (lane == 0) ninlw[wave] = 0;
          } else {
            for (int i = tid; i < NV; i += 256) reinterpret_cast<T*>(lds_raw + 2 * (NB * 64 * 64 + 64 * 4))[i] = xs[i];
            __syncthreads();
            ts_data_pass<NB, false>(A, bv, n, m, xs, Hs, g, costw, hd);
            if (lane == 0) ninlw[wave] = 0;
          }
        } else if (do_acc) {
          // fp64 with NB >= 7: 28 / 36 tiles of 8 registers do not fit the register file — two passes over the rows, half of
          // the tiles each (the pass is matrix-core bound: 32 flop / byte against a ridge of 10, reading the rows twice is free)
          constexpr bool kTwoPass = sizeof(T) == 8 && NB >= 7;
          constexpr int NTH = kTwoPass ? (NT + 1) / 2 : NT;
          Acc* mine = part + size_t(wave) * NT * 64;
          int ni = nrows;
          const T cl = gram.template pass_natural<true, TOA_LF_DEPTH, 0, NTH, true, ROBUST>(A + size_t(row0) * n, bv + row0, n, nrows, xs,
                                                                                           lane, a.loss, T(a.loss_th2), &ni);
#pragma unroll
          for (int t = 0; t < NTH; ++t) mine[t * 64 + lane] = gram.acc[t];
          if (lane < 16) {   // J^T r of this lane's NB columns and ||r||^2 (folded over the four row groups by the pass)
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) gfold[wave][NB * lane + cb] = gram.accT[Gram::ti(cb, 0)];
          }
          if (lane == 0) { costw[wave] = ROBUST ? cl : gram.accTT[0]; ninlw[wave] = ni; }   // ROBUST: sum of the losses, not r^T r
          if constexpr (kTwoPass) {
            gram.template pass_natural<true, TOA_LF_DEPTH, NTH, NT, false, ROBUST>(A + size_t(row0) * n, bv + row0, n, nrows, xs, lane,
                                                                                    a.loss, T(a.loss_th2));
#pragma unroll
            for (int t = NTH; t < NT; ++t) mine[t * 64 + lane] = gram.acc[t];
          }
        } else {
          // cost only: no matrix-core work to hide the HBM latency behind, and one wave per SIMD — three batches (24 KB
          // per wave) in flight instead of one: 245 -> 9x us per evaluate pass at n = 128, m = 4096
          int ni = nrows;
          const T c = gram.template pass_natural<false, kEvalDepth, 0, NT, true, ROBUST>(A + size_t(row0) * n, bv + row0, n, nrows, xs, lane,
                                                                                          a.loss, T(a.loss_th2), &ni);
          if (lane == 0) { costw[wave] = c; ninlw[wave] = ni; }
        }
      }
      __syncthreads();
#ifdef TOA_LF_TIMING
      if (!do_acc) tk_eval += wall_clock64() - tk_prev;
#endif
      LF_TICK(0)
#ifdef TOA_LF_TIMING
      ck_pass += clock64() - ck0;
#endif
      // ---------------- fold (fixed order) + Build (lm.h:59-120) ----------------
      const double cost_val = skip ? lin_cost : normalize_cost(double(T((costw[0] + costw[1]) + (costw[2] + costw[3]))), m, opt);
      bool built = m > 0 && cost_val != kDblMax;  // cost.h:83 isValid
      const int LD = n | 1;
      if constexpr (TS) {
        // no fold: ts_data_pass has left the finished H in the L2-resident copy AND in the image, the undamped diagonal in hd, the
        // raw J^T r in g.  (A pass whose cost turns out invalid has overwritten them too — like the reference, whose Build
        // accumulates into H_ before it looks at the cost, lm.h:59-80.)
        if (built && do_acc && !skip && opt.grad_clipping != 0)
          for (int i = tid; i < n; i += 256) { const T mm = opt.grad_clipping; g[i] = fmin(fmax(g[i], -mm), mm); }  // base.h:29-38
      } else if (built && do_acc && !skip) {
        // H = sum of the four partial Grams, to the L2-resident copy Hs (kept undamped for eval-only iterations and the
        // final export) AND straight into the LDS image the factorisation works on
#pragma unroll 3
        for (int it = 0; it < (NT * 64 + 255) / 256; ++it) {  // three tiles' partial loads in flight at once (all nine: 512 registers, spills)
          const int idx = tid + 256 * it;
          if (idx >= NT * 64) break;
          const int t = idx >> 6, l = idx & 63;
          const Acc v = (part[(0 * NT + t) * 64 + l] + part[(1 * NT + t) * 64 + l]) +
                        (part[(2 * NT + t) * 64 + l] + part[(3 * NT + t) * 64 + l]);
          int bi = 0, rem = t;  // tile t = (bi, bj), bi <= bj, row-major over the upper block triangle
          while (rem >= NB - bi) { rem -= NB - bi; ++bi; }
          const int bj = bi + rem;
          const int qj = NB * (l & 15) + bj;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int qi = NB * Mfma<T>::out_row(l, r) + bi;
            if (qi < n && qj < n) {
              Hs[qi * n + qj] = v[r];
              Aimg[qi * LD + qj] = v[r];
              if (bi != bj) { Hs[qj * n + qi] = v[r]; Aimg[qj * LD + qi] = v[r]; }
              else if (qi == qj) hd[qi] = v[r];   // the undamped diagonal (lm.h:108-117 acts on this copy)
            }
          }
        }
        for (int i = tid; i < n; i += 256) {
          T gi = (gfold[0][i] + gfold[1][i]) + (gfold[2][i] + gfold[3][i]);
          if (opt.grad_clipping != 0) { const T mm = opt.grad_clipping; gi = fmin(fmax(gi, -mm), mm); }  // base.h:29-38
          g[i] = gi;
        }
      }
      __syncthreads();
      if (do_acc && !skip && a.memo) {   // the undamped diagonal outlives the damping below (hd is damped in place)
        for (int i = tid; i < n; i += 256) hdu[i] = hd[i];
      }
      if (built && do_acc && opt.check_min_H_diag > 0) {  // lm.h:82-86 (workgroup-uniform condition)
        double low = 0;
        for (int i = tid; i < n; i += 256) low += fabs(hd[i]) < T(opt.check_min_H_diag) ? 1.0 : 0.0;
        if (wg_sum<T>(low, red) > 0) built = false;
      }
      __syncthreads();
      if (built && is_lm && S.lambda > T(0)) {  // lm.h:108-117, s in double
        const double s = S.rebuild ? 1.0 + double(S.lambda) : (1.0 + double(S.lambda)) / (1.0 + double(S.prev_lambda));
        for (int i = tid; i < n; i += 256) hd[i] = T(double(hd[i]) * s);
      }
      if (tid == 0) {
        if (skip) n_reused++; else if (do_acc) n_acc++; else n_eval++;
        S.cost_val = cost_val;
        S.cost_nres = m;
        S.cost_ninl = skip ? lin_ninl : (ROBUST ? (ninlw[0] + ninlw[1]) + (ninlw[2] + ninlw[3]) : m);
        if (do_acc) { lin_cost = cost_val; lin_ninl = S.cost_ninl; S.acc_at_x = 1; S.memo_hit = 0; }
      }
      for (int i = tid; i < NV; i += 256) rhs[i] = i < n ? g[i] : T(0);
      __syncthreads();
      LF_TICK(1)
      // ---------------- Solve (gn.h:150-171): blocked LDL^T of H with the damped diagonal (ldlt_wg.hpp) ----------------
      bool ldlt_ok = false;
      if (built) {
        if (do_acc && !skip) {  // the fold has filled the image; only the damped diagonal is missing
          for (int i = tid; i < n; i += 256) Aimg[i * LD + i] = hd[i];
        } else {       // H of the last build / of the linearisation read back (the previous factorisation overwrote the image)
          for (int e = tid; e < n * n; e += 256) {
            const int i = e / n, j = e - i * n;
            Aimg[i * LD + j] = (i == j) ? hd[i] : Hs[e];
          }
        }
        __syncthreads();
        LF_TICK(2)
        ldlt_ok = WgLdlt<T, NB>::factor(Aimg, LD, n, diag, tid);
        LF_TICK(3)
        if (ldlt_ok && tid < 64) WgLdlt<T, NB>::solve(Aimg, LD, n, diag, rhs, lane);
        __syncthreads();
        LF_TICK(4)
      }
      // ---------------- the rest of Step (optimizer.h:354-539) and of the loop body (:266-310) ----------------
      bool solver_failed = true;
      double dx_norm2 = 0, grad_norm2 = 0;
      if (built) {
        double bad = ldlt_ok ? 0.0 : 1.0, d2 = 0, g2 = 0;
        for (int i = tid; i < n; i += 256) {
          const T v = -rhs[i];
          if (!(fabs(v) <= NumLimits<T>::max())) bad = 1.0;
          dx[i] = v;
          d2 += double(v * v);
          g2 += double(g[i] * g[i]);
        }
        bad = wg_sum<T>(bad, red);
        dx_norm2 = double(T(wg_sum<T>(d2, red)));
        if (opt.min_grad_norm2 > 0.0f) grad_norm2 = double(T(wg_sum<T>(g2, red)));
        solver_failed = bad > 0;
      }
      if (tid == 0) {
        const unsigned max_tries = opt.max_consec_failures > 0 ? (opt.max_consec_failures > 1 ? opt.max_consec_failures : 1) : 255;
        int rc;  // 0 step, 1 solver failed for good, 2 early stop, -1 retry (same iteration, next trip)
        if (built) n_solves++;
        if (!solver_failed) {
          rc = 0;
        } else {  // optimizer.h:370-390
          S.num_consec = (S.num_consec + 1) & 0xff;
          S.num_failures = (S.num_failures + 1) & 0xff;
          if (S.cost_nres == 0) { S.stop = TOA_STOP_SKIPPED; rc = 2; }
          else if (isnan(S.cost_val) || isinf(S.cost_val)) { S.stop = TOA_STOP_NAN_OR_INF; rc = 2; }
          else if (opt.max_consec_failures > 0 && S.num_consec >= unsigned(opt.max_consec_failures)) {
            if (S.final_cost < double(NumLimits<T>::max())) S.stop = TOA_STOP_MAX_CONSEC_NO_DECR;
            rc = 1;
          } else {
            lm_bad_step(S, opt);  // FailedStep == BadStep  lm.h:148
            rc = (S.num_consec <= max_tries) ? -1 : 1;
          }
        }
        int action = 0;  // 1: x += dx, last_dx = dx ; 2: x -= last_dx
        int cont = 1;
        int park = 0, check = 0;
        if (rc >= 0) {
          int status = 0;
          if (rc == 1) S.stop = TOA_STOP_SOLVER_FAILED;  // :396-399
          if (rc == 0) status = lm_judge_core<T>(S, opt, res, p, dx_norm2, grad_norm2, true);
          bool eval_only = false;  // optimizer.h:269-309
          S.memo_hit = 0;           // (only the roll-back below may arm it, for the Build that follows directly)
          if (status & 1) {
            // x is about to leave an ACCEPTED point: if the step that follows is rejected the loop comes back here and accumulates
            // again — park this point's linearisation (lm_device.hpp, lm_iteration).  An eval-only iteration that succeeds leaves
            // from a point whose linearisation was never formed: nothing to park.
            if (a.memo) {
              if (S.acc_at_x) { park = 1; S.memo_valid = 1; sh_memo_idx = sh_cur; memo_cost = lin_cost; memo_ninl = lin_ninl; }
              else S.memo_valid = 0;
            }
            action = 1;
            S.acc_at_x = 0;
            S.has_last_dx = 1;
            S.last_was_success = 1;
            if (opt.check_final_cost && S.iter + 1 == S.max_iters) eval_only = true;
          } else {
            if (S.has_last_dx) { action = 2; S.has_last_dx = 0; S.acc_at_x = 0; check = (a.memo && S.memo_valid) ? 1 : 0; }
            else if (status & 2) { action = 1; S.has_last_dx = 1; S.acc_at_x = 0; }
            eval_only = (S.last_was_success == 0);
            S.last_was_success = 0;
          }
          if (is_lm) S.rebuild = eval_only ? 0 : 1;
          S.num_iters = S.num_iters + 1;
          S.iter = S.iter + 1;
          cont = (S.stop == TOA_STOP_NONE && S.iter < S.max_iters) ? 1 : 0;
        }
        sh_action = action;
        sh_cont = cont;
        sh_park = park;
        sh_check = check;
      }
      __syncthreads();
      const int action = sh_action;
      if (sh_park) for (int i = tid; i < n; i += 256) { g_m[i] = g[i]; hdu_m[i] = hdu[i]; xs_m[i] = xs[i]; }   // (x BEFORE the step)
      if (action == 1) for (int i = tid; i < n; i += 256) { const T d = dx[i]; xs[i] += d; ldx[i] = d; }  // traits.h:184-190
      if (action == 2) for (int i = tid; i < n; i += 256) xs[i] -= ldx[i];
      __syncthreads();
      if (sh_check) {
        // (x + dx) - dx is x again only when both roundings cancel: compare the BIT PATTERNS with the parked point's and let the
        // next Build read the memo back only on a match in every component — never an approximation
        double diff = 0;
        for (int i = tid; i < n; i += 256) diff += bits_equal(xs[i], xs_m[i]) ? 0.0 : 1.0;
        diff = wg_sum<T>(diff, red);
        if (tid == 0) S.memo_hit = diff == 0 ? 1 : 0;
        __syncthreads();
      }
      LF_TICK(5)
      if (!sh_cont) break;
    }
    // ---------------- optimizer.h:313-327: the problem is done ----------------
    for (int i = tid; i < n; i += 256) a.x[p * n + i] = xs[i];
    if (opt.save_last && res.final_hessian) {  // undamped (lm.h:157-171)
      double* Hout = res.final_hessian + size_t(p) * n * n;
      const T* const Hs = Hs_base + size_t(sh_cur) * n * n;   // H of the last build
      for (size_t e = tid; e < size_t(n) * n; e += 256) {
        const int i = int(e / n), j = int(e % n);
        T v = Hs[e];
        if (i == j) { v = hd[i]; if (is_lm && S.prev_lambda > T(0)) v = v / (T(1.0f) + S.prev_lambda); }
        Hout[e] = double(v);
      }
    }
    if (tid == 0) {
      if (S.stop == TOA_STOP_NONE && S.num_iters >= S.max_iters) S.stop = TOA_STOP_MAX_ITERS;  // :320-321
      res.stop_reason[p] = S.stop;
      res.num_iters[p] = S.num_iters;
      res.final_cost[p] = S.final_cost;
      if (res.num_failures) res.num_failures[p] = int(S.num_failures);
      if (res.num_consec_failures) res.num_consec_failures[p] = int(S.num_consec);
      if (res.final_num_residuals) res.final_num_residuals[p] = S.final_nres;
      if (res.final_rerr_dec) res.final_rerr_dec[p] = S.final_rerr;
      if (res.final_inlier_ratio) res.final_inlier_ratio[p] = S.final_nres > 0 ? float(S.final_ninl) / float(S.final_nres) : 1.0f;
      n_problems++;
    }
    __syncthreads();  // Hs / LDS of this problem are free again
  }
#ifdef TOA_LF_TIMING
  if (tid == 0 && blockIdx.x == 0)
    printf("large_fused wg0: passes %llu+%llu  shader clock %.0f MHz during the data pass; data pass %.1f us (of which the cost-only passes %.1f us)  fold+build %.1f us  image %.1f us  factor %.1f us  substitutions %.1f us  step %.1f us\n",
           n_acc, n_eval, double(ck_pass) / (tk[0] * 0.01), tk[0] * 0.01, tk_eval * 0.01, tk[1] * 0.01, tk[2] * 0.01, tk[3] * 0.01, tk[4] * 0.01, tk[5] * 0.01);
#endif
  if (tid == 0) {
    if (a.counters) {
      atomicAdd(&a.counters[0], n_acc);
      atomicAdd(&a.counters[1], n_eval);
      atomicAdd(&a.counters[2], n_solves);
      atomicAdd(&a.counters[3], n_problems);
      if (n_reused) atomicAdd(&a.counters[4], n_reused);
    }
    const int gone = atomicAdd(&a.queue[16], 1);  // the last workgroup to leave resets the queue for the next launch
    if (gone == int(gridDim.x) - 1) {
      __hip_atomic_store(&a.queue[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&a.queue[16], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// K1 / K2 seam for 64 <= n <= 128 (toa_accumulate with TOA_MODEL_DENSE_ROW_NATURAL): the data pass + fold of the kernel above
// on their own — g [P][n], H [P][n][n] full symmetric, cost, nres — a workgroup per problem, grid-stride.
template <typename T, int NB>
__global__ void __launch_bounds__(256) large_accumulate_kernel(const T* __restrict__ data, const T* __restrict__ x, const int n,
                                                               const int m, const long long P, const int want_grad,
                                                               T* __restrict__ g_out, T* __restrict__ H_out,
                                                               double* __restrict__ cost, int* __restrict__ nres,
                                                               char* __restrict__ scratch, const size_t scratch_per_wg) {
  using Gram = DenseRowGram<T, NB, 1>;
  using Acc = typename Mfma<T>::Acc;
  constexpr int NT = Gram::NT;
  constexpr int NV = 16 * NB;
  constexpr int kEvalDepth = sizeof(T) == 4 ? 4 : (NB <= 4 ? 3 : 2);
  __shared__ T xs[NV];
  __shared__ T gfold[4][NV];
  __shared__ T costw[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  Acc* part = reinterpret_cast<Acc*>(scratch + size_t(blockIdx.x) * scratch_per_wg);   // [4][NT][64]
  const int rows_per_wave = (((m + 3) / 4 + 3) / 4) * 4;
  const int row0 = wave * rows_per_wave;
  const int nrows = row0 >= m ? 0 : (m - row0 < rows_per_wave ? m - row0 : rows_per_wave);
  for (long long p = blockIdx.x; p < P; p += gridDim.x) {
    const T* A = data + size_t(p) * m * (size_t(n) + 1);
    const T* bv = A + size_t(m) * n;
    for (int i = tid; i < NV; i += 256) xs[i] = i < n ? x[p * n + i] : T(0);
    __syncthreads();
    {
      Gram gram;
      if (want_grad) {
        constexpr bool kTwoPass = sizeof(T) == 8 && NB >= 7;
        constexpr int NTH = kTwoPass ? (NT + 1) / 2 : NT;
        Acc* mine = part + size_t(wave) * NT * 64;
        gram.template pass_natural<true, TOA_LF_DEPTH, 0, NTH, true>(A + size_t(row0) * n, bv + row0, n, nrows, xs, lane);
#pragma unroll
        for (int t = 0; t < NTH; ++t) mine[t * 64 + lane] = gram.acc[t];
        if (lane < 16) {
#pragma unroll
          for (int cb = 0; cb < NB; ++cb) gfold[wave][NB * lane + cb] = gram.accT[Gram::ti(cb, 0)];
        }
        if (lane == 0) costw[wave] = gram.accTT[0];
        if constexpr (kTwoPass) {
          gram.template pass_natural<true, TOA_LF_DEPTH, NTH, NT, false>(A + size_t(row0) * n, bv + row0, n, nrows, xs, lane);
#pragma unroll
          for (int t = NTH; t < NT; ++t) mine[t * 64 + lane] = gram.acc[t];
        }
      } else {
        const T c = gram.template pass_natural<false, kEvalDepth>(A + size_t(row0) * n, bv + row0, n, nrows, xs, lane);
        if (lane == 0) costw[wave] = c;
      }
    }
    __syncthreads();
    if (tid == 0) {
      cost[p] = double(T((costw[0] + costw[1]) + (costw[2] + costw[3])));
      if (nres) nres[p] = m;
    }
    if (want_grad) {
      T* Hp = H_out + size_t(p) * n * n;
#pragma unroll 3
      for (int it = 0; it < (NT * 64 + 255) / 256; ++it) {
        const int idx = tid + 256 * it;
        if (idx >= NT * 64) break;
        const int t = idx >> 6, l = idx & 63;
        const Acc v = (part[(0 * NT + t) * 64 + l] + part[(1 * NT + t) * 64 + l]) +
                      (part[(2 * NT + t) * 64 + l] + part[(3 * NT + t) * 64 + l]);
        int bi = 0, rem = t;
        while (rem >= NB - bi) { rem -= NB - bi; ++bi; }
        const int bj = bi + rem;
        const int qj = NB * (l & 15) + bj;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qi = NB * Mfma<T>::out_row(l, r) + bi;
          if (qi < n && qj < n) {
            Hp[qi * n + qj] = v[r];
            if (bi != bj) Hp[qj * n + qi] = v[r];
          }
        }
      }
      for (int i = tid; i < n; i += 256) g_out[p * n + i] = (gfold[0][i] + gfold[1][i]) + (gfold[2][i] + gfold[3][i]);
    }
    __syncthreads();   // xs / gfold / the scratch block are free for the next problem
  }
}

template <typename T, int NB>
int launch_large_accumulate(toa_handle h, int n, int m, int64_t P, const T* data, const T* x, int want_grad, T* g, T* H,
                            double* cost, int32_t* nres) {
  using Acc = typename Mfma<T>::Acc;
  constexpr int NT = NB * (NB + 1) / 2;
  long long grid = std::min<long long>(P, (long long)h->num_cus * 2);
  const size_t per_wg = (size_t(4) * NT * 64 * sizeof(Acc) + 255) & ~size_t(255);
  const size_t need = per_wg * size_t(grid);
  if (need > h->scratch_bytes) {
    if (int rc = grow_sync(h, "device workspace")) return rc;
    toa_release_workspace(h, h->scratch);
    h->scratch = nullptr;
    h->scratch_bytes = 0;
    HIP_TRY(hipMalloc(&h->scratch, need));
    h->scratch_bytes = need;
  }
  hipLaunchKernelGGL((large_accumulate_kernel<T, NB>), dim3((unsigned)grid), dim3(256), 0, h->stream, data, x, n, m, (long long)P,
                     want_grad, g, H, cost, nres, static_cast<char*>(h->scratch), per_wg);
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}

template <typename T>
int large_accumulate_dispatch(toa_handle h, int n, int m, int64_t P, const T* data, const T* x, int want_grad, T* g, T* H,
                              double* cost, int32_t* nres) {
  switch ((n + 15) / 16) {
    case 4: return launch_large_accumulate<T, 4>(h, n, m, P, data, x, want_grad, g, H, cost, nres);
    case 5: return launch_large_accumulate<T, 5>(h, n, m, P, data, x, want_grad, g, H, cost, nres);
    case 6: return launch_large_accumulate<T, 6>(h, n, m, P, data, x, want_grad, g, H, cost, nres);
    case 7: return launch_large_accumulate<T, 7>(h, n, m, P, data, x, want_grad, g, H, cost, nres);
    case 8: return launch_large_accumulate<T, 8>(h, n, m, P, data, x, want_grad, g, H, cost, nres);
    default: return toa_fail(TOA_E_UNSUPPORTED, "large-n accumulate: n out of range");
  }
}

template <typename T, int NB, bool ROBUST, bool TS = false>
int launch_large_fused_r(toa_handle h, int n, int m, int64_t P, const T* data, T* x, const toa_options& opt, const toa_results& res,
                         uint64_t* counters) {
  using Acc = typename Mfma<T>::Acc;
  constexpr int NT = NB * (NB + 1) / 2;
  void (*kern)(const LfArgs<T>);
  if constexpr (TS) kern = large_fused_ts_kernel<T, NB>; else kern = large_fused_kernel<T, NB, ROBUST, false>;
  size_t lds = ((size_t(n) * (n | 1) + 16) * sizeof(T) + 15) & ~size_t(15);  // + the slack WgLdlt's unconditional reads may touch
  if (TS) lds = std::max(lds, size_t(2) * (size_t(NB) * 64 * 64 + 64 * 4) + size_t(16 * NB) * sizeof(T));   // the two stage buffers of the tile-split pass (+ x) live in the image
  int wg_per_cu = 0;
  for (int i = 0; i < h->ncfg; ++i)
    if (h->cfg[i].fn == (const void*)kern && h->cfg[i].lds == lds && h->cfg[i].wg_per_cu > 0) wg_per_cu = h->cfg[i].wg_per_cu;
  if (wg_per_cu == 0) {
    HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&wg_per_cu, kern, 256, lds));
    if (wg_per_cu < 1) return toa_fail(TOA_E_UNSUPPORTED, "large-n fused kernel does not fit this device");
    if (h->ncfg < 256) h->cfg[h->ncfg++] = {(const void*)kern, lds, wg_per_cu};
  }
  long long grid = (long long)h->num_cus * wg_per_cu;
  if (grid > P) grid = P;
  if (grid < 1) return TOA_OK;
  // 4 partial Grams + two H slots (the current linearisation and the parked one) + four vectors (undamped diagonal; memo: g, diagonal, x)
  const size_t per_wg = ((size_t(4) * NT * 64 * sizeof(Acc) + (size_t(2) * n * n + size_t(4) * 16 * NB) * sizeof(T)) + 255) & ~size_t(255);
  const size_t need = per_wg * size_t(grid);
  if (need > h->scratch_bytes) {
    if (int rc = grow_sync(h, "device workspace")) return rc;
    toa_release_workspace(h, h->scratch);
    h->scratch = nullptr;
    h->scratch_bytes = 0;
    HIP_TRY(hipMalloc(&h->scratch, need));
    h->scratch_bytes = need;
  }
  if (h->queue_dirty) {
    HIP_TRY(hipMemsetAsync(h->queue, 0, 48 * sizeof(int), h->stream));
    h->queue_dirty = false;
  }
  LfArgs<T> a;
  a.data = data; a.x = x; a.n = n; a.m = m; a.P = P; a.opt = opt; a.res = res;
  a.queue = h->queue;
  a.scratch = static_cast<char*>(h->scratch);
  a.scratch_per_wg = per_wg;
  a.counters = reinterpret_cast<unsigned long long*>(counters);
  a.loss = h->loss;
  a.loss_th2 = h->loss_th2;
  a.memo = h->tune.memo_off ? 0 : 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, h->stream, a);
  if (hipError_t e_ = hipGetLastError(); e_ != hipSuccess) {
    h->queue_dirty = true;
    return toa_fail(TOA_E_HIP, std::string("large_fused_kernel launch: ") + hipGetErrorString(e_));
  }
  return TOA_OK;
}

template <typename T, int NB>
int launch_large_fused(toa_handle h, int n, int m, int64_t P, const T* data, T* x, const toa_options& opt, const toa_results& res,
                       uint64_t* counters) {
  if (h->loss != TOA_LOSS_L2) {
    // fp64 beyond n = 96 (the two half-tile passes): with the M-estimator's registers on top hipcc copies in-flight load
    // destinations (tools/isa_lint.py rejects that code) — refused rather than built
    if constexpr (sizeof(T) == 8 && NB >= 7) return toa_fail(TOA_E_UNSUPPORTED, "toa_set_loss with fp64 natural-layout rows: n <= 96");
    else return launch_large_fused_r<T, NB, true>(h, n, m, P, data, x, opt, res, counters);
  }
  if constexpr (sizeof(T) == 4 && NB == 8) {   // (NB = 4: ten tiles do not divide by four waves, and hipcc copies accumulators between the
                                               //  waves' unequal tile lists right behind an MFMA — tools/isa_lint.py rejects that code)
    // the tile-split data pass (two workgroups per CU): rows of whole 16-byte column groups, 16-byte aligned.  toa_tuning::large_row_split: the
    // row-split pass (A/B; it also serves every other shape)
    const bool ts_off = h->tune.large_row_split != 0;
    if (!ts_off && n % 4 == 0 && (size_t(m) * (size_t(n) + 1)) % 4 == 0 && reinterpret_cast<uintptr_t>(data) % 16 == 0)
      return launch_large_fused_r<T, NB, false, true>(h, n, m, P, data, x, opt, res, counters);
  }
  return launch_large_fused_r<T, NB, false>(h, n, m, P, data, x, opt, res, counters);
}

template <typename T>
int large_fused_dispatch(toa_handle h, int n, int m, int64_t P, const T* data, T* x, const toa_options& opt,
                         const toa_results& res, uint64_t* counters) {
  switch ((n + 15) / 16) {
    case 4: return launch_large_fused<T, 4>(h, n, m, P, data, x, opt, res, counters);
    case 5: return launch_large_fused<T, 5>(h, n, m, P, data, x, opt, res, counters);
    case 6: return launch_large_fused<T, 6>(h, n, m, P, data, x, opt, res, counters);
    case 7: return launch_large_fused<T, 7>(h, n, m, P, data, x, opt, res, counters);   // (fp64, NB >= 7: two half-tile passes)
    case 8: return launch_large_fused<T, 8>(h, n, m, P, data, x, opt, res, counters);
    default: return toa_fail(TOA_E_UNSUPPORTED, "large-n fused kernel: n out of range");
  }
}

}  // namespace
}  // namespace toa

// 64 <= n <= 128 (fp64 beyond 96: two half-tile passes, see the kernel), and the LDL^T
// image n (n + 1) sizeof(T) must fit the LDS.
bool toa_large_fused_eligible(toa_handle h, int dtype, int n, int m) {
  if (h->tune.large_pipeline) return false;
  const size_t esz = dtype == TOA_F32 ? 4 : 8;
  if (n < 64 || n > 128) return false;
  if (size_t(n) * (n + 1) * esz + 16384 > size_t(h->max_lds)) return false;
  if ((unsigned long long)m * (unsigned long long)(n + 1) * esz >= (1ull << 32)) return false;  // 32-bit buffer offsets
  return true;
}

int toa_large_fused_lm_run(toa_handle h, int dtype, int n, int m, int64_t P, const void* data, void* x, const toa_options* options,
                           const toa_results* results, uint64_t* counters) {
  if (dtype == TOA_F32)
    return toa::large_fused_dispatch<float>(h, n, m, P, static_cast<const float*>(data), static_cast<float*>(x), *options, *results, counters);
  return toa::large_fused_dispatch<double>(h, n, m, P, static_cast<const double*>(data), static_cast<double*>(x), *options, *results, counters);
}

int toa_large_accumulate(toa_handle h, int dtype, int n, int m, int64_t P, const void* data, const void* x, int want_grad, void* g,
                         void* H, double* cost, int32_t* nres) {
  if (dtype == TOA_F32)
    return toa::large_accumulate_dispatch<float>(h, n, m, P, static_cast<const float*>(data), static_cast<const float*>(x), want_grad,
                                                 static_cast<float*>(g), static_cast<float*>(H), cost, nres);
  return toa::large_accumulate_dispatch<double>(h, n, m, P, static_cast<const double*>(data), static_cast<const double*>(x), want_grad,
                                                static_cast<double*>(g), static_cast<double*>(H), cost, nres);
}
