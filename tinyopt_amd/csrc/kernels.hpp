// Kernels of the LM hot path + their host launchers (templates).  Included by capi.hip (C-ABI,
// dispatch) and by inst.hip, which is compiled once per (dtype, block count) so that the ~40
// instantiations of the fused kernel build in parallel (see __graft_entry__.build()).
//
// Kernel inventory:
//   lm_fused_kernel        K1+K2+K3+K4: whole LM solves, one wavefront per problem at a time,
//                          dynamic problem queue, no inter-wave or host synchronisation.
//   accumulate_kernel      K1/K2 seam: the Accumulate callback for a batch (g, H, cost out).
//   solve_damped_kernel    K3 seam: damping + LDL^T solve for a batch.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#ifndef __HIPCC_RTC__
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <memory>
#include <string>
#include <vector>
#endif

#ifdef __HIPCC_RTC__   // run-time compilation (jit.hip): the headers are handed to hiprtc by NAME, embedded in the library
#include "tinyopt_amd.h"
#else
#include "../../include/tinyopt_amd.h"
#endif
#include "dense_row.hpp"
#include "jet.hpp"
#include "ldlt_blocked.hpp"
#include "ldlt_lds.hpp"
#include "ldlt_regs.hpp"
#include "lm_device.hpp"
#include "robust.hpp"
#include "wave_utils.hpp"

namespace toa {

// ------------------------------------------------------------------------------------------------
// Device residual models.  Concept (see lm_device.hpp): Scalar, kNpad, init(n, m, data), bind(p),
// accumulate / evaluate / write_sym.
// ------------------------------------------------------------------------------------------------
// x (+)= sign * d for Euclidean parameter blocks (traits.h:184-190)
template <typename T>
__device__ __forceinline__ void euclid_plus_eq(WaveLds<T>& L, const T* d, T sign, int lane) {
  L.xs[lane] += sign * d[lane];
}

// ---- cooperative passes (the tail of a fused launch) ------------------------------------------------------------------------
// Once the work queue is dry, the waves of a workgroup that have nothing left help the ones that do: a data pass is K CHUNKS
// of rows, handed out by ticket to whoever asks (the owner of the problem included), each accumulated from zero and folded
// into the owner's LDS total in TICKET ORDER.  The fold is therefore the same fixed-order sum whether the owner computed all
// K chunks itself (steady state: nobody is idle) or its three siblings took some: results do not depend on timing, on the
// batch size or on the position of a problem in the batch.  All of it lives in LDS at workgroup scope; no barrier (the
// four waves run different problems at their own pace), no HBM traffic.
struct CoopSlot {          // one per wave, written by the OWNER except ticket (everybody) and turn (whoever folds)
  int p;                   // problem of the open pass
  int ticket;              // next chunk of the open ACCUMULATE pass to hand out; >= K: no open pass.  Evaluate-only passes never
                           // touch it, so whatever ticket a helper draws — however long ago it looked at the counter — belongs
                           // to an accumulate pass of this slot, and the acquire half of the fetch-add shows it that pass's
                           // problem and x
  int turn;                // next chunk whose partial may be folded; == K: the pass is complete
  int pad_[5];
};
constexpr int kCoopMaxWaves = 4;
struct CoopCtl {
  CoopSlot slot[kCoopMaxWaves];
  int active[kCoopMaxWaves];   // wave w still has (or may still get) problems of its own
};
constexpr int kCoopCtlBytes = 512;   // (the size the round-3 .. 5 layouts reserved: LDS geometry, and with it every measured number, unchanged)
static_assert(sizeof(CoopCtl) <= kCoopCtlBytes, "control block");

// ROBUST = true: the variant whose passes apply the handle's M-estimator (toa_set_loss) to every residual.  It exists only in
// the small kernels of the launch-per-iteration forms (accumulate_kernel, wide_partial_kernel): compiled into the fused
// kernel, the estimators' exp / log / atan2 raise its register count from 168 to 232 (3 -> 2 waves per SIMD for everybody).
// COOP = true: the variant whose passes are ALWAYS the ticketed chunk form (coop_K >= 1; one chunk = the classic pass, bit
// for bit) — instantiated by the fused kernel only.  A compile-time property, not a run-time branch: two MFMA loops over
// the same accumulators in one kernel made hipcc keep two AGPR sets (156 -> 196 registers at n = 50: 3 -> 2 waves / SIMD).
// (A TEAM form of this kernel — twelve-wave workgroups of which two pull problems, so that the rows of the problems in flight stay in
// the 256 MiB Infinity Cache — was built in round 5, bit-identical, measured slower (9.28 vs 7.3 ms at C4) and removed from the
// library in round 6: profiles/r05_ab_log.md §1, profiles/r06_pruned_arms.patch.)
template <typename T, int NBM, int THIN, bool ROBUST = false, bool COOP = false>
struct DenseRowModel {
  using Scalar = T;
  static constexpr int kWaves = 4;                // waves per workgroup of the fused kernel
  static constexpr int kXdim = 0;  // parameters per problem as stored in x; 0 = n (Euclidean)
  __device__ __forceinline__ void plus_eq(WaveLds<T>& L, const T* d, T sign, int, int lane) const { euclid_plus_eq(L, d, sign, lane); }
  // register-LDL^T width: the largest n this (NBM, THIN) layout serves, rounded to the 8-column chunk (n = 50: 56, not 64)
  static constexpr int kNmax = THIN > 0 ? 16 * NBM + THIN - 1 : 16 * NBM - 1;
  static constexpr int kNpad = (kNmax + 7) & ~7;
  DenseRowGram<T, NBM, THIN> gram;
  const T* data;
  const T* prob;
  DenseRowLayout lay;
  int m;
  int loss;        // TOA_LOSS_* applied to every residual (toa_set_loss; 0 = plain squared L2)
  T th2;
  int rows_real;   // rows of the bound problem / chunk that exist (the packed layout pads to a multiple of 4)
  int ninl;        // inlier residuals of the last pass; -1 = all of them (no loss)
  // cooperative passes (fused kernel only; see CoopCtl above): chunks per pass (0 = off), steps per chunk, and where the
  // workgroup's control block / the per-wave carves sit in LDS
  // COOP on a 64-row super-batch layout (fp64, n <= 15) selects the fused kernel's OTHER special form instead: the row-per-lane
  // pass through an LDS stage of the wave (DenseRowGram::pass16s).  (Its cooperative form was measured and rejected.)
  static constexpr bool kStaged = COOP && DenseRowGram<T, NBM, THIN>::kSuper16;
  static constexpr bool kCoop = COOP && !kStaged;
  static constexpr size_t kStageBytes = kStaged ? size_t(DenseRowGram<T, NBM, THIN>::kStageBytes) : 0;
  unsigned char* stage;   // kStaged: this wave's LDS stage
  static constexpr int kCoopPeriod = DenseRowGram<T, NBM, THIN>::kSuper16 ? 16 : 8;   // steps per super-batch / per turn of the load ring (kDepth * U)
  static_assert(!(COOP && ROBUST), "no cooperative form of the robust passes");
  int coop_K, coop_cs, coop_lds_per_wave, coop_tot_off, cur_p, helping, help_o, help_c;
  __device__ __forceinline__ void init(int n, int m_, const void* d) {
    m = m_;
    lay = DenseRowLayout::make(n, m_);
    data = static_cast<const T*>(d);
    loss = TOA_LOSS_L2; th2 = T(0); rows_real = m_; ninl = -1;
    coop_K = 0; coop_cs = 0; coop_lds_per_wave = 0; coop_tot_off = 0; cur_p = 0; helping = 0; help_o = 0; help_c = 0;
    stage = nullptr;
  }
  // the workgroup's control block sits behind the kWaves carves
  __device__ __forceinline__ CoopCtl* coop_ctl() const {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    return reinterpret_cast<CoopCtl*>(smem + size_t(kWaves) * coop_lds_per_wave);
  }
  // tot_off: where in a wave's carve the chunk partials of ITS passes are summed — 0 = its LDL^T workspace M (free during a
  // pass; the carve starts with it), or an area of its own when M is smaller than the Gram registers (n = 12 fp64)
  __device__ __forceinline__ void coop_init(int K, int chunk_steps, int lds_per_wave, int tot_off) { coop_K = K; coop_cs = chunk_steps; coop_lds_per_wave = lds_per_wave; coop_tot_off = tot_off; }
  __device__ __forceinline__ void set_loss(int kind, double t2) { loss = kind; th2 = T(t2); }
#ifdef TOA_ABL_REUSE  // ablation: every wave streams one of TOA_ABL_REUSE problems (cache-resident data, same instruction stream).
  // 64 (26 MB): concurrent readers of a problem share an XCD's L2; 518 = 4 * 129 + 2 (211 MB): the co-readers p, p + 518, ...
  // sit on consecutive XCDs (workgroup -> XCD is round-robin), so every re-read is served by the Infinity Cache, none by an L2
  __device__ __forceinline__ void bind(long long p) { prob = data + size_t(p % (TOA_ABL_REUSE)) * lay.elems_per_problem(); rows_real = m; }
#else
  __device__ __forceinline__ void bind(long long p) { prob = data + size_t(p) * lay.elems_per_problem(); rows_real = m; cur_p = int(p); }
#endif
  // row-split execution: restrict the model to rows [row0, row0 + rows) of problem p (rows % 4 == 0)
  __device__ __forceinline__ void bind_chunk(long long p, int row0, int rows, int n) {
    const DenseRowLayout full = DenseRowLayout::make(n, m);
    prob = data + size_t(p) * full.elems_per_problem() + size_t(row0) * full.rs;
    lay = full;
    lay.m4 = rows;
    rows_real = max(0, min(rows, m - row0));
  }
  // One pass in the ticketed chunk form — the owner's side AND the helper's side, in ONE loop: hipcc keeps a separate AGPR
  // set alive for every MFMA loop over the accumulators it finds in a kernel (two inlined copies of the chunk loop took
  // the n = 50 kernel from 156 to 196 registers, 3 -> 2 waves / SIMD), so the kernel may contain exactly one.
  //   owner   (helping == 0): opens a pass on its own slot, takes its tickets like everybody else, waits for the last fold,
  //           reads the total back into the Gram registers;
  //   helper  (helping == 1 — see lm_fused_kernel's "ghost problem"): this wave
  //           has no problem left; it serves the siblings' open ACCUMULATE passes until none of them is active.  (Evaluate-
  //           only passes — one in seven at C4 — are the same chunks summed in the same order by the owner alone, coop_eval:
  //           a second kind of chunk in this loop costs the kernel its third wave per SIMD.)
  // The shape of the loop is what hipcc's register allocation tolerated (A/B log, profiles/r03_ab_log.md): do-while, the
  // scalars that cross the pass re-derived behind optimisation barriers, the total read back through in-out asm operands.
  __device__ __forceinline__ T coop_eval(WaveLds<T>& L, const int n, const int lane) {
    // Evaluate-only pass: the same chunks, summed in the same order, by the owner alone — no ticket, no slot.
    // (A helper that looked at this slot's counter during the previous accumulate pass and draws its ticket only now must
    // never land in a pass of a different kind — ADVICE r03: the accumulate counter stays closed across evaluate-only passes.)
    (void)L;
    const int st = lay.m4 >> 2;
    T tot = T(0);
    for (int c = 0; c < coop_K; ++c) {
      reg_fence();
      const T part = gram.template pass_chunk<false>(prob, lay, n, L.xs, lane, c * coop_cs, min(st, (c + 1) * coop_cs));
      reg_fence();
      tot = c == 0 ? part : tot + part;
    }
    return tot;
  }
  __device__ __forceinline__ void coop_acc(WaveLds<T>& L, const int n, const int lane) {
    constexpr bool WANT_H = true;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int w = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
    const int st = lay.m4 >> 2;
    const bool help = helping != 0;
    int c, o;
    if (!help) {
      o = w;
      CoopSlot& S = coop_ctl()->slot[w];
      if (lane == 0) {
        S.p = cur_p;
        S.turn = 0;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // x (L.xs) and the fields above, before the counter opens
      // the counter opens at 1: chunk 0 is the owner's (it always has a valid ticket when it enters the loop below)
      if (lane == 0) __hip_atomic_store(&S.ticket, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      c = 0;
    } else {   // the ticket lm_fused_kernel's search found for this ghost
      o = __builtin_amdgcn_readfirstlane(help_o);
      c = __builtin_amdgcn_readfirstlane(help_c);
    }
    do {   // (both sides arrive with a valid ticket: no guard — a guard costs the kernel its third wave per SIMD)
      const T* xs_o = WaveLds<T>::carve(smem + size_t(o) * coop_lds_per_wave, n).xs;
      const T* pr = help ? data + size_t(__builtin_amdgcn_readfirstlane(coop_ctl()->slot[o].p)) * lay.elems_per_problem()
                         : prob;
      reg_fence();
      const T part = gram.template pass_chunk<WANT_H>(pr, lay, n, xs_o, lane, c * coop_cs, min(st, (c + 1) * coop_cs));
      reg_fence();
      // Everything the fold needs is re-derived from the two scalars that crossed the pass, behind an optimisation barrier.
      c = __builtin_amdgcn_readfirstlane(c);
      o = __builtin_amdgcn_readfirstlane(o);
      asm volatile("" : "+s"(c), "+s"(o));
      CoopSlot& S = coop_ctl()->slot[o];
      T* totp = reinterpret_cast<T*>(smem + size_t(o) * coop_lds_per_wave + coop_tot_off);
      // fold in ticket order
      // (bounded: a protocol bug must end in a trapped launch, not in a GPU that never comes back — ~1 s of polling)
      for (int spin = 0; __hip_atomic_load(&S.turn, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != c; ++spin) {
        __builtin_amdgcn_s_sleep(2);
        if (spin > (1 << 24)) asm volatile("s_trap 2");
      }
      (void)part;
      if (coop_K > 1) {   // (one chunk per pass: the registers ARE the total)
        if (c == 0) gram.memo_save(totp, lane);
        else gram.memo_add(totp, lane);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_store(&S.turn, c + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      // next ticket of the same pass
      if (lane == 0) c = __hip_atomic_fetch_add(&S.ticket, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
      c = __builtin_amdgcn_readfirstlane(c);
    } while (c < coop_K);
    // (A helper falls through the owner's epilogue as well — its own slot's turn has been K since its last pass, and what
    //  the read-back puts into its dead Gram registers does not matter: an early return for it here, i.e. a path on which
    //  the accumulators die, made hipcc allocate 16 more registers for the whole kernel.)
    CoopSlot& S = coop_ctl()->slot[w];
    for (int spin = 0; __hip_atomic_load(&S.turn, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < coop_K; ++spin) {
      __builtin_amdgcn_s_sleep(2);
      if (spin > (1 << 24)) asm volatile("s_trap 2");
    }
    if (coop_K > 1) gram.memo_load_inplace(reinterpret_cast<T*>(smem + size_t(w) * coop_lds_per_wave + coop_tot_off), lane);
    gram.fold_thin();
  }
  // A wave whose queue is dry looks for a sibling's open ACCUMULATE pass and takes a ticket of it (only accumulate passes
  // ever open the counter: a ticket drawn late still names a chunk of an accumulate pass).  false: no sibling is active any more.
  __device__ __forceinline__ bool coop_find(const int lane) {
    const int w = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
    CoopCtl* ctl = coop_ctl();
    if (!helping) {
      helping = 1;
      help_o = w;
      if (lane == 0) __hip_atomic_store(&ctl->active[w], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    for (int spins = 0;;) {
      bool any = false;
      for (int t = 1; t <= 3; ++t) {
        const int q = (help_o + t) & 3;
        if (q == w) continue;
        if (__hip_atomic_load(&ctl->active[q], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) continue;
        any = true;
        if (__hip_atomic_load(&ctl->slot[q].ticket, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < coop_K) {
          int cc = 0;
          if (lane == 0) cc = __hip_atomic_fetch_add(&ctl->slot[q].ticket, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
          cc = __builtin_amdgcn_readfirstlane(cc);
          if (cc < coop_K) {
            help_o = __builtin_amdgcn_readfirstlane(q);
            help_c = cc;
            return true;
          }
        }
      }
      if (!any) return false;
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1 << 24)) asm volatile("s_trap 2");   // (a sibling that never finishes: trap rather than hang)
    }
  }
  __device__ __forceinline__ void accumulate(WaveLds<T>& L, int n, int lane, T& cost, int& nres) {
    if constexpr (kCoop) {
      ninl = -1;
      coop_acc(L, n, lane);
      if (helping) {   // the ghost problem of a wave whose queue is dry (lm_fused_kernel): "no residuals" ends it at once
        cost = T(0);
        nres = 0;
        return;
      }
      cost = gram.extract_g_diag_cost(L.g, L.hd, lay, n, lane, L.tmp);
      nres = m;
    } else {
      const T cl = gram.template pass<true, ROBUST, kStaged>(prob, lay, n, L.xs, lane, loss, th2, rows_real, &ninl, stage);
      cost = gram.extract_g_diag_cost(L.g, L.hd, lay, n, lane, L.tmp);
      if constexpr (ROBUST) cost = cl;   // sum of the robust losses, not the Gram's r^T r (which is scaled by s)
      nres = m;
    }
  }
  __device__ __forceinline__ void evaluate(WaveLds<T>& L, int n, int lane, T& cost, int& nres) {
    if constexpr (kCoop) {
      ninl = -1;
      cost = coop_eval(L, n, lane);
    } else {
      cost = gram.template pass<false, ROBUST, kStaged>(prob, lay, n, L.xs, lane, loss, th2, rows_real, &ninl, stage);
    }
    nres = m;
  }
  template <typename O>
  __device__ __forceinline__ void write_sym(O* M, int LD, int n, int lane) const {
    gram.write_sym(M, LD, lay, n, lane);
  }
  // memo of the last accepted linearisation (lm_device.hpp): the Gram registers parked in / read back from the wave's HBM slot
  static constexpr bool kMemo = !ROBUST;   // (with a loss the cost is the pass's own sum, not a Gram entry)
  static constexpr size_t kMemoBytes = size_t(DenseRowGram<T, NBM, THIN>::kMemoElems) * sizeof(T);
  __device__ __forceinline__ void memo_save(WaveLds<T>& L, int lane) const { gram.memo_save(reinterpret_cast<T*>(L.st->memo_slot), lane); }
  __device__ __forceinline__ void memo_reextract(WaveLds<T>& L, int n, int lane, T& cost, int& nres) {
    cost = gram.extract_g_diag_cost(L.g, L.hd, lay, n, lane, L.tmp);
    nres = m;
  }
  __device__ __forceinline__ void memo_restore(WaveLds<T>& L, int n, int lane, T& cost, int& nres) {
    gram.memo_load(reinterpret_cast<const T*>(L.st->memo_slot), lane);
    memo_reextract(L, n, lane, cost, nres);
  }
};

// The model whose data passes honour toa_set_loss, for the kernels that only run passes (Model itself where the family
// has no separate variant: the Jet models branch at run time, the others have no M-estimator).
template <typename M> struct RobustOf { using type = M; };
template <typename T, int NBM, int THIN, bool COOP> struct RobustOf<DenseRowModel<T, NBM, THIN, false, COOP>> { using type = DenseRowModel<T, NBM, THIN, true, false>; };

// Gaussian prior  r = (x - y) / sigma,  m = n — the residual of the reference's published dense
// benchmark, with the semantics of its manual Accumulate callback (benchmarks/dense.cpp:57-66, 90-99;
// losses/mahalanobis.h:124-136): grad = J * res with J = diag(1/sigma), H.diagonal() = sigma^-2 (H was
// cleared: off-diagonals are 0), returns res.squaredNorm() as a SCALAR => Cost(v, 1) (cost.h:22).
// Data per problem: [y (n) | sigma (n)].  Same operation order as the oracle => g and H bit-identical.
template <typename T, int NPAD>
struct GaussianPriorModel {
  using Scalar = T;
  __device__ __forceinline__ void set_loss(int, double) {}  // no M-estimator on this family
  static constexpr int kXdim = 0;
  __device__ __forceinline__ void plus_eq(WaveLds<T>& L, const T* d, T sign, int, int lane) const { euclid_plus_eq(L, d, sign, lane); }
  static constexpr int kNpad = NPAD;
  const T* data;
  const T* y;
  const T* sigma;
  int n_;
  __device__ __forceinline__ void init(int n, int, const void* d) { n_ = n; data = static_cast<const T*>(d); }
  __device__ __forceinline__ void bind(long long p) { y = data + size_t(p) * 2 * n_; sigma = y + n_; }
  __device__ __forceinline__ void bind_chunk(long long p, int, int, int) { bind(p); }  // not row-splittable: one chunk
  __device__ __forceinline__ T residual(const WaveLds<T>& L, int n, int lane, T& inv_sigma) const {
    if (lane >= n) { inv_sigma = T(0); return T(0); }
    const T s = sigma[lane];
    inv_sigma = T(1) / s;
    return (L.xs[lane] - y[lane]) / s;
  }
  __device__ __forceinline__ void accumulate(WaveLds<T>& L, int n, int lane, T& cost, int& nres) {
    T is;
    const T r = residual(L, n, lane, is);
    if (lane < n) { L.g[lane] = is * r; L.hd[lane] = is * is; }
    cost = wave_allreduce_sum(r * r);
    nres = 1;
    wave_sync();
  }
  __device__ __forceinline__ void evaluate(WaveLds<T>& L, int n, int lane, T& cost, int& nres) {
    T is;
    const T r = residual(L, n, lane, is);  // MahaSquaredNorm(x - y, stdevs), dense.cpp:63-65
    cost = wave_allreduce_sum(r * r);
    nres = 1;
  }
  template <typename O>
  __device__ __forceinline__ void write_sym(O* M, int LD, int n, int lane) const {
    if (lane < n)
      for (int j = 0; j < n; ++j) M[lane * LD + j] = O(0);
  }
};

// Gaussian prior with a GENERAL covariance, whitened by the upper Cholesky factor U of the information matrix:
// res = U (x - y), J = U  (losses/mahalanobis.h:160-171 MahaWhitenedInfoU; tests/cov.cpp:91-146), folded as the AD
// bridge folds a residual vector: grad = J^T res, H = J^T J (= cov^-1), cost = ||res||^2 over n residuals.
// data: [P][n + n*n] = y, then U row-major (upper triangular).  Lane a owns residual a / gradient entry a / row a of H.
// A parity model (tests/cov.cpp: the covariance of the solve must equal the prior's), not a throughput model: H is
// recomputed from U at every build.
template <typename T, int NPAD>
struct MahaPriorModel {
  using Scalar = T;
  __device__ __forceinline__ void set_loss(int, double) {}  // no M-estimator on this family
  static constexpr int kXdim = 0;
  static constexpr int kNpad = NPAD;
  __device__ __forceinline__ void plus_eq(WaveLds<T>& L, const T* d, T sign, int, int lane) const { euclid_plus_eq(L, d, sign, lane); }
  const T* data;
  const T* y;
  const T* U;
  int n_;
  __device__ __forceinline__ void init(int n, int, const void* d) { n_ = n; data = static_cast<const T*>(d); }
  __device__ __forceinline__ void bind(long long p) { y = data + size_t(p) * (n_ + size_t(n_) * n_); U = y + n_; }
  __device__ __forceinline__ void bind_chunk(long long p, int, int, int) { bind(p); }
  __device__ __forceinline__ T residual(WaveLds<T>& L, int n, int lane) const {
    L.tmp[lane] = lane < n ? L.xs[lane] - y[lane] : T(0);
    wave_sync();
    T r = 0;
    if (lane < n)
      for (int j = lane; j < n; ++j) r += U[size_t(lane) * n + j] * L.tmp[j];  // triangularView<Upper>
    return r;
  }
  __device__ __forceinline__ void accumulate(WaveLds<T>& L, int n, int lane, T& cost, int& nres) {
    const T r = residual(L, n, lane);
    L.vec[lane] = r;
    wave_sync();
    if (lane < n) {
      T g = 0, hd = 0;
      for (int i = 0; i <= lane; ++i) {  // column `lane` of U has its non-zeros in rows 0..lane
        const T u = U[size_t(i) * n + lane];
        g += u * L.vec[i];
        hd += u * u;
      }
      L.g[lane] = g;
      L.hd[lane] = hd;
    }
    cost = wave_allreduce_sum(r * r);
    nres = n;
    wave_sync();
  }
  __device__ __forceinline__ void evaluate(WaveLds<T>& L, int n, int lane, T& cost, int& nres) {
    const T r = residual(L, n, lane);
    cost = wave_allreduce_sum(r * r);
    nres = n;
    wave_sync();
  }
  template <typename O>
  __device__ __forceinline__ void write_sym(O* M, int LD, int n, int lane) const {
    if (lane < n)
      for (int b = 0; b < n; ++b) {  // H[a][b] = sum_i U[i][a] U[i][b], i <= min(a, b)
        const int top = lane < b ? lane : b;
        T h = 0;
        for (int i = 0; i <= top; ++i) h += U[size_t(i) * n + lane] * U[size_t(i) * n + b];
        M[lane * LD + b] = O(h);
      }
  }
};

// The analytic test functions of the reference's optimizer tests as MANUAL Accumulate callbacks
// (`auto loss = [&](const auto& v, auto& grad, auto& H)`), exact Hessians included — they drive the LM state
// machine through its bad-step, failed-solve (indefinite H) and rollback branches:
//   0 Rosenbrock  tests/optimize_easy.cpp:35-79     1 plateau (Easom-like)  :88-144     2 Powell singular  :153-221
//   3 Beale       tests/optimize_hard.cpp:34-63     4 Himmelblau            :72-102   (residual vectors, J^T J / J^T r)
//   5 x - 2       tests/basic.cpp:41-54,72-87 (n = 1): grad = res, H = 1, cost = |res|
// data: [1] = function id (as T).  Every lane evaluates the same scalars (n <= 4): no divergence, no reductions.
template <typename T>
struct TestFnModel {
  using Scalar = T;
  __device__ __forceinline__ void set_loss(int, double) {}  // no M-estimator on this family
  static constexpr int kXdim = 0;
  static constexpr int kNpad = 16;
  __device__ __forceinline__ void plus_eq(WaveLds<T>& L, const T* d, T sign, int, int lane) const { euclid_plus_eq(L, d, sign, lane); }
  int fn;
  T H[16];
  __device__ __forceinline__ void init(int, int, const void* d) { fn = int(static_cast<const T*>(d)[0]); }
  __device__ __forceinline__ void bind(long long) {}
  __device__ __forceinline__ void bind_chunk(long long, int, int, int) {}
  static __device__ __forceinline__ T pw(T t, int e) { return T(::pow(double(t), double(e))); }  // std::pow(t, 3): double
  template <bool WANT>
  __device__ __forceinline__ T eval(const WaveLds<T>& L, T* g, int& nres) {
    const T v0 = L.xs[0], v1 = L.xs[1], v2 = L.xs[2], v3 = L.xs[3];
    nres = 1;
    if (fn == 5) {
      const T res = v0 - T(2);
      if (WANT) { g[0] = res; H[0] = T(1); }
      return res < T(0) ? -res : res;
    }
    if (fn == 0) {
      const T t1 = T(1.0) - v0, t2 = v1 - v0 * v0;
      if (WANT) {
        g[0] = T(-2.0) * t1 - T(400.0) * v0 * t2;
        g[1] = T(200.0) * t2;
        H[0] = T(2.0) - T(400.0) * v1 + T(1200.0) * v0 * v0;
        H[1] = H[4] = T(-400.0) * v0;
        H[5] = T(200.0);
      }
      return t1 * t1 + T(100.0) * t2 * t2;
    }
    if (fn == 1) {
      const T PI = T(3.14159265358979323846);
      const T dx = v0 - PI, dy = v1 - PI;
      const T ex = T(::exp(-(dx * dx + dy * dy)));
      const T cx = T(::cos(v0)), cy = T(::cos(v1)), sx = T(::sin(v0)), sy = T(::sin(v1));
      if (WANT) {
        g[0] = cy * ex * (sx + T(2.0) * dx * cx);
        g[1] = cx * ex * (sy + T(2.0) * dy * cy);
        H[0] = cy * ex * (cx - T(4.0) * dx * sx + (T(2.0) - T(4.0) * dx * dx) * cx);
        H[5] = cx * ex * (cy - T(4.0) * dy * sy + (T(2.0) - T(4.0) * dy * dy) * cy);
        H[1] = H[4] = ex * (sx + T(2.0) * dx * cx) * (sy + T(2.0) * dy * cy);
      }
      return T(1.0) - (cx * cy * ex);
    }
    if (fn == 2) {
      const T t1 = v0 + T(10.0) * v1, t2 = v2 - v3, t3 = v1 - T(2.0) * v2, t4 = v0 - v3;
      if (WANT) {
        g[0] = T(2.0) * t1 + T(40.0) * pw(t4, 3);
        g[1] = T(20.0) * t1 + T(4.0) * pw(t3, 3);
        g[2] = T(10.0) * t2 - T(8.0) * pw(t3, 3);
        g[3] = T(-10.0) * t2 - T(40.0) * pw(t4, 3);
#pragma unroll
        for (int i = 0; i < 16; ++i) H[i] = T(0);
        const T d3 = T(12.0) * t3 * t3, d4 = T(120.0) * t4 * t4;
        H[0 * 4 + 0] = T(2.0) + d4;  H[0 * 4 + 1] = T(20.0);           H[0 * 4 + 3] = -d4;
        H[1 * 4 + 0] = T(20.0);      H[1 * 4 + 1] = T(200.0) + d3;     H[1 * 4 + 2] = T(-2.0) * d3;
        H[2 * 4 + 1] = T(-2.0) * d3; H[2 * 4 + 2] = T(10.0) + T(4.0) * d3; H[2 * 4 + 3] = T(-10.0);
        H[3 * 4 + 0] = -d4;          H[3 * 4 + 2] = T(-10.0);          H[3 * 4 + 3] = T(10.0) + d4;
      }
      return t1 * t1 + T(5.0) * t2 * t2 + pw(t3, 4) + pw(t4, 4) * T(10.0);
    }
    // residual-vector functions: grad = J^T r, H = J^T J, cost = ||r||^2 (optimize_autodiff.h:151-164)
    T r[3], J[3][2];
    int mr;
    if (fn == 3) {
      mr = 3;
      r[0] = T(1.5) - v0 + v0 * v1; r[1] = T(2.25) - v0 + v0 * v1 * v1; r[2] = T(2.625) - v0 + v0 * v1 * v1 * v1;
      J[0][0] = T(-1) + v1;           J[0][1] = v0;
      J[1][0] = T(-1) + v1 * v1;      J[1][1] = T(2) * v0 * v1;
      J[2][0] = T(-1) + v1 * v1 * v1; J[2][1] = T(3) * v0 * v1 * v1;
    } else {
      mr = 2;
      r[0] = v0 * v0 + v1 - T(11.0); r[1] = v0 + v1 * v1 - T(7.0); r[2] = T(0);
      J[0][0] = T(2) * v0; J[0][1] = T(1);
      J[1][0] = T(1);      J[1][1] = T(2) * v1;
      J[2][0] = J[2][1] = T(0);
    }
    nres = mr;
    if (WANT) {
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        T s = 0;
        for (int i = 0; i < mr; ++i) s += J[i][a] * r[i];
        g[a] = s;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          T q = 0;
          for (int i = 0; i < mr; ++i) q += J[i][a] * J[i][b];
          H[a * 4 + b] = q;
        }
      }
    }
    T c = 0;
    for (int i = 0; i < mr; ++i) c += r[i] * r[i];
    return c;
  }
  __device__ __forceinline__ void accumulate(WaveLds<T>& L, int n, int lane, T& cost, int& nres) {
    T g[4] = {0, 0, 0, 0};
    cost = eval<true>(L, g, nres);
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
        if (a < n) { L.g[a] = g[a]; L.hd[a] = H[a * 4 + a]; }
    }
    wave_sync();
  }
  __device__ __forceinline__ void evaluate(WaveLds<T>& L, int, int, T& cost, int& nres) {
    T g[4];
    cost = eval<false>(L, g, nres);
  }
  template <typename O>
  __device__ __forceinline__ void write_sym(O* M, int LD, int n, int lane) const {
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if (a < n && b < n) M[a * LD + b] = O(H[a * 4 + b]);
    }
  }
};

// sqrt(2):  r = x*x - 2, n = m = 1 (tests/sqrt2.cpp:30-70): grad = J r, H = J^2, cost = r^2 (1 residual).
template <typename T>
struct Sqrt2Model {
  using Scalar = T;
  __device__ __forceinline__ void set_loss(int, double) {}  // no M-estimator on this family
  static constexpr int kXdim = 0;
  __device__ __forceinline__ void plus_eq(WaveLds<T>& L, const T* d, T sign, int, int lane) const { euclid_plus_eq(L, d, sign, lane); }
  static constexpr int kNpad = 16;
  __device__ __forceinline__ void init(int, int, const void*) {}
  __device__ __forceinline__ void bind(long long) {}
  __device__ __forceinline__ void bind_chunk(long long, int, int, int) {}
  __device__ __forceinline__ void accumulate(WaveLds<T>& L, int, int lane, T& cost, int& nres) {
    const T x = L.xs[0];
    const T r = x * x - T(2), J = T(2) * x;
    if (lane == 0) { L.g[0] = J * r; L.hd[0] = J * J; }
    cost = r * r;
    nres = 1;
    wave_sync();
  }
  __device__ __forceinline__ void evaluate(WaveLds<T>& L, int, int, T& cost, int& nres) {
    const T x = L.xs[0];
    const T r = x * x - T(2);
    cost = r * r;
    nres = 1;
  }
  template <typename O>
  __device__ __forceinline__ void write_sym(O* M, int, int, int lane) const {
    if (lane == 0) M[0] = O(0);
  }
};

// ---- manifold policies: how a step is applied to the stored parameters ---------------------------------
template <typename T>
struct EuclidManifold {
  static constexpr int kXdim = 0;
  static __device__ __forceinline__ void plus_eq(WaveLds<T>& L, const T* d, T sign, int, int lane) { euclid_plus_eq(L, d, sign, lane); }
};
template <typename T>
struct Se3Manifold {
  static constexpr int kXdim = 12;
  // pose <- pose * exp(sign * delta): SO3 Rodrigues with small-angle series, SE3 V matrix (Sophus' formulas)
  static __device__ __forceinline__ void plus_eq(WaveLds<T>& L, const T* dv, T sign, int, int lane) {
    T dl[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) dl[i] = sign * dv[i];
    const T wx = dl[3], wy = dl[4], wz = dl[5];
    const T t2 = wx * wx + wy * wy + wz * wz;
    const T th = sqrt(t2);
    T A, B, Cc;
    if (t2 < T(1e-10)) { A = T(1) - t2 / T(6); B = T(0.5) - t2 / T(24); Cc = T(1) / T(6) - t2 / T(120); }
    else { T sn, cs; sincos_t(th, &sn, &cs); A = sn / th; B = (T(1) - cs) / t2; Cc = (th - sn) / (t2 * th); }
    T Rd[9];
    Rd[0] = T(1) - B * (wy * wy + wz * wz); Rd[1] = -A * wz + B * wx * wy;          Rd[2] = A * wy + B * wx * wz;
    Rd[3] = A * wz + B * wx * wy;          Rd[4] = T(1) - B * (wx * wx + wz * wz); Rd[5] = -A * wx + B * wy * wz;
    Rd[6] = -A * wy + B * wx * wz;         Rd[7] = A * wx + B * wy * wz;          Rd[8] = T(1) - B * (wx * wx + wy * wy);
    const T c1[3] = {wy * dl[2] - wz * dl[1], wz * dl[0] - wx * dl[2], wx * dl[1] - wy * dl[0]};
    const T c2[3] = {wy * c1[2] - wz * c1[1], wz * c1[0] - wx * c1[2], wx * c1[1] - wy * c1[0]};
    T td[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) td[i] = dl[i] + B * c1[i] + Cc * c2[i];
    T x[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) x[i] = L.xs[i];
    wave_sync();
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) L.xs[3 * i + j] = x[3 * i] * Rd[j] + x[3 * i + 1] * Rd[3 + j] + x[3 * i + 2] * Rd[6 + j];
        L.xs[9 + i] = x[3 * i] * td[0] + x[3 * i + 1] * td[1] + x[3 * i + 2] * td[2] + x[9 + i];
      }
    }
    wave_sync();
  }
};

// ---- SE3 / SO3 log maps over the scalar type S (plain T for cost-only passes, Jet<T, 6> for differentiated ones), from a
//      rotation MATRIX (the published Sophus formulas; Sophus itself is an un-vendored dependency of the reference):
//        SO3: omega = (theta / sin theta) vee(R - R^T)/2, cos theta = (tr R - 1)/2
//        SE3: upsilon = V^-1 t, V^-1 = I - 1/2 [w]x + (1 - theta cos(theta/2) / (2 sin(theta/2))) / theta^2 [w]x^2
//      Near the identity (cos theta > 0.999) the coefficients come from their power series, smooth there: a square root
//      of a vanishing quantity would make every Jet derivative infinite exactly at the solution of a pose prior.
template <typename T> __device__ __forceinline__ T jet_scalar(const T& x) { return x; }
template <typename T, int N> __device__ __forceinline__ T jet_scalar(const Jet<T, N>& x) { return x.a; }

template <typename S, typename T>
__device__ __forceinline__ void se3_log(const S* R, const S* t, S* xi) {
  const S c = (R[0] + R[4] + R[8] - T(1.0)) * T(0.5);
  const S v[3] = {(R[7] - R[5]) * T(0.5), (R[2] - R[6]) * T(0.5), (R[3] - R[1]) * T(0.5)};  // sin(theta) * axis
  const S s2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  const bool small = jet_scalar(c) > T(0.999);
  S k, coef;
  if (small) {  // asin(x)/x, x = sin(theta)
    k = T(1.0) + s2 * (T(1.0 / 6.0) + s2 * (T(3.0 / 40.0) + s2 * (T(5.0 / 112.0) + s2 * T(35.0 / 1152.0))));
  } else {
    const S sn = sqrt(s2);
    k = atan2(sn, c) / sn;
  }
  const S th2 = s2 * k * k;
  if (small) {  // (1 - (theta/2) cot(theta/2)) / theta^2
    coef = T(1.0 / 12.0) + th2 * (T(1.0 / 720.0) + th2 * (T(1.0 / 30240.0) + th2 * T(1.0 / 1209600.0)));
  } else {
    const S th = sqrt(th2), h = th * T(0.5);
    coef = (T(1.0) - th * cos(h) / (T(2.0) * sin(h))) / th2;
  }
  const S w[3] = {v[0] * k, v[1] * k, v[2] * k};
  const S c1[3] = {w[1] * t[2] - w[2] * t[1], w[2] * t[0] - w[0] * t[2], w[0] * t[1] - w[1] * t[0]};
  const S c2[3] = {w[1] * c1[2] - w[2] * c1[1], w[2] * c1[0] - w[0] * c1[2], w[0] * c1[1] - w[1] * c1[0]};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    xi[i] = t[i] - c1[i] * T(0.5) + coef * c2[i];
    xi[3 + i] = w[i];
  }
}

// SE3 pose prior — the reference's own manifold test (tests/sophus.cpp:26-44): residual(x) = log(prior_inv * x) in R^6,
// differentiated on the device by Jet<T, 6> over the RIGHT perturbation x * exp(delta) at delta = 0, exactly what
// OptimizeWithAutoDiff does for a user type (optimize_autodiff.h:48-77 with sophus.h:24-26): exp(delta) enters the Jets
// through its first-order part I + [omega]x, upsilon (exact for first derivatives at 0).  data: [P][12] = prior_inv
// (R row-major, t); x: [P][12].  One wave per problem; the 6 x 6 system is evaluated redundantly by every lane.
template <typename T>
struct Se3PriorModel {
  using Scalar = T;
  __device__ __forceinline__ void set_loss(int, double) {}  // no M-estimator on this family
  static constexpr int kNpad = 16;
  static constexpr int kXdim = 12;
  const T* data;
  const T* P;
  T G[28];  // upper Gram of [J | r] (7 x 7)
  static __device__ __forceinline__ constexpr int tt(int a, int b) { return a * 7 - a * (a - 1) / 2 + (b - a); }
  __device__ __forceinline__ void init(int, int, const void* dp) { data = static_cast<const T*>(dp); }
  __device__ __forceinline__ void bind(long long p) { P = data + size_t(p) * 12; }
  __device__ __forceinline__ void bind_chunk(long long p, int, int, int) { bind(p); }
  template <typename S>
  __device__ __forceinline__ void residual(const S* Rx, const S* tx, S* xi) const {
    S RA[9], tA[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) RA[3 * i + j] = Rx[j] * P[3 * i] + Rx[3 + j] * P[3 * i + 1] + Rx[6 + j] * P[3 * i + 2];
      tA[i] = tx[0] * P[3 * i] + tx[1] * P[3 * i + 1] + tx[2] * P[3 * i + 2] + P[9 + i];
    }
    se3_log<S, T>(RA, tA, xi);
  }
  __device__ __forceinline__ void accumulate(WaveLds<T>& L, int, int lane, T& cost, int& nres) {
    using J6 = Jet<T, 6>;
    T R[9], t[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = L.xs[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = L.xs[9 + i];
    J6 d[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) d[k] = J6(T(0), k);  // delta = (upsilon, omega) seeded at 0
    J6 Rj[9], tj[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {  // R (I + [omega]x)
      Rj[3 * i + 0] = R[3 * i + 0] + (d[5] * R[3 * i + 1] - d[4] * R[3 * i + 2]);
      Rj[3 * i + 1] = R[3 * i + 1] + (d[3] * R[3 * i + 2] - d[5] * R[3 * i + 0]);
      Rj[3 * i + 2] = R[3 * i + 2] + (d[4] * R[3 * i + 0] - d[3] * R[3 * i + 1]);
      tj[i] = t[i] + (d[0] * R[3 * i] + d[1] * R[3 * i + 1] + d[2] * R[3 * i + 2]);
    }
    J6 xi[6];
    residual<J6>(Rj, tj, xi);
#pragma unroll
    for (int i = 0; i < 28; ++i) G[i] = T(0);
#pragma unroll
    for (int i = 0; i < 6; ++i) {  // fold residual i: w = [J_i | r_i]
      T w[7];
#pragma unroll
      for (int k = 0; k < 6; ++k) w[k] = xi[i].v[k];
      w[6] = xi[i].a;
#pragma unroll
      for (int a = 0; a < 7; ++a)
#pragma unroll
        for (int b = a; b < 7; ++b) G[tt(a, b)] += w[a] * w[b];
    }
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < 6; ++a) { L.g[a] = G[tt(a, 6)]; L.hd[a] = G[tt(a, a)]; }
    }
    cost = G[tt(6, 6)];
    nres = 6;
    wave_sync();
  }
  __device__ __forceinline__ void evaluate(WaveLds<T>& L, int, int, T& cost, int& nres) {
    T R[9], t[3], xi[6];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = L.xs[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = L.xs[9 + i];
    residual<T>(R, t, xi);
    T c = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) c += xi[i] * xi[i];
    cost = c;
    nres = 6;
  }
  template <typename O>
  __device__ __forceinline__ void write_sym(O* M, int LD, int, int lane) const {
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = a; b < 6; ++b) { M[a * LD + b] = O(G[tt(a, b)]); M[b * LD + a] = O(G[tt(a, b)]); }
    }
  }
  __device__ __forceinline__ void plus_eq(WaveLds<T>& L, const T* dv, T sign, int n, int lane) const {
    Se3Manifold<T>::plus_eq(L, dv, sign, n, lane);
  }
};

// SE3 pinhole reprojection: parameters = a pose stored as R (row-major 9) + t (3) = 12 scalars,
// tangent n = 6 in Sophus order (upsilon, omega); residual pair per point r = (f X/Z + cx - u, f Y/Z + cy - v),
// p_c = R p + t; Jacobian w.r.t. the RIGHT perturbation at delta = 0 (what OptimizeWithAutoDiff's user-type
// branch differentiates, optimize_autodiff.h:48-55,73-77): d p_c/d upsilon = R, d p_c/d omega = -R [p]x; update
// pose <- pose * exp(delta) (3rdparty/traits/sophus.h:24-26).  Thread-per-residual evaluation: lane l handles
// points l, l+64, ...; the 7x7 upper Gram of [J | r] (28 values) is accumulated in registers and folded across
// the wave once per pass.  Data per problem: [f cx cy 0 0 0 0 0 | x y z u v ...] (coalesced 5-scalar records).
template <typename T>
struct Se3ReprojModel {
  using Scalar = T;
  __device__ __forceinline__ void set_loss(int, double) {}  // this family carries its loss in the data header
  static constexpr int kNpad = 16;
  static constexpr int kXdim = 12;
  // address_space(1): the data pointer reaches the kernels through a parameter block in memory, so hipcc cannot prove it
  // global and would emit flat loads, which count on the LDS counter too and serialise against the LDS-resident state machine
  using GP = const __attribute__((address_space(1))) T*;
  GP data;
  GP d;
  int npts, pt0, pt1;
  int ninl;  // inlier residuals of the last pass (cost.h:84 NumInliers)
  T G[28];
  static __device__ __forceinline__ constexpr int tt(int a, int b) { return a * 7 - a * (a - 1) / 2 + (b - a); }
  __device__ __forceinline__ void init(int, int m, const void* dp) { npts = m / 2; data = (GP)static_cast<const T*>(dp); }
  __device__ __forceinline__ void bind(long long p) { d = data + size_t(p) * (8 + 5 * size_t(npts)); pt0 = 0; pt1 = npts; }
  __device__ __forceinline__ void bind_chunk(long long p, int row0, int rows, int) {
    d = data + size_t(p) * (8 + 5 * size_t(npts));
    pt0 = row0 / 2;
    pt1 = min(npts, (row0 + rows) / 2);
  }

  template <bool WANT_H>
  __device__ __forceinline__ T pass(const WaveLds<T>& L, int lane) {
    T R[9], t[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = L.xs[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = L.xs[9 + i];
    const T f = d[0], cx = d[1], cy = d[2];
    const int loss = int(d[3]);  // TOA_LOSS_*; 0 = plain squared L2 (wave-uniform)
    const T th2 = d[4];
    if (WANT_H) {
#pragma unroll
      for (int i = 0; i < 28; ++i) G[i] = T(0);
    }
    T csum = 0;
    T inl = 0;  // exact in T: <= 2 * points per lane
    GP pts = d + 8;
    // one point ahead: the next point's five scalars are in flight while this one is folded (a single resident wave
    // per chunk on the row-split path would otherwise pay one HBM round trip per point)
    T nq[5];
    int i = pt0 + lane;
    if (i < pt1) {
#pragma unroll
      for (int k = 0; k < 5; ++k) nq[k] = pts[size_t(i) * 5 + k];
    }
    for (; i < pt1; i += 64) {
      T q[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) q[k] = nq[k];
      if (i + 64 < pt1) {
#pragma unroll
        for (int k = 0; k < 5; ++k) nq[k] = pts[size_t(i + 64) * 5 + k];
      }
      const T px = q[0], py = q[1], pz = q[2];
      const T X = R[0] * px + R[1] * py + R[2] * pz + t[0];
      const T Y = R[3] * px + R[4] * py + R[5] * pz + t[1];
      const T Z = R[6] * px + R[7] * py + R[8] * pz + t[2];
      const T iz = T(1) / Z;
      T w[2][7];
      w[0][6] = f * X * iz + cx - q[3];
      w[1][6] = f * Y * iz + cy - q[4];
      if (WANT_H) {
        const T du0 = f * iz, du2 = -f * X * iz * iz;
        const T dv1 = f * iz, dv2 = -f * Y * iz * iz;
        T D[3][6];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          D[a][0] = R[3 * a]; D[a][1] = R[3 * a + 1]; D[a][2] = R[3 * a + 2];
          D[a][3] = -(R[3 * a + 1] * pz - R[3 * a + 2] * py);
          D[a][4] = -(-R[3 * a] * pz + R[3 * a + 2] * px);
          D[a][5] = -(R[3 * a] * py - R[3 * a + 1] * px);
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          w[0][c] = du0 * D[0][c] + du2 * D[2][c];
          w[1][c] = dv1 * D[1][c] + dv2 * D[2][c];
        }
        if (loss == TOA_LOSS_L2) {
#pragma unroll
          for (int row = 0; row < 2; ++row)
#pragma unroll
            for (int a = 0; a < 7; ++a)
#pragma unroll
              for (int b = a; b < 7; ++b) G[tt(a, b)] += w[row][a] * w[row][b];
        } else {  // M-estimator: cost += l, the point's J^T J and J^T r are scaled by s (robust_norms.h:20-26)
          const T n2 = w[0][6] * w[0][6] + w[1][6] * w[1][6];
          T l, s;
          robust_norm(loss, n2, th2, l, s);
          csum += l;
          inl += n2 <= th2 ? T(2) : T(0);
#pragma unroll
          for (int row = 0; row < 2; ++row)
#pragma unroll
            for (int a = 0; a < 6; ++a) {
              const T sw = s * w[row][a];
#pragma unroll
              for (int b = a; b < 7; ++b) G[tt(a, b)] += sw * w[row][b];
            }
        }
      } else {
        const T n2 = w[0][6] * w[0][6] + w[1][6] * w[1][6];
        if (loss == TOA_LOSS_L2) csum += n2;
        else {
          T l, s;
          robust_norm(loss, n2, th2, l, s);
          csum += l;
          inl += n2 <= th2 ? T(2) : T(0);
        }
      }
    }
    if (loss == TOA_LOSS_L2) ninl = 2 * (pt1 - pt0);
    else ninl = int(wave_allreduce_sum(inl));
    if (WANT_H) {
      wave_allreduce_many(G, lane);
      if (loss == TOA_LOSS_L2) return G[tt(6, 6)];
    }
    return wave_allreduce_sum(csum);
  }
  __device__ __forceinline__ void accumulate(WaveLds<T>& L, int, int lane, T& cost, int& nres) {
    cost = pass<true>(L, lane);
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < 6; ++a) { L.g[a] = G[tt(a, 6)]; L.hd[a] = G[tt(a, a)]; }
    }
    nres = 2 * npts;
    wave_sync();
  }
  __device__ __forceinline__ void evaluate(WaveLds<T>& L, int, int lane, T& cost, int& nres) {
    cost = pass<false>(L, lane);
    nres = 2 * npts;
  }
  template <typename O>
  __device__ __forceinline__ void write_sym(O* M, int LD, int, int lane) const {
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = a; b < 6; ++b) { M[a * LD + b] = O(G[tt(a, b)]); M[b * LD + a] = O(G[tt(a, b)]); }
    }
  }
  __device__ __forceinline__ void plus_eq(WaveLds<T>& L, const T* dv, T sign, int n, int lane) const {
    Se3Manifold<T>::plus_eq(L, dv, sign, n, lane);
  }
};

// ------------------------------------------------------------------------------------------------
// Automatic differentiation on the device: JetModel turns a residual functor written ONCE as a template
// over its scalar type — the form tinyopt users write (`Optimize(x, [](const auto& x) { return r(x); })`,
// docs/API.md:21-35) — into the Accumulate contract, exactly as OptimizeWithAutoDiff does on the host
// (diff/optimize_autodiff.h:91-166): seed x_jet[i].v[i] = 1 (:56-69), evaluate r(x_jet), J.row = r.v, then
// grad = J^T r, H = J^T J, cost = ||r||^2 (:151-164).  Cost-only calls evaluate the SAME functor on plain T
// (no wasted dual arithmetic).  Thread-per-item evaluation with the (kN+1)(kN+2)/2 upper Gram of [J | r] in
// registers, folded across the wave once per pass.
//
// Functor concept (all static):  kN parameters (<= 12), kR residuals per item, kD data scalars per item,
//   kH header scalars per problem;  template <class S> static void eval(const S* x, const T* header,
//   const T* item, S* r).   Data per problem: [kH | items x kD].
// A user family = one functor + one line in inst.hip (INTEGRATION.md "bring your own functor").
//
// MANIFOLD (round 4; TOA_MANIFOLD_*): 0 = Euclidean parameters, x (+)= dx (traits.h:184-190).  1 = ONE SE3 pose stored as
//   R (row-major 9) + t (3) = 12 scalars, tangent kN = 6 in Sophus order (upsilon, omega): the functor sees the pose through
//   x[0..11] — Jets seeded over the RIGHT perturbation x * exp(delta) at delta = 0, what OptimizeWithAutoDiff does for a user
//   type (optimize_autodiff.h:48-77 with 3rdparty/traits/sophus.h:13-27; tests/sophus.cpp:26-44 `Optimize(pose, lambda)`);
//   the update is pose <- pose * exp(delta).
// A functor with `kManual = true` is a manual Accumulate callback instead (docs/API.md:37-57, tests/optimize_easy.cpp:35-79:
//   the user writes the Jacobian rows, no AD):  template <bool WANT_GRAD> eval_manual(const T* x, header, item, T* r, T (*J)[kN]).
// ------------------------------------------------------------------------------------------------
template <typename F, typename = void>
struct FunctorManual { static constexpr bool value = false; };
template <typename F>
struct FunctorManual<F, std::enable_if_t<F::kManual>> { static constexpr bool value = true; };

// the pose as Jet<T, 6> over the right perturbation at delta = 0: R (I + [omega]x), t + R upsilon (exact to first order)
template <typename T>
__device__ __forceinline__ void se3_seed_pose(const T* x, Jet<T, 6>* xj) {
  using J6 = Jet<T, 6>;
  J6 d[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) d[k] = J6(T(0), k);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    xj[3 * i + 0] = x[3 * i + 0] + (d[5] * x[3 * i + 1] - d[4] * x[3 * i + 2]);
    xj[3 * i + 1] = x[3 * i + 1] + (d[3] * x[3 * i + 2] - d[5] * x[3 * i + 0]);
    xj[3 * i + 2] = x[3 * i + 2] + (d[4] * x[3 * i + 0] - d[3] * x[3 * i + 1]);
    xj[9 + i] = x[9 + i] + (d[0] * x[3 * i] + d[1] * x[3 * i + 1] + d[2] * x[3 * i + 2]);
  }
}

// MANIFOLD == 2: a USER manifold (run-time models, TOA_MANIFOLD_USER; the reference's extension point traits::params_trait<T>,
// traits.h:103-359 — e.g. 3rdparty/traits/lieplusplus.h).  The functor carries the parameter container's size kX (scalars as
// stored) and ONE function, written over the scalar type like the residual:
//     template <class S> static void plus(const T* x, const S* d, S* xp)      xp = x (+) d,  d in the kN-dimensional tangent
// from which both uses follow: the update x <- x (+) (+-delta) on plain T (PlusEq, traits.h:184-190) and the differentiation —
// the residual is evaluated on xp = plus(x, Jets seeded on d at d = 0), exactly optimize_autodiff.h:48-77.
template <typename F, typename = void>
struct FunctorX { static constexpr int value = F::kN; };
template <typename F>
struct FunctorX<F, std::enable_if_t<(F::kX > 0)>> { static constexpr int value = F::kX; };
template <typename T, typename F>
struct UserManifoldOf {
  static constexpr int kXdim = FunctorX<F>::value;
  static __device__ __forceinline__ void plus_eq(WaveLds<T>& L, const T* dv, T sign, int, int lane) {
    T xo[kXdim], dd[F::kN], xn[kXdim];
#pragma unroll
    for (int i = 0; i < kXdim; ++i) xo[i] = L.xs[i];
#pragma unroll
    for (int a = 0; a < F::kN; ++a) dd[a] = sign * dv[a];
    F::template plus<T>(xo, dd, xn);
    wave_sync();
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < kXdim; ++i) L.xs[i] = xn[i];
    }
    wave_sync();
  }
};

template <typename T, typename F, int MANIFOLD = 0>
struct JetModel {
  using Scalar = T;
  static constexpr int kNpad = 16;
  static constexpr int kXdim = MANIFOLD == 1 ? 12 : (MANIFOLD == 2 ? FunctorX<F>::value : 0);
  static constexpr int kN = F::kN, kW = F::kN + 1, kG = kW * (kW + 1) / 2;
  static constexpr int kX = MANIFOLD == 1 ? 12 : (MANIFOLD == 2 ? FunctorX<F>::value : F::kN);   // stored scalars of x
  static constexpr bool kManual = FunctorManual<F>::value;
  static_assert(F::kN >= 1 && F::kN <= 12, "register Gram: kN <= 12");
  static_assert(MANIFOLD != 1 || F::kN == 6, "an SE3 pose has a 6-dimensional tangent");
  static_assert(kX <= 32, "stored scalars of x");
  const T* data;
  const T* d;
  int items, it0, it1;
  int loss;   // TOA_LOSS_* on each ITEM's squared residual norm (toa_set_loss; robust_norms.h:20-26); 0 = plain L2
  T th2;
  int ninl;   // inlier residuals of the last pass
  T G[kG];
  static __device__ __forceinline__ constexpr int tt(int a, int b) { return a * kW - a * (a - 1) / 2 + (b - a); }
  __device__ __forceinline__ void init(int, int m, const void* dp) {
    items = m / F::kR; data = static_cast<const T*>(dp);
    loss = TOA_LOSS_L2; th2 = T(0); ninl = -1;
  }
  __device__ __forceinline__ void set_loss(int kind, double t2) { loss = kind; th2 = T(t2); }
  __device__ __forceinline__ void bind(long long p) {
    d = data + size_t(p) * (F::kH + size_t(items) * F::kD);
    it0 = 0; it1 = items;
  }
  // rows [row0, row0 + rows) of the problem = whole items (the launchers cut chunks at multiples of kR rows): the row-split form
  __device__ __forceinline__ void bind_chunk(long long p, int row0, int rows, int) {
    bind(p);
    it0 = row0 / F::kR;
    it1 = min(items, (row0 + rows) / F::kR);
  }
  __device__ __forceinline__ void plus_eq(WaveLds<T>& L, const T* dv, T sign, int n, int lane) const {
    if constexpr (MANIFOLD == 1) Se3Manifold<T>::plus_eq(L, dv, sign, n, lane);
    else if constexpr (MANIFOLD == 2) UserManifoldOf<T, F>::plus_eq(L, dv, sign, n, lane);
    else euclid_plus_eq(L, dv, sign, lane);
  }

  template <bool WANT_H>
  __device__ __forceinline__ T pass(const WaveLds<T>& L, int lane) {
    T x[kX];
#pragma unroll
    for (int i = 0; i < kX; ++i) x[i] = L.xs[i];
    if (WANT_H) {
#pragma unroll
      for (int i = 0; i < kG; ++i) G[i] = T(0);
    }
    T csum = 0;
    T inl = 0;
    const bool robust = loss != TOA_LOSS_L2;   // wave-uniform
    const T* itemsp = d + F::kH;
    for (int i = it0 + lane; i < it1; i += 64) {
      const T* item = itemsp + size_t(i) * F::kD;
      if (WANT_H) {
        T rv[F::kR], Jv[F::kR][kN];   // residuals and their Jacobian rows
        if constexpr (kManual) {
          F::template eval_manual<true>(x, d, item, rv, Jv);          // the user's own derivatives (docs/API.md:37-57)
        } else {
          Jet<T, kN> xj[kX], r[F::kR];
          if constexpr (MANIFOLD == 1) {
            se3_seed_pose<T>(x, xj);                                  // optimize_autodiff.h:48-55, 73-77
          } else if constexpr (MANIFOLD == 2) {
            Jet<T, kN> dj[kN];                                        // x (+) delta over Jets seeded on delta at delta = 0
#pragma unroll
            for (int k = 0; k < kN; ++k) dj[k] = Jet<T, kN>(T(0), k);
            F::template plus<Jet<T, kN>>(x, dj, xj);
          } else {
#pragma unroll
            for (int k = 0; k < kN; ++k) xj[k] = Jet<T, kN>(x[k], k);   // optimize_autodiff.h:56-69
          }
          F::template eval<Jet<T, kN>>(xj, d, item, r);
#pragma unroll
          for (int q = 0; q < F::kR; ++q) {
            rv[q] = r[q].a;
#pragma unroll
            for (int a = 0; a < kN; ++a) Jv[q][a] = r[q].v[a];        // J.row(i) = res[i].v   (:127-148)
          }
        }
        T s = T(1);
        if (robust) {   // the item's ||r||^2 through the M-estimator: cost += l, its J^T J and J^T r scaled by s
          T n2 = 0, l;
#pragma unroll
          for (int q = 0; q < F::kR; ++q) n2 += rv[q] * rv[q];
          robust_norm(loss, n2, th2, l, s);
          csum += l;
          inl += n2 <= th2 ? T(F::kR) : T(0);
        }
#pragma unroll
        for (int q = 0; q < F::kR; ++q) {
          T w[kW];
#pragma unroll
          for (int a = 0; a < kN; ++a) w[a] = Jv[q][a];
          w[kN] = rv[q];
#pragma unroll
          for (int a = 0; a < kW; ++a) {
            const T sw = s * w[a];
#pragma unroll
            for (int b = a; b < kW; ++b) G[tt(a, b)] += sw * w[b];
          }
        }
      } else {
        T r[F::kR];
        if constexpr (kManual) F::template eval_manual<false>(x, d, item, r, static_cast<T(*)[kN]>(nullptr));
        else F::template eval<T>(x, d, item, r);
        T n2 = 0;
#pragma unroll
        for (int q = 0; q < F::kR; ++q) n2 += r[q] * r[q];
        if (robust) {
          T l, s;
          robust_norm(loss, n2, th2, l, s);
          csum += l;
          inl += n2 <= th2 ? T(F::kR) : T(0);
        } else {
          csum += n2;
        }
      }
    }
    ninl = robust ? int(wave_allreduce_sum(inl)) : -1;
    if (WANT_H) {
      wave_allreduce_many(G, lane);
      if (!robust) return G[tt(kN, kN)];
    }
    return wave_allreduce_sum(csum);
  }
  __device__ __forceinline__ void accumulate(WaveLds<T>& L, int, int lane, T& cost, int& nres) {
    cost = pass<true>(L, lane);
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < kN; ++a) { L.g[a] = G[tt(a, kN)]; L.hd[a] = G[tt(a, a)]; }
    }
    nres = items * F::kR;
    wave_sync();
  }
  __device__ __forceinline__ void evaluate(WaveLds<T>& L, int, int lane, T& cost, int& nres) {
    cost = pass<false>(L, lane);
    nres = items * F::kR;
  }
  template <typename O>
  __device__ __forceinline__ void write_sym(O* M, int LD, int, int lane) const {
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < kN; ++a)
#pragma unroll
        for (int b = a; b < kN; ++b) { M[a * LD + b] = O(G[tt(a, b)]); M[b * LD + a] = O(G[tt(a, b)]); }
    }
  }
};

// tests/circle.cpp:32-68: x = (cx, cy, radius); one residual per observed point p: ||p - c||^2 - radius^2
template <typename T>
struct CircleFitFunctor {
  static constexpr int kN = 3, kR = 1, kD = 2, kH = 0;
  template <class S, class X>
  static __device__ __forceinline__ void eval(const X& x, const T*, const T* p, S* r) {
    const S dx = p[0] - x[0];
    const S dy = p[1] - x[1];
    r[0] = dx * dx + dy * dy - x[2] * x[2];
  }
};
// The DenseRow residual written the tinyopt way (no hand-derived Jacobian): item = [a_0 .. a_{N-1}, b].
// Exists to cross-check the AD machinery (Jet sin / products) against the analytic MFMA path.
template <typename T, int NN>
struct DenseRowAdFunctor {
  static constexpr int kN = NN, kR = 1, kD = NN + 1, kH = 0;
  template <class S, class X>
  static __device__ __forceinline__ void eval(const X& x, const T*, const T* item, S* r) {
    S t = x[0] * item[0];
    // wide blocks: a rolled loop (fully unrolled, 50 seeded Jets are live at once: 370 VGPRs)
    constexpr int kUnroll = NN <= 12 ? NN : 2;
#pragma unroll kUnroll
    for (int j = 1; j < NN; ++j) t = t + x[j] * item[j];
    r[0] = t + T(0.1) * sin(t) - item[NN];
  }
};

template <typename F, typename = void>
struct FunctorComputeBound { static constexpr bool value = false; };
template <typename F>
struct FunctorComputeBound<F, std::enable_if_t<F::kComputeBound>> { static constexpr bool value = true; };
template <typename F, typename = void>
struct FunctorIndexed { static constexpr bool value = false; };
template <typename F>
struct FunctorIndexed<F, std::enable_if_t<F::kIndexedOperands>> { static constexpr bool value = true; };
// Row models for WIDE parameter blocks (13 <= kN <= 63): a row is a lane — the user's Jacobian rows, or chunked Jets, staged
// through LDS into the matrix cores' operand layout (row_model.hpp; SURVEY §8f rank 1, optimize_autodiff.h:91-166).
}  // namespace toa
#include "row_model.hpp"
namespace toa {

// Per-problem LM state parked in HBM between launches: the stepping form (`Optimizer_::Step`, optimizer.h:331-539, one
// loop pass per call) and the launch-per-iteration row-split path both resume the same state machine from it.
template <typename T>
struct WideState {
  LmState<T> st;
  T xs[64], g[64], hd[64], dx[64], ldx[64];
};

template <typename T>
__device__ __forceinline__ void wide_load_state(WaveLds<T>& L, const WideState<T>* ws, int lane) {
  const int* src = reinterpret_cast<const int*>(&ws->st);
  int* dst = reinterpret_cast<int*>(L.st);
  for (int i = lane; i < int(sizeof(LmState<T>) / 4); i += 64) dst[i] = src[i];
  L.xs[lane] = ws->xs[lane]; L.g[lane] = ws->g[lane]; L.hd[lane] = ws->hd[lane];
  L.dx[lane] = ws->dx[lane]; L.ldx[lane] = ws->ldx[lane];
  wave_sync();
}
template <typename T>
__device__ __forceinline__ void wide_store_state(const WaveLds<T>& L, WideState<T>* ws, int lane) {
  wave_sync();
  const int* src = reinterpret_cast<const int*>(L.st);
  int* dst = reinterpret_cast<int*>(&ws->st);
  for (int i = lane; i < int(sizeof(LmState<T>) / 4); i += 64) dst[i] = src[i];
  ws->xs[lane] = L.xs[lane]; ws->g[lane] = L.g[lane]; ws->hd[lane] = L.hd[lane];
  ws->dx[lane] = L.dx[lane]; ws->ldx[lane] = L.ldx[lane];
}

struct FusedParams {
  const void* data;
  void* x;
  long long P;
  int n, m;
  toa_options opt;
  toa_results res;
  unsigned long long* counters;  // [4] or null
  int* queue;                    // [0] pop counter, [16] waves that have left the kernel (separate cache lines)
  unsigned long long* timeline;  // debug (toa_debug_timeline): [P][2] start / end of every problem in 100 MHz ticks
  int lds_per_wave;
  int mode;                      // 0: whole solve; 1: begin (state <- x0, lm_init); 2: ONE loop pass per problem (stepping form);
                                 // 3: finalise the problems named in stop_request with that StopReason (host-side stop controls)
  void* state;                   // modes 1, 2, 3: caller's state block (see launch_wide)
  int* active;                   // mode 2 (optional): += 1 per problem that is still running after this pass
  const int* stop_request;       // mode 3: [P] StopReason to impose on a still-running problem (0 = leave it running)
  int loss;                      // TOA_LOSS_* of the handle (toa_set_loss): applied per residual by the DenseRow / Jet families
  double loss_th2;
  void* memo;                    // mode 0, models with kMemo: one slot of memo_stride bytes per resident wave (null = off)
  unsigned long long memo_stride;
  int stage_off;                 // row-per-lane fp64 pass: byte offset of the wave's LDS stage in its LDS region
  int carve_off;                 // byte offset of the wave's carve in its LDS region (> 0: a stage in front of it, overlapping the carve's
                                 // pass-dead head — WaveLds::pass_dead_bytes)
  int coop_tot_off;              // cooperative passes: byte offset of the chunk-partial total in a wave's carve (0 = its M)
  int memo_lds_off;              // != 0: the memo slot is in LDS instead, at this byte offset of the wave's carve (small Grams)
  int coop_K;                    // cooperative passes (CoopCtl): chunks per pass, 0 = off
  int coop_cs;                   // steps (of 4 rows) per chunk, a multiple of the load ring's period
  int reserved_[2];              // (the team form's two fields, round 5: the block's layout is unchanged)
};

template <typename M, typename = void>
struct ModelStageBytes { static constexpr size_t value = 0; };
template <typename M>
struct ModelStageBytes<M, std::enable_if_t<(M::kStageBytes > 0)>> { static constexpr size_t value = M::kStageBytes; };
// A staged model's LDS stage begins at the wave's region and ends INSIDE its carve, over the part of it that is dead while a
// pass runs (WaveLds::pass_dead_bytes): the carve starts stage_carve_off bytes into the region.  One rule for every kernel and
// for the host's LDS sizing (lds_fit): a function of the stage size and n only.
template <typename T>
__host__ __device__ inline size_t stage_carve_off(size_t stage_bytes, int n) {
  const size_t st = (stage_bytes + 15) & ~size_t(15), dead = WaveLds<T>::pass_dead_bytes(n);
  return st > dead ? st - dead : 0;
}
// launch-per-iteration kernels: binds the model's stage, returns where the wave's carve starts
template <typename Model>
__device__ __forceinline__ char* model_bind_stage(Model& model, char* wave_base, int n) {
  if constexpr (ModelStageBytes<Model>::value > 0) {
    model.stage = reinterpret_cast<unsigned char*>(wave_base);
    return wave_base + stage_carve_off<typename Model::Scalar>(ModelStageBytes<Model>::value, n);
  } else {
    (void)model; (void)n;
    return wave_base;
  }
}
template <typename M, typename = void>
struct ModelWaves { static constexpr int value = 4; };
template <typename M>
struct ModelWaves<M, std::enable_if_t<(M::kWaves > 0)>> { static constexpr int value = M::kWaves; };
template <typename M, typename = void>
struct ModelCoop { static constexpr bool value = false; };
template <typename M>
struct ModelCoop<M, std::enable_if_t<M::kCoop>> { static constexpr bool value = true; };

// (An occupancy request via __launch_bounds__'s second argument is NOT usable here: under the tighter register budget
// hipcc parks the destination registers of the in-flight asm loads in AGPRs right after issuing them — tools/isa_lint.py
// caught exactly that when 5 waves/SIMD were requested for the fp64 n <= 15 kernel.)
template <typename Model>
#ifndef TOA_FUSED_ATTR
#define TOA_FUSED_ATTR   // run-time builds may ask for an occupancy here (jit.hip: __attribute__((amdgpu_waves_per_eu(3, 3))))
#endif
__global__ void __launch_bounds__(64 * ModelWaves<Model>::value) TOA_FUSED_ATTR lm_fused_kernel(const FusedParams* __restrict__ prm_g) {
  using T = typename Model::Scalar;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int kW = ModelWaves<Model>::value;   // waves per workgroup
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int n = prm_g->n;
  constexpr int NO = kW;                         // every wave of a workgroup pulls problems
  constexpr bool owner = true;
  WaveLds<T> L = WaveLds<T>::carve(smem + size_t(wave) * prm_g->lds_per_wave + prm_g->carve_off, n);
  // private per-wave copies of the option / result PODs (no inter-wave synchronisation anywhere)
  {
    const int* src_o = reinterpret_cast<const int*>(&prm_g->opt);
    int* dst_o = reinterpret_cast<int*>(L.opt);
    for (int i = lane; i < int(sizeof(toa_options) / 4); i += 64) dst_o[i] = src_o[i];
    const int* src_r = reinterpret_cast<const int*>(&prm_g->res);
    int* dst_r = reinterpret_cast<int*>(L.res);
    for (int i = lane; i < int(sizeof(toa_results) / 4); i += 64) dst_r[i] = src_r[i];
    L.st->acc_passes = 0; L.st->eval_passes = 0; L.st->solves = 0; L.st->problems = 0; L.st->reused_passes = 0;
    L.st->memo_slot = 0;
    if constexpr (ModelMemo<Model>::value) {
      if (prm_g->memo_lds_off)   // (a generic pointer into LDS: the flat stores / loads of memo_save / memo_load reach it too)
        L.st->memo_slot = reinterpret_cast<unsigned long long>(static_cast<void*>(smem + size_t(wave) * prm_g->lds_per_wave + prm_g->memo_lds_off));
      else if (prm_g->memo)
        L.st->memo_slot = owner ? reinterpret_cast<unsigned long long>(prm_g->memo) + (size_t(blockIdx.x) * NO + wave) * prm_g->memo_stride : 0ull;
    }
  }
  wave_sync();
  const long long P = prm_g->P;
  Model model;
  model.init(n, prm_g->m, prm_g->data);
  model.set_loss(prm_g->loss, prm_g->loss_th2);
  if constexpr (ModelStageBytes<Model>::value > 0) model.stage = reinterpret_cast<unsigned char*>(smem) + size_t(wave) * prm_g->lds_per_wave + prm_g->stage_off;
  T* X = static_cast<T*>(prm_g->x);
  const int xd = Model::kXdim ? Model::kXdim : n;  // stored parameters per problem (SE3: 12 for n = 6)
  int* queue = prm_g->queue;
  if constexpr (ModelCoop<Model>::value) {   // the workgroup's control block: every wave marks itself as an owner, no pass open
    model.coop_init(prm_g->coop_K, prm_g->coop_cs, prm_g->lds_per_wave, prm_g->coop_tot_off);
    CoopCtl* ctl = reinterpret_cast<CoopCtl*>(smem + size_t(kW) * prm_g->lds_per_wave);
    if (lane == 0) {
      ctl->slot[wave].ticket = prm_g->coop_K;
      ctl->slot[wave].turn = prm_g->coop_K;
      ctl->active[wave] = owner ? 1 : 0;
    }
    __syncthreads();   // the only workgroup barrier of the kernel: nobody scans the slots before they exist
  }
  int solved = 0;
  // The first problem of every wave is assigned statically (wave w of the launch takes problem w); the shared counter hands
  // out the rest.  4 096 waves popping the same address at launch time serialise in the L2 (~5 ns per atomic = 20-30 us
  // before the last wave has its first problem: 4 % of a C3 launch, visible in the launch timeline).
  const int nwaves = int(gridDim.x) * NO;   // (owners of the launch)
  bool first = owner, dry = !owner;           // (a helper of the team form starts where an owner ends up: queue dry, looking for tickets)
  for (;;) {  // one work item = one whole problem
    int p = 0;
    if (first) {
      p = int(blockIdx.x) * NO + wave;
      first = false;
    } else if (dry) {
      p = int(P);
    } else {
      // (An "end game" that stops two — or three — waves of every workgroup from pulling once fewer than two problems per
      //  workgroup are left, so that each remaining problem is worked by an owner plus helpers, was measured and rejected: C4
      //  12.93 -> 12.60 M it/s with two pullers, 12.07 with one (profiles/r03_ab_log.md).  A helper shares the Accumulate
      //  passes only; the solve, the step test and the evaluate-only passes stay with the owner, and two waves on one problem
      //  are well short of twice as fast.)
      if (lane == 0) p = atomicAdd(queue, 1) + nwaves;
      p = __builtin_amdgcn_readfirstlane(p);
    }
    bool ghost = false;
    if (p >= P) {
      // The queue is dry.  A wave of a cooperative model does not leave yet: it looks for a sibling's open Accumulate pass,
      // takes a chunk ticket of it (DenseRowModel::coop_find) and runs a GHOST problem through the very same state-machine
      // code — whose single Accumulate call is where the chunk loop lives (DenseRowModel::coop_acc: hipcc tolerates
      // exactly one MFMA loop per kernel).  That call works the ticket (and the pass's remaining ones) off, then reports
      // "no residuals", which ends the ghost at once (kSkipped, optimizer.h:372-375) with nothing written anywhere (p < 0);
      // the wave comes back here for the next ticket until no sibling is active any more.
      dry = true;
      if constexpr (ModelCoop<Model>::value) {
        if (model.coop_K > 1) ghost = model.coop_find(lane);
      }
      if (!ghost) break;
    }
    // Fairness between the waves of a SIMD.  The issue arbiter serves the OLDEST wave first, and a wave keeps its age for
    // the whole (persistent) kernel: the launch timeline shows the oldest wave of each SIMD solving a problem in 0.8 ms
    // while the youngest needs up to 6.9 ms for its first one and is still far from done when the queue runs dry — the
    // drain is then as long as those starved problems.  Priority outranks age, so the waves that are behind are given
    // the issue slots: a wave drops one level per problem it has finished.  (Measured and rejected, profiles/r02_ab_log.md:
    // no priorities; a level that follows the lag behind the average wave; re-queueing unfinished problems iteration by
    // iteration through HBM during the drain.)
    {
      const int lag = 1 - solved;
      if (lag >= 1) __builtin_amdgcn_s_setprio(3);
      else if (lag == 0) __builtin_amdgcn_s_setprio(2);
      else if (lag == -1) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
    ++solved;
    if (ghost) p = 0;
    model.bind(p);
    wave_sync();
    L.xs[lane] = (lane < xd && !ghost) ? X[size_t(p) * xd + lane] : T(0);
    wave_sync();
    const unsigned long long tl0 = prm_g->timeline ? wall_clock64() : 0ull;
    lm_solve_problem<T>(model, L, n, lane, ghost ? -1ll : (long long)p);
    if (ghost) {
      if (lane == 0) L.st->acc_passes -= 1;   // the ghost's Build streamed nothing of its own
      continue;
    }
    if (lane < xd) X[size_t(p) * xd + lane] = L.xs[lane];
    if (prm_g->timeline && lane == 0) { prm_g->timeline[2 * size_t(p)] = tl0; prm_g->timeline[2 * size_t(p) + 1] = wall_clock64(); }
  }
  unsigned long long* counters = prm_g->counters;
  if (counters && lane == 0) {
    atomicAdd(&counters[0], L.st->acc_passes);
    atomicAdd(&counters[1], L.st->eval_passes);
    atomicAdd(&counters[2], L.st->solves);
    atomicAdd(&counters[3], L.st->problems);
    if (L.st->reused_passes) atomicAdd(&counters[4], L.st->reused_passes);
  }
  // The work queue cleans itself: the last wave to leave puts the pop counter (and this exit counter) back to zero, so the
  // next launch on the stream needs no memset in front of it (one stream operation, ~5 us, per solve: 1 % of a C3 launch).
  if (lane == 0) {
    const int gone = atomicAdd(&queue[16], 1);
    if (gone == int(gridDim.x) * kW - 1) {
      __hip_atomic_store(&queue[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&queue[16], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// K1/K2 seam: one wave per problem (grid-stride), writes g [P][n], H [P][n*n], cost, nres.
template <typename Model>
__global__ void __launch_bounds__(256) accumulate_kernel(const void* data_, const void* x_, long long P, int n, int m,
                                                         int want_grad, void* g_, void* H_, double* cost, int* nres,
                                                         int lds_per_wave, int loss, double loss_th2) {
  using T = typename Model::Scalar;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T* X = static_cast<const T*>(x_);
  Model model;
  model.init(n, m, data_);
  model.set_loss(loss, loss_th2);
  WaveLds<T> L = WaveLds<T>::carve(model_bind_stage(model, smem + size_t(wave) * lds_per_wave, n), n);
  const int xd = Model::kXdim ? Model::kXdim : n;
  for (long long p = (long long)blockIdx.x * 4 + wave; p < P; p += (long long)gridDim.x * 4) {
    wave_sync();
    L.xs[lane] = lane < xd ? X[size_t(p) * xd + lane] : T(0);
    wave_sync();
    model.bind(p);
    T c;
    int nr;
    if (want_grad) {
      model.accumulate(L, n, lane, c, nr);
      T* G = static_cast<T*>(g_) + size_t(p) * n;
      T* H = static_cast<T*>(H_) + size_t(p) * n * n;
      if (lane < n) G[lane] = L.g[lane];
      model.write_sym(H, n, n, lane);
      wave_sync();
      if (lane < n) H[lane * n + lane] = L.hd[lane];  // the (undamped) diagonal always comes from hd
    } else {
      model.evaluate(L, n, lane, c, nr);
    }
    if (lane == 0) { cost[p] = double(c); if (nres) nres[p] = nr; }
  }
}

// K3 seam: H_ii *= scale (double), dx = -H^-1 g with Eigen's acceptance rule.
template <typename T, int NPAD>
__global__ void __launch_bounds__(256) solve_damped_kernel(const void* H_, const void* g_, long long P, int n,
                                                           double scale, void* dx_, int* ok_, int lds_per_wave) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  WaveLds<T> L = WaveLds<T>::carve(smem + size_t(wave) * lds_per_wave, n);
  const T* Hg = static_cast<const T*>(H_);
  const T* gg = static_cast<const T*>(g_);
  T* dxg = static_cast<T*>(dx_);
  for (long long p = (long long)blockIdx.x * 4 + wave; p < P; p += (long long)gridDim.x * 4) {
    wave_sync();
    const T* H = Hg + size_t(p) * n * n;
    // upper triangle is authoritative (math.h:235 selfadjointView<Upper>): M[i][j] = H(min,max) (col-major)
    auto fill = [&]() __attribute__((always_inline)) {
      for (int e = lane; e < n * n; e += 64) {
        const int i = e / n, j = e % n;
        const int a = i < j ? i : j, b = i < j ? j : i;
        T v = H[size_t(b) * n + a];
        if (i == j) v = T(double(v) * scale);
        L.M[i * L.LD + j] = v;
      }
      wave_sync();
    };
    fill();
    const T gl = lane < n ? gg[size_t(p) * n + lane] : T(0);
    bool ok;
    T dx = 0;
    {
      LdltFast<T, NPAD> F;
      ok = F.factor(L.M, L.LD, n, lane);
      if (ok) dx = F.solve(L.M, L.LD, n, lane, -gl);
      else if (LdltFast<T, NPAD>::kClobbersM) { wave_sync(); fill(); }
    }
    if (!ok) {
      ok = ldlt_factor_wave<T>(L.M, L.LD, L.perm, L.tmp, n, lane);
      if (ok) dx = ldlt_solve_wave<T>(L.M, L.LD, L.perm, L.vec, n, lane, -gl);
    }
    if (lane < n) dxg[size_t(p) * n + lane] = dx;
    if (lane == 0) ok_[p] = ok ? 1 : 0;
  }
}

// Covariance seam: C = H^-1 by LDL^T against the identity — tinyopt::InvCov / DenseInvCov (include/tinyopt/math.h:41-57:
// `chol = m.selfadjointView<Upper>().ldlt(); if (Success && isPositive()) return chol.solve(Identity)`; cols()==1:
// unprotected 1/m), used by Output::Covariance (output.h:80-94) and SolverLM::Covariance (lm.h:174).
template <typename T, int NPAD>
__global__ void __launch_bounds__(256) inv_cov_kernel(const void* H_, long long P, int n, void* C_, int* ok_, int lds_per_wave) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  WaveLds<T> L = WaveLds<T>::carve(smem + size_t(wave) * lds_per_wave, n);
  const T* Hg = static_cast<const T*>(H_);
  T* Cg = static_cast<T*>(C_);
  for (long long p = (long long)blockIdx.x * 4 + wave; p < P; p += (long long)gridDim.x * 4) {
    wave_sync();
    const T* H = Hg + size_t(p) * n * n;
    T* C = Cg + size_t(p) * n * n;
    if (n == 1) {  // math.h:49-50
      if (lane == 0) { C[0] = T(1) / H[0]; ok_[p] = 1; }
      continue;
    }
    auto fill = [&]() __attribute__((always_inline)) {
      for (int e = lane; e < n * n; e += 64) {
        const int i = e / n, j = e % n;
        const int a = i < j ? i : j, b = i < j ? j : i;
        L.M[i * L.LD + j] = H[size_t(b) * n + a];  // upper triangle is authoritative
      }
      wave_sync();
    };
    fill();
    LdltFast<T, NPAD> F;
    bool ok = F.factor(L.M, L.LD, n, lane);
    if (ok) {
      for (int j = 0; j < n; ++j) {
        const T x = F.solve(L.M, L.LD, n, lane, lane == j ? T(1) : T(0));
        if (lane < n) C[size_t(j) * n + lane] = x;  // column j (symmetric: row j)
      }
    } else {
      if (LdltFast<T, NPAD>::kClobbersM) { wave_sync(); fill(); }
      ok = ldlt_factor_wave<T>(L.M, L.LD, L.perm, L.tmp, n, lane);
      if (ok)
        for (int j = 0; j < n; ++j) {
          const T x = ldlt_solve_wave<T>(L.M, L.LD, L.perm, L.vec, n, lane, lane == j ? T(1) : T(0));
          if (lane < n) C[size_t(j) * n + lane] = x;
        }
    }
    if (!ok)  // rejected (std::nullopt in the reference): define the output instead of leaving caller memory untouched
      for (int e = lane; e < n * n; e += 64) C[e] = T(0);
    if (lane == 0) ok_[p] = ok ? 1 : 0;
  }
}

// ================================================================================================
// Row-split ("wide") execution for few, huge problems (BASELINE configs C2 / C5: P = 1, m = 10^3..5*10^4).
// One wavefront per problem would leave the chip idle, so the rows of every problem are split over S chunks:
//   wide_partial_kernel  (P*S waves)  K1/K2 on one chunk each -> partial (H, g, cost) in HBM scratch
//   wide_step_kernel     (P waves)    sums the S partials in a fixed order (deterministic), then runs ONE
//                                     iteration of the same state machine (lm_iteration) with the state parked
//                                     in global memory between launches
// The host enqueues init + (partial, step) x max_iters on the stream without reading anything back: problems
// that have stopped make their later launches no-ops.
// ================================================================================================
struct WideParams {
  const void* data;
  void* x;
  long long P;
  int n, m, splits, chunk_rows;
  toa_options opt;
  toa_results res;
  unsigned long long* counters;
  void* state;     // WideState<T>[P]
  void* partials;  // T[P][splits][n*n + n + 2]  (H, g, cost, inlier residuals)
  void* hsum;      // T[P][n*n]
  int step_mode;   // stepping form: publish x and the running results at every pass
  int* active;     // stepping form (optional): += 1 per problem still running after the pass
  const int* stop_request;  // stepping form, wide_stop_kernel: [P] StopReason to impose (0 = none)
  int loss;                 // toa_set_loss
  double loss_th2;
  unsigned* sync;  // persistent form: [P][2] = (arrive, go) generation counters, then [1] abort flag; zeroed per launch
  int lds_per_wave;
};

template <typename T>
__device__ __forceinline__ void wide_copy_pods(WaveLds<T>& L, const WideParams* prm, int lane) {
  const int* src_o = reinterpret_cast<const int*>(&prm->opt);
  int* dst_o = reinterpret_cast<int*>(L.opt);
  for (int i = lane; i < int(sizeof(toa_options) / 4); i += 64) dst_o[i] = src_o[i];
  const int* src_r = reinterpret_cast<const int*>(&prm->res);
  int* dst_r = reinterpret_cast<int*>(L.res);
  for (int i = lane; i < int(sizeof(toa_results) / 4); i += 64) dst_r[i] = src_r[i];
}
template <typename T, int XD>
__global__ void __launch_bounds__(256) wide_init_kernel(const WideParams* __restrict__ prm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long p = (long long)blockIdx.x * 4 + wave;
  if (p >= prm->P) return;
  const int n = prm->n;
  WaveLds<T> L = WaveLds<T>::carve(smem + size_t(wave) * prm->lds_per_wave, n);
  wide_copy_pods(L, prm, lane);
  wave_sync();
  const int xd = XD ? XD : n;
  const T* X = static_cast<const T*>(prm->x);
  L.xs[lane] = lane < xd ? X[size_t(p) * xd + lane] : T(0);
  L.g[lane] = T(0);
  L.hd[lane] = T(0);
  L.st->acc_passes = 0; L.st->eval_passes = 0; L.st->solves = 0; L.st->problems = 0;
  lm_init<T>(L, lane);
  wide_store_state(L, static_cast<WideState<T>*>(prm->state) + p, lane);
}

template <typename Model>
__global__ void __launch_bounds__(256) wide_partial_kernel(const WideParams* __restrict__ prm) {
  using T = typename Model::Scalar;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = prm->n, S = prm->splits;
  const long long unit = (long long)blockIdx.x * 4 + wave;
  if (unit >= prm->P * S) return;
  const long long p = unit / S;
  const int sidx = int(unit % S);
  const WideState<T>* ws = static_cast<const WideState<T>*>(prm->state) + p;
  if (ws->st.stop != TOA_STOP_NONE || ws->st.iter >= ws->st.max_iters) return;  // this problem is finished
  Model model;
  model.init(n, prm->m, prm->data);
  model.set_loss(prm->loss, prm->loss_th2);
  WaveLds<T> L = WaveLds<T>::carve(model_bind_stage(model, smem + size_t(wave) * prm->lds_per_wave, n), n);
  L.xs[lane] = ws->xs[lane];
  wave_sync();
  const bool do_acc = prm->opt.solver_type != 0 || ws->st.rebuild;
  const int m4 = (prm->m + 3) & ~3;
  const int row0 = sidx * prm->chunk_rows;
  const int rows = min(prm->chunk_rows, m4 - row0);
  model.bind_chunk(p, row0, rows, n);
  const int stride = n * n + n + 2;
  T* part = static_cast<T*>(prm->partials) + (size_t(p) * S + sidx) * stride;
  T c;
  int nr;
  if (do_acc) {
    model.accumulate(L, n, lane, c, nr);
    model.write_sym(part, n, n, lane);
    wave_sync();
    if (lane < n) {
      part[lane * n + lane] = L.hd[lane];
      part[n * n + lane] = L.g[lane];
    }
  } else {
    model.evaluate(L, n, lane, c, nr);
  }
  if (lane == 0) { part[n * n + n] = c; part[n * n + n + 1] = T(model_inliers(model, -1, 0)); }  // -1: model has no robust loss
}

// Model for the step kernel: "accumulate" = fold the S chunk partials (fixed order => deterministic).
template <typename T, int NPAD, typename Manifold>
struct PartialSumModel {
  using Scalar = T;
  __device__ __forceinline__ void set_loss(int, double) {}  // no M-estimator on this family
  static constexpr int kNpad = NPAD;
  static constexpr int kXdim = Manifold::kXdim;
  const T* part;
  T* hsum;
  int S, n_, m;
  int ninl;
  bool direct = false;  // the partials live in LDS (team form, S <= 8): lane e sums element e over the chunks itself — S LDS reads
                        // and S - 1 additions instead of the transposed wave reductions that hide HBM latency in fold_small
  __device__ __forceinline__ T fold(int off) const {
    T s = 0;
    const int stride = n_ * n_ + n_ + 2;
    int k = 0;
    for (; k + 4 <= S; k += 4) {  // four loads in flight, added in index order (fixed association => deterministic)
      const T a0 = part[size_t(k) * stride + off], a1 = part[size_t(k + 1) * stride + off];
      const T a2 = part[size_t(k + 2) * stride + off], a3 = part[size_t(k + 3) * stride + off];
      s += a0; s += a1; s += a2; s += a3;
    }
    for (; k < S; ++k) s += part[size_t(k) * stride + off];
    return s;
  }
  // Small systems (n*n + n + 2 <= 64, i.e. n <= 6: BASELINE configs C2 / C5): lane j fetches the WHOLE partial of
  // chunk j with all its loads in flight at once, then each element is summed across the lanes by a fixed reduction
  // tree — ~3 us for 64 chunks, where a per-element serial walk over the chunks is 64 dependent HBM round trips.
  // Returns this lane's element total (lane e <-> element e of [H | g | cost | inliers]).
  __device__ __forceinline__ T fold_small(int lane, bool full) const {
    const int stride = n_ * n_ + n_ + 2;
    const int e0 = full ? 0 : n_ * n_ + n_;  // cost-only: just the last two elements
    T tot = 0;
    for (int base = 0; base < S; base += 64) {
      const int k = base + lane;
      const bool valid = k < S;
      const T* src = part + size_t(valid ? k : 0) * stride;
      T v[64];
#pragma unroll
      for (int e = 0; e < 64; ++e) v[e] = (valid && e >= e0 && e < stride) ? src[e] : T(0);
      if (full) {
        // two transposed 32-value reductions (~230 instructions each in fp64) instead of one all-reduce per element
        // (n^2 + n + 2 = 44 of them at n = 6, ~1 100 instructions): this fold is on the critical path of EVERY iteration of a
        // single-problem solve, executed by one wave on an otherwise idle CU
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          if (32 * half < stride) {  // wave-uniform
            T p32[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) p32[i] = v[32 * half + i];
            const T r = wave_transposed_reduce32(p32, lane);   // lane l: total of element 32 half + (l & 31)
            tot += ((lane >> 5) == half) ? r : T(0);
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 64; ++e) {
          if (e >= e0 && e < stride) {  // wave-uniform
            const T t = wave_allreduce_sum(v[e]);
            tot += (lane == e) ? t : T(0);
          }
        }
      }
    }
    return tot;
  }
  __device__ __forceinline__ bool small() const { return n_ * n_ + n_ + 2 <= 64; }
  __device__ __forceinline__ void accumulate(WaveLds<T>& L, int n, int lane, T& cost, int& nres) {
    if (small()) {
      const T tot = direct ? (lane < n * n + n + 2 ? fold(lane) : T(0)) : fold_small(lane, true);
      const int nn = n * n;
      if (lane < nn) {
        hsum[lane] = tot;
        if (lane / n == lane % n) L.hd[lane / n] = tot;
      } else if (lane < nn + n) {
        L.g[lane - nn] = tot;
      }
      cost = wave_bcast(tot, nn + n);
      ninl = int(wave_bcast(tot, nn + n + 1));
    } else {
      for (int e = lane; e < n * n; e += 64) hsum[e] = fold(e);
      if (lane < n) { L.g[lane] = fold(n * n + lane); L.hd[lane] = fold(lane * n + lane); }
      cost = fold(n * n + n);
      ninl = int(fold(n * n + n + 1));
    }
    if (ninl < 0) ninl = m;
    nres = m;
    wave_sync();
  }
  __device__ __forceinline__ void evaluate(WaveLds<T>&, int n, int lane, T& cost, int& nres) {
    if (small()) {
      const T tot = direct ? (lane >= n * n + n && lane < n * n + n + 2 ? fold(lane) : T(0)) : fold_small(lane, false);
      cost = wave_bcast(tot, n * n + n);
      ninl = int(wave_bcast(tot, n * n + n + 1));
    } else {
      cost = fold(n * n + n);
      ninl = int(fold(n * n + n + 1));
    }
    if (ninl < 0) ninl = m;
    nres = m;
  }
  template <typename O>
  __device__ __forceinline__ void write_sym(O* M, int LD, int n, int lane) const {
    for (int e = lane; e < n * n; e += 64) M[(e / n) * LD + (e % n)] = O(hsum[e]);
  }
  __device__ __forceinline__ void plus_eq(WaveLds<T>& L, const T* d, T sign, int n, int lane) const {
    Manifold::plus_eq(L, d, sign, n, lane);
  }
};

template <typename T, int NPAD, typename Manifold>
__global__ void __launch_bounds__(256) wide_step_kernel(const WideParams* __restrict__ prm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long p = (long long)blockIdx.x * 4 + wave;
  if (p >= prm->P) return;
  const int n = prm->n;
  WideState<T>* ws = static_cast<WideState<T>*>(prm->state) + p;
  if (ws->st.stop != TOA_STOP_NONE || ws->st.iter >= ws->st.max_iters) return;
  WaveLds<T> L = WaveLds<T>::carve(smem + size_t(wave) * prm->lds_per_wave, n);
  wide_copy_pods(L, prm, lane);
  wide_load_state(L, ws, lane);
  PartialSumModel<T, NPAD, Manifold> model;
  model.S = prm->splits;
  model.n_ = n;
  model.m = prm->m;
  model.part = static_cast<const T*>(prm->partials) + size_t(p) * prm->splits * (n * n + n + 2);
  model.hsum = static_cast<T*>(prm->hsum) + size_t(p) * n * n;
  const bool more = lm_iteration<T>(model, L, n, lane, p);
  if (more && prm->step_mode) {  // the reference's Step updates x in place every time (optimizer.h:271-279)
    const int xd = Manifold::kXdim ? Manifold::kXdim : n;
    T* X = static_cast<T*>(prm->x);
    if (lane < xd) X[size_t(p) * xd + lane] = L.xs[lane];
    if (lane == 0) {
      prm->res.num_iters[p] = L.st->num_iters;
      prm->res.final_cost[p] = L.st->final_cost;
      prm->res.stop_reason[p] = TOA_STOP_NONE;
      if (prm->active) atomicAdd(prm->active, 1);
    }
  }
  if (!more) {
    lm_finalize<T>(model, L, n, lane, p);  // sets a non-zero StopReason: later launches skip this problem
    const int xd = Manifold::kXdim ? Manifold::kXdim : n;
    T* X = static_cast<T*>(prm->x);
    if (lane < xd) X[size_t(p) * xd + lane] = L.xs[lane];
    if (prm->counters && lane == 0) {
      atomicAdd(&prm->counters[0], L.st->acc_passes);
      atomicAdd(&prm->counters[1], L.st->eval_passes);
      atomicAdd(&prm->counters[2], L.st->solves);
      atomicAdd(&prm->counters[3], L.st->problems);
    }
  }
  wide_store_state(L, ws, lane);
}

// Host-side stop controls of the stepping form (`Options::stop_callback`, `stop_callback2`, `max_duration_ms`;
// optimizer.h:302-305, 529-534): the host evaluates them between two toa_lm_step calls on what toa_lm_step_info reads
// back and names the problems to stop; this kernel ends those problems exactly as the loop would have — StopReason set,
// then the finalisation of OptimizeAcc (undamped final Hessian, Output fields; optimizer.h:313-321).  x already holds
// the iterate the reference would return: its Step sets the StopReason first and OptimizeAcc still applies the step
// before leaving the loop (optimizer.h:271-309), which is what the completed toa_lm_step has done.
template <typename T, int NPAD, typename Manifold>
__global__ void __launch_bounds__(256) wide_stop_kernel(const WideParams* __restrict__ prm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long p = (long long)blockIdx.x * 4 + wave;
  if (p >= prm->P) return;
  const int req = prm->stop_request[p];
  if (req == TOA_STOP_NONE) return;
  const int n = prm->n;
  WideState<T>* ws = static_cast<WideState<T>*>(prm->state) + p;
  if (ws->st.stop != TOA_STOP_NONE || ws->st.iter >= ws->st.max_iters) return;  // already finished on its own
  WaveLds<T> L = WaveLds<T>::carve(smem + size_t(wave) * prm->lds_per_wave, n);
  wide_copy_pods(L, prm, lane);
  wide_load_state(L, ws, lane);
  PartialSumModel<T, NPAD, Manifold> model;
  model.S = prm->splits;
  model.n_ = n;
  model.m = prm->m;
  model.part = static_cast<const T*>(prm->partials) + size_t(p) * prm->splits * (n * n + n + 2);
  model.hsum = static_cast<T*>(prm->hsum) + size_t(p) * n * n;
  L.st->stop = req;
  wave_sync();
  lm_finalize<T>(model, L, n, lane, p);
  if (prm->counters && lane == 0) {
    atomicAdd(&prm->counters[0], L.st->acc_passes);
    atomicAdd(&prm->counters[1], L.st->eval_passes);
    atomicAdd(&prm->counters[2], L.st->solves);
    atomicAdd(&prm->counters[3], L.st->problems);
  }
  wide_store_state(L, ws, lane);
}

// What the host-side stop controls look at after a step (optimizer.h:529-534: `stop_callback(err, |dx|^2, |g|^2)`,
// `stop_callback2(err, dx, g)`): the cost, step and gradient of each problem's LAST iteration, out of the state block.
template <typename T>
__global__ void __launch_bounds__(256) step_info_kernel(const void* state_, long long P, int n, double* err, double* dx2,
                                                        double* g2, T* dx_out, T* g_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long p = (long long)blockIdx.x * 4 + wave;
  if (p >= P) return;
  const WideState<T>* ws = static_cast<const WideState<T>*>(state_) + p;
  const T d = lane < n ? ws->dx[lane] : T(0);
  const T g = lane < n ? ws->g[lane] : T(0);
  const double sd = double(wave_allreduce_sum(d * d));   // same arithmetic as lm_judge_step (optimizer.h:412-415)
  const double sg = double(wave_allreduce_sum(g * g));
  if (lane == 0) {
    if (err) err[p] = ws->st.cost_val;
    if (dx2) dx2[p] = sd;
    if (g2) g2[p] = sg;
  }
  if (dx_out && lane < n) dx_out[size_t(p) * n + lane] = d;
  if (g_out && lane < n) g_out[size_t(p) * n + lane] = g;
}

// toa_lm_step_log: what the per-iteration log line prints besides step_info's numbers (optimizer.h:463-516)
template <typename T>
__global__ void __launch_bounds__(256) step_log_kernel(const void* state_, long long P, double* lambda, int* nres, int* ninl) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const WideState<T>* ws = static_cast<const WideState<T>*>(state_) + p;
  if (lambda) lambda[p] = double(ws->st.lambda);
  if (nres) nres[p] = ws->st.cost_nres;
  if (ninl) ninl[p] = ws->st.cost_ninl;
}

// ------------------------------------------------------------------------------------------------
// Persistent form of the row-split solve: ONE launch for the whole solve instead of 1 + 2 x (max_iters + 1).
// The multi-launch form is bound by the GPU's kernel-to-kernel dependency latency (~10 us per launch, the same
// eager or replayed from a hipGraph); here the S chunk-waves of a problem stay resident and hand over through two
// generation counters in HBM:
//     every chunk wave: partial (H, g, cost) of its rows -> HBM, release, arrive += 1
//     leader (chunk 0): waits arrive == S * gen, acquire, folds the partials in fixed order, runs ONE lm_iteration
//                       (state in its LDS for the whole solve), publishes x + flags, release, go = gen + 1
//     the others      : wait go > gen, acquire, pick up x (or leave when the problem has stopped)
// Agent-scope release/acquire fences order the HBM hand-over across XCDs (separate L2s).  The launcher uses this
// form only when every workgroup is certainly co-resident (P * S <= #CUs, one 64-thread workgroup each); the waits
// poll with s_sleep and give up after ~5 s (abort flag -> StopReason kTimedOut) instead of hanging the device.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool persistent_wait(unsigned* addr, const unsigned target, unsigned* abort_flag) {
  const unsigned long long t0 = wall_clock64();  // constant 100 MHz
  for (;;) {
    if (__hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
    if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
    if (wall_clock64() - t0 > 500000000ull) {  // 5 s at the constant 100 MHz
      __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
    __builtin_amdgcn_s_sleep(2);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return true;
}

// Team form for ONE SMALL problem per workgroup (BASELINE config C2: n = 6, 1000 residuals, 56 KB): the S chunk-waves
// of a problem are the waves of ONE workgroup, their partials and the leader's x / flags live in LDS, and the two
// hand-overs of an iteration are workgroup barriers (~1 us) instead of release / acquire round trips through HBM
// (~10 us each across XCDs).  Same arithmetic in the same order as the persistent form (fixed-order fold of S partials).
template <typename Model, int NPAD, typename Manifold>
__global__ void __launch_bounds__(512) wide_team_kernel(const WideParams* __restrict__ prm) {
  using T = typename Model::Scalar;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = prm->n, S = prm->splits;  // S == blockDim.x / 64
  const long long p = blockIdx.x;
  const bool leader = wave == 0;
  const size_t pw = size_t(prm->lds_per_wave);
  WaveLds<T> L = WaveLds<T>::carve(smem + size_t(wave) * pw, n);
  const int stride = n * n + n + 2;
  T* parts = reinterpret_cast<T*>(smem + size_t(S) * pw);
  T* xshare = parts + size_t(S) * stride;
  int* flags = reinterpret_cast<int*>(xshare + 64);  // [0] stopped, [1] rebuild
  const int xd = Manifold::kXdim ? Manifold::kXdim : n;

  const int m4 = (prm->m + 3) & ~3;
  const int row0 = wave * prm->chunk_rows;
  const int rows = min(prm->chunk_rows, m4 - row0);
  Model model;
  model.init(n, prm->m, prm->data);
  model.set_loss(prm->loss, prm->loss_th2);
  static_assert(ModelStageBytes<Model>::value == 0, "the team form of the row-split kernels has no room for a model's LDS stage");
  model.bind_chunk(p, row0, rows, n);
  T* part = parts + size_t(wave) * stride;

  PartialSumModel<T, NPAD, Manifold> fold;
  fold.S = S; fold.n_ = n; fold.m = prm->m;
  fold.direct = true;
  fold.part = parts;
  fold.hsum = (n * n <= 64) ? L.aux : static_cast<T*>(prm->hsum) + size_t(p) * n * n;

  if (leader) {
    wide_copy_pods(L, prm, lane);
    wave_sync();
    const T* X = static_cast<const T*>(prm->x);
    L.xs[lane] = lane < xd ? X[size_t(p) * xd + lane] : T(0);
    L.g[lane] = T(0);
    L.hd[lane] = T(0);
    L.st->acc_passes = 0; L.st->eval_passes = 0; L.st->solves = 0; L.st->problems = 0;
    lm_init<T>(L, lane);
    xshare[lane] = L.xs[lane];
    if (lane == 0) { flags[0] = 0; flags[1] = L.st->rebuild; }
  }
  __syncthreads();
#ifdef TOA_TEAM_TIMING
  unsigned long long tt_[4] = {0, 0, 0, 0}, ttp_ = wall_clock64();
#define TEAM_TICK(i) { const unsigned long long n_ = wall_clock64(); tt_[i] += n_ - ttp_; ttp_ = n_; }
#else
#define TEAM_TICK(i)
#endif
  for (;;) {
    if (flags[0] != 0) break;  // workgroup-uniform: written before the barrier every wave has just passed
    if (!leader) { L.xs[lane] = xshare[lane]; wave_sync(); }
    const bool do_acc = prm->opt.solver_type != 0 || flags[1] != 0;
    T c;
    int nr;
    if (do_acc) {
      model.accumulate(L, n, lane, c, nr);
      model.write_sym(part, n, n, lane);
      wave_sync();
      if (lane < n) {
        part[lane * n + lane] = L.hd[lane];
        part[n * n + lane] = L.g[lane];
      }
    } else {
      model.evaluate(L, n, lane, c, nr);
    }
    if (lane == 0) { part[n * n + n] = c; part[n * n + n + 1] = T(model_inliers(model, -1, 0)); }
    TEAM_TICK(0)
    __syncthreads();
    TEAM_TICK(1)
    if (leader) {
      const bool more = lm_iteration<T>(fold, L, n, lane, p);
      TEAM_TICK(2)
      if (!more) {
        lm_finalize<T>(fold, L, n, lane, p);
        T* X = static_cast<T*>(prm->x);
        if (lane < xd) X[size_t(p) * xd + lane] = L.xs[lane];
        if (prm->counters && lane == 0) {
          atomicAdd(&prm->counters[0], L.st->acc_passes);
          atomicAdd(&prm->counters[1], L.st->eval_passes);
          atomicAdd(&prm->counters[2], L.st->solves);
          atomicAdd(&prm->counters[3], L.st->problems);
        }
      }
      xshare[lane] = L.xs[lane];
      if (lane == 0) { flags[0] = more ? 0 : 1; flags[1] = L.st->rebuild; }
    }
    __syncthreads();
    TEAM_TICK(3)
  }
#ifdef TOA_TEAM_TIMING
  if (threadIdx.x == 0 && blockIdx.x == 0)
    printf("team p=0 S=%d: data pass %.1f us  barrier %.1f us  iteration %.1f us  publish + barrier %.1f us\n", S, tt_[0] * 0.01, tt_[1] * 0.01,
           tt_[2] * 0.01, tt_[3] * 0.01);
#endif
}

template <typename Model, int NPAD, typename Manifold>
__global__ void __launch_bounds__(64) wide_persistent_kernel(const WideParams* __restrict__ prm) {
  using T = typename Model::Scalar;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const int n = prm->n, S = prm->splits;
  const long long p = (long long)blockIdx.x / S;
  const int sidx = int((long long)blockIdx.x % S);
  const bool leader = sidx == 0;
  WaveLds<T> L = WaveLds<T>::carve(smem + stage_carve_off<T>(ModelStageBytes<Model>::value, n), n);
  WideState<T>* ws = static_cast<WideState<T>*>(prm->state) + p;  // mailbox: x + flags published by the leader
  unsigned* arrive = prm->sync + 2 * p;
  unsigned* go = arrive + 1;
  unsigned* abort_flag = prm->sync + 2 * prm->P;
  const int xd = Manifold::kXdim ? Manifold::kXdim : n;

  const int m4 = (prm->m + 3) & ~3;
  const int row0 = sidx * prm->chunk_rows;
  const int rows = min(prm->chunk_rows, m4 - row0);
  Model model;
  model.init(n, prm->m, prm->data);
  model.set_loss(prm->loss, prm->loss_th2);
  (void)model_bind_stage(model, smem, n);
  model.bind_chunk(p, row0, rows, n);
  const int stride = n * n + n + 2;
  T* part = static_cast<T*>(prm->partials) + (size_t(p) * S + sidx) * stride;

  PartialSumModel<T, NPAD, Manifold> fold;
  fold.S = S; fold.n_ = n; fold.m = prm->m;
  fold.part = static_cast<const T*>(prm->partials) + size_t(p) * S * stride;
  // n <= 8: the folded H stays in LDS (L.aux) instead of making an HBM round trip between fold and factorisation
  fold.hsum = (n * n <= 64) ? L.aux : static_cast<T*>(prm->hsum) + size_t(p) * n * n;

  auto publish = [&](int stop) __attribute__((always_inline)) {
    ws->xs[lane] = L.xs[lane];
    if (lane == 0) { ws->st.rebuild = L.st->rebuild; ws->st.stop = stop; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  };

  if (leader) {
    wide_copy_pods(L, prm, lane);
    wave_sync();
    const T* X = static_cast<const T*>(prm->x);
    L.xs[lane] = lane < xd ? X[size_t(p) * xd + lane] : T(0);
    L.g[lane] = T(0);
    L.hd[lane] = T(0);
    L.st->acc_passes = 0; L.st->eval_passes = 0; L.st->solves = 0; L.st->problems = 0;
    lm_init<T>(L, lane);
    publish(TOA_STOP_NONE);
    if (lane == 0) __hip_atomic_store(go, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

#ifdef TOA_PERSIST_TIMING
  unsigned long long tk[4] = {0, 0, 0, 0}, tprev = wall_clock64();
#define TOA_TICK(i) { const unsigned long long tn = wall_clock64(); tk[i] += tn - tprev; tprev = tn; }
#else
#define TOA_TICK(i)
#endif
  for (unsigned gen = 1;; ++gen) {
    bool do_acc;
    if (leader) {
      do_acc = prm->opt.solver_type != 0 || L.st->rebuild;
    } else {
      if (!persistent_wait(go, gen, abort_flag)) return;
      if (ws->st.stop != TOA_STOP_NONE) return;  // the problem has finished
      L.xs[lane] = ws->xs[lane];
      wave_sync();
      do_acc = prm->opt.solver_type != 0 || ws->st.rebuild;
    }
    T c;
    int nr;
    if (do_acc) {
      model.accumulate(L, n, lane, c, nr);
      model.write_sym(part, n, n, lane);
      wave_sync();
      if (lane < n) {
        part[lane * n + lane] = L.hd[lane];
        part[n * n + lane] = L.g[lane];
      }
    } else {
      model.evaluate(L, n, lane, c, nr);
    }
    if (lane == 0) { part[n * n + n] = c; part[n * n + n + 1] = T(model_inliers(model, -1, 0)); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!leader) continue;
    TOA_TICK(0)
    if (!persistent_wait(arrive, unsigned(S) * gen, abort_flag)) {
      if (lane == 0) prm->res.stop_reason[p] = TOA_STOP_TIMED_OUT;
      return;
    }
    TOA_TICK(1)
    const bool more = lm_iteration<T>(fold, L, n, lane, p);
    TOA_TICK(2)
    if (!more) {
      lm_finalize<T>(fold, L, n, lane, p);
      T* X = static_cast<T*>(prm->x);
      if (lane < xd) X[size_t(p) * xd + lane] = L.xs[lane];
      if (prm->counters && lane == 0) {
        atomicAdd(&prm->counters[0], L.st->acc_passes);
        atomicAdd(&prm->counters[1], L.st->eval_passes);
        atomicAdd(&prm->counters[2], L.st->solves);
        atomicAdd(&prm->counters[3], L.st->problems);
      }
    }
    publish(more ? TOA_STOP_NONE : (L.st->stop != TOA_STOP_NONE ? L.st->stop : TOA_STOP_MAX_ITERS));
    if (lane == 0) __hip_atomic_store(go, gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    TOA_TICK(3)
#ifdef TOA_PERSIST_TIMING
    if (!more && lane == 0)
      printf("persistent p=%lld S=%d gens=%u  partial %.1f us  wait %.1f us  iteration %.1f us  publish+finalize %.1f us\n", p, S, gen,
             tk[0] * 0.01, tk[1] * 0.01, tk[2] * 0.01, tk[3] * 0.01);
#endif
    if (!more) return;
  }
}

}  // namespace toa

// Everything below is host code (launchers, the handle).  A run-time compiled model (toa_model_compile: hiprtc of a
// JetModel<T, UserFunctor> instantiation, csrc/jit.hip) includes this header for the device code above only.
#ifndef __HIPCC_RTC__
// ================================================================================================
// host side shared by the translation units
// ================================================================================================
struct toa_context {
  int device = 0;
  hipStream_t stream = nullptr;
  int num_cus = 0;
  int clock_khz = 0;
  int max_lds = 0;
  char name[128] = {0};
  int* queue = nullptr;  // device work-queue head
  bool queue_dirty = true;  // the queue block needs a memset before the next fused launch (first use, or after a failure)
  void* params_dev = nullptr;  // device copy of the fused kernel's parameter block
  int loss = TOA_LOSS_L2;      // toa_set_loss: the M-estimator of this handle's cost functor (DenseRow / Jet families)
  double loss_th2 = 0;
  toa_tuning tune = {};        // toa_set_tuning: A/B arms (all-zero = the library's choices)
  std::string timeline_path;   // toa_debug_timeline
  unsigned char params_shadow[1024] = {0};  // what params_dev holds (or will hold, in stream order): see upload_params
  size_t params_shadow_bytes = 0;
  void* scratch = nullptr;     // row-split path: state + partials + folded H (grown on demand)
  size_t scratch_bytes = 0;
  std::vector<std::unique_ptr<char[]>> captured_blocks;   // parameter blocks of launches captured into hipGraphs (upload_params)
  bool shadow_retired = false;
  // Workspaces a captured hipGraph may still point into.  Once ANY launch of this handle has been captured (shadow_retired),
  // a workspace that has to grow is not freed but parked here until toa_destroy: a graph bakes the raw device pointers of the
  // scratch / memo / aux blocks of its capture time into its nodes, and a later eager call with a larger shape must not pull
  // them from under a replay.  toa_release_workspace() is the only way a handle's workspace is given up.
  std::vector<void*> retired_blocks;
  void* memo = nullptr;        // fused kernel: one parked linearisation per resident wave (lm_device.hpp; grown on demand)
  size_t memo_bytes = 0;
  void* aux = nullptr;         // bundle adjustment with visibility lists: its work arrays (`scratch` belongs to the solver it calls)
  size_t aux_bytes = 0;
  // launch-per-stage pipelines (BA lists, n > 128): the ring through which the host reads "is anything still running" a few
  // passes late — pinned flags + one event per slot, created on first use and kept (hipHostMalloc costs ~1 ms per call)
  // set by a pipeline around its toa_large_solve call (bundle adjustment with lists): matrix p is factorised only where
  // solve_mask[p * solve_mask_stride] != 0 — the workgroups of finished scenes leave at once (own kernels only; the library
  // path solves everything, as before).  Device pointer; NULL = solve all.
  const int32_t* solve_mask = nullptr;
  int64_t solve_mask_stride = 0;
  static constexpr int kPassRing = 4, kLanes = 4;   // (lanes: the n > 128 pipeline runs the batch as up to four lanes on as many streams)
  int* pass_flags = nullptr;   // [kLanes][kPassRing][2], pinned host memory
  hipEvent_t pass_done[kLanes * kPassRing] = {};
  hipStream_t lane_stream[kLanes - 1] = {};
  hipEvent_t lane_fork = nullptr, lane_join[kLanes - 1] = {}, lane_gram[kLanes * kPassRing] = {};
  // row-split path: optional hipGraph of the (init, [partial, step] x iters) launch sequence (toa_tuning::wide_graph)
  struct WideGraph { const void* k_init; const void* k_part; const void* k_step; unsigned g_p, g_u; size_t lds; int iters; hipGraphExec_t exec; };
  WideGraph wgraphs[16];
  int nwgraphs = 0;
  // launch-configuration cache: (kernel, dynamic LDS bytes) -> resident workgroups per CU.
  // hipFuncSetAttribute / hipOccupancy* cost milliseconds per call; pay them once per variant.
  struct Cfg { const void* fn; size_t lds; int wg_per_cu; };
  Cfg cfg[256];
  int ncfg = 0;
  // large-n K3 (large_n.hip): rocBLAS handle created on the first n > 63 solve, and how to destroy it
  void* blas = nullptr;
  int (*blas_destroy)(void*) = nullptr;
  // one-matrix-per-call solves spread over side streams (toa_large_solve_each): streams, their rocBLAS handles, events
  static constexpr int kSide = 8;
  hipStream_t side_stream[kSide] = {};
  void* side_blas[kSide] = {};
  hipEvent_t side_done[kSide] = {};
  hipEvent_t side_fork = nullptr;
  int nside = 0;
};

// large_fused.hip: the n in [64, 128] loop as one persistent kernel (called by toa_large_lm_run when eligible)
bool toa_large_fused_eligible(toa_context* h, int dtype, int n, int m);
int toa_large_fused_lm_run(toa_context* h, int dtype, int n, int m, int64_t P, const void* data, void* x, const toa_options* options,
                           const toa_results* results, uint64_t* counters);

int toa_large_accumulate(toa_context* h, int dtype, int n, int m, int64_t P, const void* data, const void* x, int want_grad, void* g,
                         void* H, double* cost, int32_t* nres);

// error reporting lives in capi.hip (one thread_local message for the whole library)
int toa_fail(int code, const std::string& msg);
#define HIP_TRY(expr)                                                                           \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return toa_fail(e_ == hipErrorOutOfMemory ? TOA_E_NOMEM : TOA_E_HIP,                      \
                      std::string(#expr) + ": " + hipGetErrorString(e_));                       \
  } while (0)

// Give up a workspace block of the handle that is about to be replaced by a larger one: freed at once — unless a launch of
// this handle has ever been captured into a hipGraph, whose nodes may hold pointers into it (kept until toa_destroy then).
inline void toa_release_workspace(toa_context* h, void* block) {
  if (!block) return;
  if (h->shadow_retired) h->retired_blocks.push_back(block);
  else (void)hipFree(block);
}

// Before a device workspace is re-allocated: everything queued on the stream may still use the old block, so the stream is
// drained first — which, like the hipMalloc that follows, cannot happen while the stream is being CAPTURED into a hipGraph.
// Workspaces only ever grow and are kept, so one un-captured call of the same shape beforehand is all a capturing caller
// needs; without it the call is refused here instead of failing inside the runtime with the capture invalidated.
inline int grow_sync(toa_context* h, const char* what) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(h->stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
    return toa_fail(TOA_E_UNSUPPORTED, std::string(what) + ": a device workspace has to grow, which cannot happen while the stream is being captured; "
                                       "run this shape once before hipStreamBeginCapture (workspaces only grow and are kept by the handle)");
  HIP_TRY(hipStreamSynchronize(h->stream));
  return TOA_OK;
}

namespace toa {
// Every C entry point runs on its handle's GPU and leaves the CALLER's current device as it found it: torch (and any
// other HIP user of the process) reads its "current device" through hipGetDevice, so a library that switched it as a
// side effect would silently redirect the caller's later allocations in a single-process multi-GPU program.
struct DeviceGuard {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    int cur = -1;
    err = hipGetDevice(&cur);
    if (err == hipSuccess && cur != dev) {
      err = hipSetDevice(dev);
      if (err == hipSuccess) prev = cur;
    }
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define TOA_ON_DEVICE(dev)          \
  toa::DeviceGuard guard_(dev);     \
  HIP_TRY(guard_.err)

// Raise a kernel's dynamic-LDS limit once per (kernel, size): hipFuncSetAttribute costs ~1 ms per call.
inline int ensure_lds_attr(toa_handle h, const void* fn, size_t bytes) {
  size_t max_set = 0;
  for (int i = 0; i < h->ncfg; ++i)
    if (h->cfg[i].fn == fn) {
      if (h->cfg[i].lds == bytes) return TOA_OK;
      if (h->cfg[i].lds > max_set) max_set = h->cfg[i].lds;
    }
  // the limit only ever grows: a smaller request (another n on the same instantiation) must not lower it under a larger
  // size whose cache entry would make later launches skip this call
  if (bytes > max_set) HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  if (h->ncfg < 256) h->cfg[h->ncfg++] = {fn, bytes, -1};
  return TOA_OK;
}

// Stream-ordered upload of a kernel's parameter block into the context's device copy.  Repeated solves over the same
// buffers (an outer loop re-solving, the stepping form, the benchmark) present byte-identical blocks: the upload — a
// staged ~10 us stream operation in front of every launch — is skipped when the block already there is the same.
inline int upload_params(toa_handle h, const void* blk, size_t bytes) {
  // Under stream capture the copy below is only RECORDED, host POINTER included: the graph reads the block when it is
  // launched, long after the caller's stack copy is gone — so the block is parked in host memory the handle keeps for its
  // lifetime (1 KB per captured launch).  And once a graph of ours exists, a replay can rewrite the device block behind the
  // shadow's back at any time: from then on every eager call uploads (~10 us), the shadow is retired for this handle.
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(h->stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
    h->captured_blocks.emplace_back(new char[bytes]);
    std::memcpy(h->captured_blocks.back().get(), blk, bytes);
    HIP_TRY(hipMemcpyAsync(h->params_dev, h->captured_blocks.back().get(), bytes, hipMemcpyHostToDevice, h->stream));
    h->params_shadow_bytes = 0;
    h->shadow_retired = true;
    return TOA_OK;
  }
  if (h->shadow_retired) {
    HIP_TRY(hipMemcpyAsync(h->params_dev, blk, bytes, hipMemcpyHostToDevice, h->stream));
    return TOA_OK;
  }
  if (bytes == h->params_shadow_bytes && std::memcmp(h->params_shadow, blk, bytes) == 0) return TOA_OK;
  HIP_TRY(hipMemcpyAsync(h->params_dev, blk, bytes, hipMemcpyHostToDevice, h->stream));
  std::memcpy(h->params_shadow, blk, bytes);
  h->params_shadow_bytes = bytes;
  return TOA_OK;
}

inline int ensure_pass_ring(toa_handle h) {
  if (h->pass_flags) return TOA_OK;
  constexpr int kSlots = toa_context::kPassRing * toa_context::kLanes;
  for (int i = 0; i < kSlots; ++i) HIP_TRY(hipEventCreateWithFlags(&h->pass_done[i], hipEventDisableTiming));
  for (hipStream_t& s : h->lane_stream) HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  HIP_TRY(hipEventCreateWithFlags(&h->lane_fork, hipEventDisableTiming));
  for (hipEvent_t& e : h->lane_join) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (hipEvent_t& e : h->lane_gram) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->pass_flags), kSlots * 2 * sizeof(int), hipHostMallocDefault));
  return TOA_OK;
}

// waves per workgroup: 4 (256 threads) everywhere but the team form of the fused kernel; LDS per wave decides how many WGs fit per CU.
template <typename T>
inline int lds_fit(toa_handle h, int n, size_t* per_wave, size_t* per_wg, int waves = 4, size_t stage = 0) {
  size_t pw = WaveLds<T>::bytes(n);
  pw = (pw + 15) & ~size_t(15);
  pw += stage_carve_off<T>(stage, n);   // a model's LDS stage (row_model.hpp) overlays the carve's pass-dead head: what sticks out in front
  *per_wave = pw;
  *per_wg = pw * waves;
  if (*per_wg > 160 * 1024) return toa_fail(TOA_E_UNSUPPORTED, "LDS footprint exceeds 160 KiB per workgroup");
  (void)h;
  return TOA_OK;
}

// Chunks per pass of a cooperative model — a property of the SHAPE, never of the batch, the position in it, or of which
// form of the kernel runs (classic / team), so that a problem's bits depend on none of them: ~1024 rows per chunk
// (256 for the 64-row super-batch layouts); toa_tuning::coop_chunks overrides (experiments, and the team form's tests).
template <typename Model>
inline void coop_chunking(toa_handle h, int n, int m, int* K_out, int* cs_out) {
  (void)n;
  const int steps_total = (m + 3) / 4;
  constexpr bool super16 = Model::kCoopPeriod == 16;
  int K = super16 ? std::max(2, std::min(16, (m + 128) / 256)) : std::max(2, std::min(16, (m + 512) / 1024));
  if (h->tune.coop_chunks >= 2 && h->tune.coop_chunks <= 64) K = h->tune.coop_chunks;
  const int period = Model::kCoopPeriod;   // steps per ring turn / super-batch: chunk boundaries fall on it
  int cs = (steps_total + K - 1) / K;
  cs = (cs + period - 1) / period * period;
  *cs_out = cs;
  *K_out = (steps_total + cs - 1) / cs;
}
template <typename Model>
inline int launch_accumulate(toa_handle h, int n, int m, int64_t P, const void* data, const void* x, int want_grad,
                             void* g, void* H, double* cost, int32_t* nres) {
  using T = typename Model::Scalar;
  long long grid = (P + 3) / 4;
  const long long cap = (long long)h->num_cus * 8;
  if (grid > cap) grid = cap;
  size_t pw, pwg;
  if (int rc = lds_fit<T>(h, n, &pw, &pwg, 4, ModelStageBytes<Model>::value)) return rc;
  using RModel = typename RobustOf<Model>::type;
  if (h->loss != TOA_LOSS_L2 && !std::is_same<RModel, Model>::value) {
    if (int rc = ensure_lds_attr(h, (const void*)accumulate_kernel<RModel>, pwg)) return rc;
    hipLaunchKernelGGL((accumulate_kernel<RModel>), dim3((unsigned)grid), dim3(256), pwg, h->stream, data, x, (long long)P, n, m,
                       want_grad, g, H, cost, nres, (int)pw, h->loss, h->loss_th2);
  } else {
    if (int rc = ensure_lds_attr(h, (const void*)accumulate_kernel<Model>, pwg)) return rc;
    hipLaunchKernelGGL((accumulate_kernel<Model>), dim3((unsigned)grid), dim3(256), pwg, h->stream, data, x, (long long)P, n, m,
                       want_grad, g, H, cost, nres, (int)pw, h->loss, h->loss_th2);
  }
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}

template <typename Model>
inline int launch_fused(toa_handle h, const FusedParams& prm_in) {
  using T = typename Model::Scalar;
  constexpr int kW = ModelWaves<Model>::value;
  constexpr int NO = kW;   // (every wave of a workgroup pulls problems)
  FusedParams prm = prm_in;
  size_t pw, pwg;
  if (int rc = lds_fit<T>(h, prm.n, &pw, &pwg, kW)) return rc;
  prm.lds_per_wave = (int)pw;
  prm.queue = h->queue;
  // [0] pop counter, [16] waves that have left: zeroed when the handle is created and by the last wave of every launch
  // (lm_fused_kernel); a launch that failed may have left them dirty, so the next one starts from a memset again
  if (h->queue_dirty) {
    HIP_TRY(hipMemsetAsync(h->queue, 0, 48 * sizeof(int), h->stream));
    h->queue_dirty = false;
  }
  auto kern = lm_fused_kernel<Model>;
  // resident workgroups per CU for a dynamic-LDS size (cached: the two HIP calls cost milliseconds)
  auto occupancy = [&](size_t lds_bytes, int* out) -> int {
    int w = 0;
    size_t max_set = 0;
    for (int i = 0; i < h->ncfg; ++i)
      if (h->cfg[i].fn == (const void*)kern) {
        if (h->cfg[i].lds == lds_bytes && h->cfg[i].wg_per_cu > 0) w = h->cfg[i].wg_per_cu;
        if (h->cfg[i].lds > max_set) max_set = h->cfg[i].lds;
      }
    if (w == 0) {
      if (lds_bytes > max_set)   // the limit only ever grows: a smaller request must not lower it under a cached larger one
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
      HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&w, kern, 64 * kW, lds_bytes));
      if (w < 1) w = 1;
      if (h->ncfg < 256) h->cfg[h->ncfg++] = {(const void*)kern, lds_bytes, w};
    }
    *out = w;
    return TOA_OK;
  };
  prm.stage_off = 0;
  prm.carve_off = 0;
  if constexpr (ModelStageBytes<Model>::value > 0) {
    // The LDS stage of the row-per-lane pass: it begins at the wave's region and ends INSIDE the carve, over the part of it
    // that is dead while a pass runs (WaveLds::pass_dead_bytes: LDL^T workspace, the solve's scratch, the step).  C3 (fp64,
    // n = 12): 8 192 + 6 464 = 14 656 bytes per wave were two workgroups per compute unit (round 3 / 4 ran this kernel at
    // two waves per SIMD without noticing); overlaid 11 104, with the memo's 2 048 behind them 13 152 <= 160 KiB / 12: three.
    const size_t dead = std::min(WaveLds<T>::pass_dead_bytes(prm.n), size_t(ModelStageBytes<Model>::value));
    prm.carve_off = int(ModelStageBytes<Model>::value - dead);
    pw += prm.carve_off;
    pwg = pw * kW;
    prm.lds_per_wave = (int)pw;
  }
  int wg_per_cu = 0;
  if (int rc = occupancy(pwg, &wg_per_cu)) return rc;
  prm.memo = nullptr;
  prm.memo_stride = 0;
  prm.memo_lds_off = 0;
  bool memo_on = false;
  if constexpr (ModelMemo<Model>::value) {
    // One parked linearisation per resident wave (the Gram registers of the last accepted point: ~10 KB at n = 50, 2 KB at
    // n = 12 fp64): the re-accumulation that follows a rejected step reads it back instead of streaming the problem's rows
    // again.  toa_tuning::memo_off switches it off (A/B, and the test that the results do not depend on it).
    memo_on = !h->tune.memo_off;
    if (memo_on) {
      // a small Gram is parked in LDS when that costs no resident workgroup (C3: parking in HBM after every accepted step
      // measured 1.5 % of the launch for a workload that never rejects a step)
      const size_t mb = (Model::kMemoBytes + 15) & ~size_t(15);
      if (mb <= 4096 && (pw + mb) * 4 <= 160 * 1024) {
        int w2 = 0;
        if (int rc = occupancy((pw + mb) * 4, &w2)) return rc;
        if (w2 == wg_per_cu) {
          prm.memo_lds_off = (int)pw;
          pw += mb;
          pwg = pw * 4;
          prm.lds_per_wave = (int)pw;
        }
      }
    }
  }
  prm.coop_K = 0;
  prm.coop_cs = 0;
  prm.coop_tot_off = 0;
  if constexpr (ModelCoop<Model>::value) {
    prm.coop_K = 1;                       // one chunk = the classic pass, bit for bit
    prm.coop_cs = (prm.m + 3) / 4;
    pwg += kCoopCtlBytes;                 // the control block (the waves' carves are far below the LDS limit of a smaller grid)
    // Cooperative passes (CoopCtl): on for a SHAPE (never for a batch size or a position in the batch, so that a problem's
    // bits do not depend on them), when a pass has enough rows to be worth sharing.  The chunk total of a pass is summed in
    // the owner's LDL^T workspace when the Gram registers fit it, in an area of its own otherwise (if that costs no
    // resident workgroup).  toa_tuning::coop_off switches it off (A/B).
    const bool coop_on = !h->tune.coop_off;
    constexpr bool super16 = Model::kCoopPeriod == 16;   // fp64 n <= 15: 64-row super-batches, 52 KB problems — share from 256 rows
    bool room = Model::kMemoBytes <= WaveLds<T>::m_elems(prm.n) * sizeof(T);
    if (coop_on && prm.m >= (super16 ? 256 : 1024) && !room) {
      const size_t mb = (Model::kMemoBytes + 15) & ~size_t(15);
      int w2 = 0;
      if ((pw + mb) * kW + kCoopCtlBytes <= 160 * 1024) {
        if (int rc = occupancy((pw + mb) * kW + kCoopCtlBytes, &w2)) return rc;
        if (w2 == wg_per_cu) {
          prm.coop_tot_off = (int)pw;
          pw += mb;
          pwg = pw * kW + kCoopCtlBytes;
          prm.lds_per_wave = (int)pw;
          room = true;
        }
      }
    }
    if (coop_on && prm.m >= (super16 ? 256 : 1024) && room) {
      // chunks per pass: ~1024 rows each (256 for the super-batch form).  Same box, C4 (m = 2000), three interleaved rounds
      // (profiles/r03_ab_log.md): K = 2: 12.83 M it/s, K = 3: 12.69, K = 4: 12.71, K = 8: 12.41, off: 12.52 — every chunk pays
      // its own ramp of the load ring, so the coarsest split that still lets a sibling help wins.  (toa_tuning::coop_chunks: experiments)
      coop_chunking<Model>(h, prm.n, prm.m, &prm.coop_K, &prm.coop_cs);
    }
  }
  long long grid = (long long)h->num_cus * wg_per_cu;
  const long long need = (prm.P + NO - 1) / NO;
  if (grid > need) grid = need;
  if (grid < 1) grid = 1;
  // (Sizing the grid to P / rounds waves so that every round is full was tried: at the BASELINE shard size 625 workgroups
  // instead of 768 are ~2 % slower, tools/grid_ab.sh — more resident waves hide more latency than full rounds save.)
  if (h->tune.max_workgroups > 0 && grid > h->tune.max_workgroups) grid = h->tune.max_workgroups;   // experiments only
  if constexpr (ModelMemo<Model>::value) {
    if (memo_on && prm.memo_lds_off == 0) {
      const size_t stride = (Model::kMemoBytes + 255) & ~size_t(255);
      const size_t need_b = stride * size_t(grid) * NO;
      if (need_b > h->memo_bytes) {
        if (int rc = grow_sync(h, "memo of the last accepted linearisation")) return rc;
        toa_release_workspace(h, h->memo);
        h->memo = nullptr;
        h->memo_bytes = 0;
        HIP_TRY(hipMalloc(&h->memo, need_b));
        h->memo_bytes = need_b;
      }
      prm.memo = h->memo;
      prm.memo_stride = stride;
    }
  }
  static_assert(sizeof(FusedParams) <= 1024, "parameter block too large");
  // stream-ordered upload of the parameter block (kept out of the kernarg segment so that its ~60
  // scalars are loaded on demand instead of being pinned in SGPRs across the hot loop)
  if (int rc = upload_params(h, &prm, sizeof(prm))) return rc;
  const char* tl_path = h->timeline_path.empty() ? nullptr : h->timeline_path.c_str();
  unsigned long long* tl_dev = nullptr;
  if (tl_path) {  // debug: per-problem start / end stamps of this launch, appended to the file as text
    HIP_TRY(hipMalloc(&tl_dev, size_t(prm.P) * 16));
    HIP_TRY(hipMemsetAsync(tl_dev, 0, size_t(prm.P) * 16, h->stream));
    prm.timeline = tl_dev;
    if (int rc = upload_params(h, &prm, sizeof(prm))) return rc;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * kW), pwg, h->stream, (const FusedParams*)h->params_dev);
  if (hipError_t e_ = hipGetLastError(); e_ != hipSuccess) {
    h->queue_dirty = true;
    return toa_fail(TOA_E_HIP, std::string("lm_fused_kernel launch: ") + hipGetErrorString(e_));
  }
  if (tl_path) {
    std::vector<unsigned long long> tl(size_t(prm.P) * 2);
    HIP_TRY(hipMemcpy(tl.data(), tl_dev, tl.size() * 8, hipMemcpyDeviceToHost));
    (void)hipFree(tl_dev);
    if (FILE* f = std::fopen(tl_path, "a")) {
      std::fprintf(f, "# launch P=%lld grid=%lld\n", prm.P, grid);
      for (long long q = 0; q < prm.P; ++q) std::fprintf(f, "%llu %llu\n", tl[2 * q], tl[2 * q + 1]);
      std::fclose(f);
    }
  }
  return TOA_OK;
}

// Stepping form (`lm::Optimizer<H_t> optimizer(options)`; `optimizer.Step(x, acc, out)`, optimizer.h:199,331-539) on the
// launch-per-iteration kernels above with ONE chunk per problem: begin = wide_init_kernel, a step = the data pass
// (wide_partial_kernel<Model>: H, g, cost of the current x into the caller's state block) + wide_step_kernel (one
// lm_iteration).  The H of the last build stays in the state block, which is what eval-only iterations keep solving
// with while x sits at a trial point (optimizer.h:281-299, lm.h:96-117).
// State block layout: [ WideState<T>[P] | partial (H, g, cost, inliers)[P] | folded H [P][n*n] ], each 256-byte aligned.
template <typename T>
inline size_t stepping_state_bytes(int n, long long P, size_t* o_part = nullptr, size_t* o_hsum = nullptr) {
  const size_t stride = size_t(n) * n + n + 2;
  const size_t b_state = (size_t(P) * sizeof(WideState<T>) + 255) & ~size_t(255);
  const size_t b_part = (size_t(P) * stride * sizeof(T) + 255) & ~size_t(255);
  const size_t b_hsum = (size_t(P) * n * n * sizeof(T) + 255) & ~size_t(255);
  if (o_part) *o_part = b_state;
  if (o_hsum) *o_hsum = b_state + b_part;
  return b_state + b_part + b_hsum;
}

template <typename Model, int NPAD, typename Manifold>
inline int launch_stepping(toa_handle h, const FusedParams& fp) {
  using T = typename Model::Scalar;
  const int n = fp.n, m = fp.m;
  const long long P = fp.P;
  size_t pw, pwg, o_part, o_hsum;
  if (int rc = lds_fit<T>(h, n, &pw, &pwg, 4, ModelStageBytes<Model>::value)) return rc;
  (void)stepping_state_bytes<T>(n, P, &o_part, &o_hsum);
  WideParams wp;
  std::memset(&wp, 0, sizeof(wp));
  wp.data = fp.data; wp.x = fp.x; wp.P = P; wp.n = n; wp.m = m;
  wp.splits = 1;
  wp.chunk_rows = (((m + 3) & ~3) + 15) & ~15;
  wp.opt = fp.opt; wp.res = fp.res; wp.counters = fp.counters;
  wp.state = fp.state;
  wp.partials = static_cast<char*>(fp.state) + o_part;
  wp.hsum = static_cast<char*>(fp.state) + o_hsum;
  wp.step_mode = 1;
  wp.active = fp.active;
  wp.stop_request = fp.stop_request;
  wp.loss = fp.loss; wp.loss_th2 = fp.loss_th2;
  wp.lds_per_wave = int(pw);
  if (int rc = upload_params(h, &wp, sizeof(wp))) return rc;
  const WideParams* dp = static_cast<const WideParams*>(h->params_dev);
  const unsigned g_p = unsigned((P + 3) / 4);
  if (fp.mode == 1) {
    auto k_init = wide_init_kernel<T, Manifold::kXdim>;
    if (int rc = ensure_lds_attr(h, (const void*)k_init, pwg)) return rc;
    hipLaunchKernelGGL(k_init, dim3(g_p), dim3(256), pwg, h->stream, dp);
  } else if (fp.mode == 3) {
    auto k_stop = wide_stop_kernel<T, NPAD, Manifold>;
    if (int rc = ensure_lds_attr(h, (const void*)k_stop, pwg)) return rc;
    hipLaunchKernelGGL(k_stop, dim3(g_p), dim3(256), pwg, h->stream, dp);
  } else {
    using RModel = typename RobustOf<Model>::type;
    const bool robust = fp.loss != TOA_LOSS_L2 && !std::is_same<RModel, Model>::value;
    void (*k_part)(const WideParams*) = robust ? wide_partial_kernel<RModel> : wide_partial_kernel<Model>;
    auto k_step = wide_step_kernel<T, NPAD, Manifold>;
    if (int rc = ensure_lds_attr(h, (const void*)k_part, pwg)) return rc;
    if (int rc = ensure_lds_attr(h, (const void*)k_step, pwg)) return rc;
    hipLaunchKernelGGL(k_part, dim3(g_p), dim3(256), pwg, h->stream, dp);
    hipLaunchKernelGGL(k_step, dim3(g_p), dim3(256), pwg, h->stream, dp);
  }
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}

// Row-split driver.  Model = the chunk-capable residual model, NPAD / Manifold as for the step kernel.
template <typename Model, int NPAD, typename Manifold>
inline int launch_wide(toa_handle h, const FusedParams& fp, int splits_req) {
  using T = typename Model::Scalar;
  if (fp.mode != 0) return launch_stepping<Model, NPAD, Manifold>(h, fp);
  const int n = fp.n, m = fp.m;
  const long long P = fp.P;
  const int m4 = (m + 3) & ~3;
  // chunking (automatic): short chunks keep the per-iteration latency down — one wave streams its chunk at HBM
  // round-trip pace (~2 us per 16-row batch) — while the fold of the S partials costs ~3 us per 64 chunks for n <= 6
  // and S * n^2 / 64 serial loads per lane beyond; and P * S <= #CUs keeps the one-launch persistent form available.
  long long S = splits_req > 0 ? splits_req : m4 / (n <= 6 ? 32 : 256);
  if (splits_req <= 0) {
    if (S > 64) S = 64;
    const long long cap = (long long)h->num_cus / (P > 0 ? P : 1);
    if (cap >= 1 && S > cap) S = cap;
  }
  if (S > m4 / 16) S = m4 / 16;
  if (S < 1) S = 1;
  int chunk = int((m4 + S - 1) / S);
  chunk = (chunk + 15) & ~15;
  S = (m4 + chunk - 1) / chunk;
  size_t pw, pwg;
  if (int rc = lds_fit<T>(h, n, &pw, &pwg, 4, ModelStageBytes<Model>::value)) return rc;
  const size_t stride = size_t(n) * n + n + 2;
  const size_t b_state = (size_t(P) * sizeof(WideState<T>) + 255) & ~size_t(255);
  const size_t b_part = (size_t(P) * S * stride * sizeof(T) + 255) & ~size_t(255);
  const size_t b_hsum = (size_t(P) * n * n * sizeof(T) + 255) & ~size_t(255);
  const size_t b_sync = (size_t(2 * P + 1) * sizeof(unsigned) + 255) & ~size_t(255);
  const size_t need = b_state + b_part + b_hsum + b_sync;
  if (need > h->scratch_bytes) {
    if (int rc = grow_sync(h, "device workspace")) return rc;
    toa_release_workspace(h, h->scratch);
    h->scratch = nullptr;
    h->scratch_bytes = 0;
    HIP_TRY(hipMalloc(&h->scratch, need));
    h->scratch_bytes = need;
  }
  WideParams wp;
  std::memset(&wp, 0, sizeof(wp));
  wp.data = fp.data; wp.x = fp.x; wp.P = P; wp.n = n; wp.m = m;
  wp.splits = int(S); wp.chunk_rows = chunk;
  wp.opt = fp.opt; wp.res = fp.res; wp.counters = fp.counters;
  wp.state = h->scratch;
  wp.partials = static_cast<char*>(h->scratch) + b_state;
  wp.hsum = static_cast<char*>(h->scratch) + b_state + b_part;
  wp.sync = reinterpret_cast<unsigned*>(static_cast<char*>(h->scratch) + b_state + b_part + b_hsum);
  wp.lds_per_wave = int(pw);
  wp.loss = fp.loss; wp.loss_th2 = fp.loss_th2;
  static_assert(sizeof(WideParams) <= 1024, "parameter block too large");
  if (int rc = upload_params(h, &wp, sizeof(wp))) return rc;
  const WideParams* dp = static_cast<const WideParams*>(h->params_dev);
  // With an M-estimator on the handle (toa_set_loss) the data pass is the ROBUST variant of the model, which exists in the
  // pass-only kernels: the solve then runs in the launch-per-iteration form (no team / persistent kernel).
  using RModel = typename RobustOf<Model>::type;
  const bool robust = fp.loss != TOA_LOSS_L2 && !std::is_same<RModel, Model>::value;
  auto k_init = wide_init_kernel<T, Manifold::kXdim>;
  void (*k_part)(const WideParams*) = robust ? wide_partial_kernel<RModel> : wide_partial_kernel<Model>;
  auto k_step = wide_step_kernel<T, NPAD, Manifold>;
  if (int rc = ensure_lds_attr(h, (const void*)k_init, pwg)) return rc;
  if (int rc = ensure_lds_attr(h, (const void*)k_part, pwg)) return rc;
  if (int rc = ensure_lds_attr(h, (const void*)k_step, pwg)) return rc;
  const unsigned g_p = unsigned((P + 3) / 4), g_u = unsigned((P * S + 3) / 4);
  const int iters = fp.opt.max_iters + 1 + (fp.opt.check_final_cost ? 1 : 0);  // optimizer.h:248-250
  // Direct launches by default: an A/B on MI355X (tests/tools/latency_probe.py) shows graph replay and eager launches
  // of this 23..103-kernel sequence within 1 % of each other (C2 71 us, C5 93-99 us device time per solve), as
  // MI355X_MICROARCH.md's "boundary" row predicts (eager == hipGraph).  toa_tuning::wide_graph selects the graph path.
  // Persistent form (one launch for the whole solve) whenever every workgroup is certainly co-resident: one
  // 64-thread workgroup per chunk, at most one per CU.  toa_tuning::wide_multilaunch forces the launch-per-iteration form.
  // Instantiated for the small systems only (n <= 15: BASELINE configs C2 / C5 are n = 6): the kernel carries a whole
  // lm_iteration with the NPAD-unrolled register LDL^T per residual-model layout, and 40 copies of it tripled the build.
  const bool multilaunch_env = h->tune.wide_multilaunch != 0;
  const bool multilaunch = multilaunch_env || robust;
  const bool noteam = h->tune.wide_no_team != 0;
  if constexpr (NPAD <= 16) {
    // Team form: a small problem (<= 4096 rows) is cheaper on ONE compute unit with barrier hand-overs than on 16-64
    // of them with HBM hand-overs.  Up to 8 waves (512 threads: two waves per SIMD keep the whole register file usable).
    if (!multilaunch && !noteam && splits_req <= 0 && m4 <= 4096 && m4 >= 32) {
      long long St = std::min<long long>(8, m4 / 16);
      int chunk_t = int((m4 + St - 1) / St);
      chunk_t = (chunk_t + 15) & ~15;
      St = (m4 + chunk_t - 1) / chunk_t;
      const size_t lds_team = size_t(St) * pw + (size_t(St) * stride + 64) * sizeof(T) + 16;
      if (lds_team <= size_t(h->max_lds)) {
        wp.splits = int(St);
        wp.chunk_rows = chunk_t;
        if (int rc = upload_params(h, &wp, sizeof(wp))) return rc;
        auto k_team = wide_team_kernel<Model, NPAD, Manifold>;
        if (int rc = ensure_lds_attr(h, (const void*)k_team, lds_team)) return rc;
        hipLaunchKernelGGL(k_team, dim3(unsigned(P)), dim3(unsigned(64 * St)), lds_team, h->stream, dp);
        HIP_TRY(hipGetLastError());
        return TOA_OK;
      }
    }
    if (!multilaunch && P * S <= (long long)h->num_cus && pw <= 64 * 1024) {
      auto k_pers = wide_persistent_kernel<Model, NPAD, Manifold>;
      if (int rc = ensure_lds_attr(h, (const void*)k_pers, pw)) return rc;
      HIP_TRY(hipMemsetAsync(wp.sync, 0, size_t(2 * P + 1) * sizeof(unsigned), h->stream));
      hipLaunchKernelGGL(k_pers, dim3(unsigned(P * S)), dim3(64), pw, h->stream, dp);
      HIP_TRY(hipGetLastError());
      return TOA_OK;
    }
  }
  const bool use_graph = h->tune.wide_graph != 0;
  if (!use_graph) {
    hipLaunchKernelGGL(k_init, dim3(g_p), dim3(256), pwg, h->stream, dp);
    for (int it = 0; it < iters; ++it) {
      hipLaunchKernelGGL(k_part, dim3(g_u), dim3(256), pwg, h->stream, dp);
      hipLaunchKernelGGL(k_step, dim3(g_p), dim3(256), pwg, h->stream, dp);
    }
    HIP_TRY(hipGetLastError());
    return TOA_OK;
  }
  hipGraphExec_t exec = nullptr;
  for (int i = 0; i < h->nwgraphs; ++i) {
    const auto& w = h->wgraphs[i];
    if (w.k_init == (const void*)k_init && w.k_part == (const void*)k_part && w.k_step == (const void*)k_step && w.g_p == g_p &&
        w.g_u == g_u && w.lds == pwg && w.iters == iters)
      exec = w.exec;
  }
  if (!exec) {
    hipGraph_t graph;
    HIP_TRY(hipGraphCreate(&graph, 0));
    void* args[1] = {(void*)&dp};
    hipGraphNode_t prev = nullptr;
    auto add = [&](const void* fn, unsigned grid) -> hipError_t {
      hipKernelNodeParams kp;
      std::memset(&kp, 0, sizeof(kp));
      kp.func = const_cast<void*>(fn);
      kp.gridDim = dim3(grid);
      kp.blockDim = dim3(256);
      kp.sharedMemBytes = (unsigned)pwg;
      kp.kernelParams = args;
      hipGraphNode_t node;
      const hipError_t e = hipGraphAddKernelNode(&node, graph, prev ? &prev : nullptr, prev ? 1 : 0, &kp);
      prev = node;
      return e;
    };
    HIP_TRY(add((const void*)k_init, g_p));
    for (int it = 0; it < iters; ++it) {
      HIP_TRY(add((const void*)k_part, g_u));
      HIP_TRY(add((const void*)k_step, g_p));
    }
    HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    HIP_TRY(hipGraphDestroy(graph));
    if (h->nwgraphs < 16) h->wgraphs[h->nwgraphs++] = {(const void*)k_init, (const void*)k_part, (const void*)k_step, g_p, g_u, pwg, iters, exec};
  }
  // the parameter block was uploaded above with hipMemcpyAsync from pageable memory (staged before returning), so
  // back-to-back calls cannot race on it; the graph itself holds kernels only
  HIP_TRY(hipGraphLaunch(exec, h->stream));
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}

template <typename T, int NPAD>
inline int launch_solve(toa_handle h, int n, int64_t P, const void* H, const void* g, double scale, void* dx, int32_t* ok) {
  long long grid = (P + 3) / 4;
  const long long cap = (long long)h->num_cus * 8;
  if (grid > cap) grid = cap;
  size_t pw, pwg;
  if (int rc = lds_fit<T>(h, n, &pw, &pwg)) return rc;
  if (int rc = ensure_lds_attr(h, (const void*)solve_damped_kernel<T, NPAD>, pwg)) return rc;
  hipLaunchKernelGGL((solve_damped_kernel<T, NPAD>), dim3((unsigned)grid), dim3(256), pwg, h->stream, H, g, (long long)P, n,
                     scale, dx, ok, (int)pw);
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}
template <typename T, int NPAD>
inline int launch_inv_cov(toa_handle h, int n, int64_t P, const void* H, void* C, int32_t* ok) {
  long long grid = (P + 3) / 4;
  const long long cap = (long long)h->num_cus * 8;
  if (grid > cap) grid = cap;
  size_t pw, pwg;
  if (int rc = lds_fit<T>(h, n, &pw, &pwg)) return rc;
  if (int rc = ensure_lds_attr(h, (const void*)inv_cov_kernel<T, NPAD>, pwg)) return rc;
  hipLaunchKernelGGL((inv_cov_kernel<T, NPAD>), dim3((unsigned)grid), dim3(256), pwg, h->stream, H, (long long)P, n, C, ok, (int)pw);
  HIP_TRY(hipGetLastError());
  return TOA_OK;
}
}  // namespace toa

// ---- per-(dtype, NBM) entry points defined in inst.hip (dtag: 0 = f32, 1 = f64) ----
int toa_inst_fused(int dtag, int nbm, int thin, toa_handle h, const toa::FusedParams& prm);
int toa_inst_accumulate(int dtag, int nbm, int thin, toa_handle h, int n, int m, int64_t P, const void* data,
                        const void* x, int want_grad, void* g, void* H, double* cost, int32_t* nres);
// Gaussian-prior / sqrt2 models (inst.hip -DTOA_INST_MISC)
int toa_inst_misc_fused(int dtag, int model, int npad, toa_handle h, const toa::FusedParams& prm);
int toa_inst_misc_accumulate(int dtag, int model, int npad, toa_handle h, int n, int m, int64_t P, const void* data,
                             const void* x, int want_grad, void* g, void* H, double* cost, int32_t* nres);
int toa_inst_wide(int dtag, int model, int nbm, int thin, toa_handle h, const toa::FusedParams& prm, int splits);
int toa_inst_inv_cov(int dtag, int npad, toa_handle h, int n, int64_t P, const void* H, void* C, int32_t* ok);
int toa_inst_solve(int dtag, int npad, toa_handle h, int n, int64_t P, const void* H, const void* g, double scale,
                   void* dx, int32_t* ok);
#endif  // !__HIPCC_RTC__
