// Device residual models, part 1: the DenseRow family (BASELINE C2-C5: r_i = a_i.x + 0.1 sin(a_i.x) - b_i) on the MFMA Gram of
// dense_row.hpp, and the cooperative tail's per-wave control blocks.  Split out of kernels.hpp in round 6 (one header per
// kernel family; kernels.hpp includes them all in order).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

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

}  // namespace toa
