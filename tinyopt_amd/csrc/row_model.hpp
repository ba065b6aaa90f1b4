// Row models for WIDE parameter blocks (13 <= kN <= 63): a user's residual — with its own Jacobian rows (the reference's manual
// Accumulate callback, docs/API.md:37-57, benchmarks/dense.cpp:57-66,90-99) or differentiated on the device (the AD bridge,
// diff/optimize_autodiff.h:91-166) — on the SAME matrix-core Gram, LDL^T and state machine as the hand-derived DenseRowModel.
//
// Round 2-5 evaluated such a functor on the 16 lanes of a row group at once (each lane a chunk of the Jet): 16 x the value part,
// no room for a hand-written Jacobian, 0.022 of the HBM roof at n = 50.  Here a ROW IS A LANE:
//
//   HBM -> LDS         the items of a super-step (IT items, one per lane; IT * kD contiguous scalars) by LDS-DMA
//                      (`buffer_load_dwordx4 ... lds`: lane-linear, 1 KiB per instruction, NO register in flight), issued one
//                      super-step AHEAD into the other of two regions of the wave's stage
//   lane i reads ITEM i  (its kD scalars; x from the wave's carve, a broadcast read): the functor runs once per item, on plain T
//                      with the user's J, or on Jet<T, CW> chunk by chunk (AdRowFunctor below)
//   [J | r] -> LDS     row rho of the super-step at rho * RSP (RSP odd: the writer is a row, conflict-free), over the raw items
//                      of the same region (every lane has consumed its item by then: the LDS operations of a wave execute in
//                      program order)
//   LDS -> MFMA        lane (k, c) of step s reads its NBM main columns and the thin columns of row 4 s + k: DenseRowGram's
//                      operand layout, straight into add_step
//
// so the functor costs 1 / 16 of what it did and the Gram gets exactly the operands of the compiled-in family.  The stage is
// laid over the part of the wave's carve that is dead while a pass runs (WaveLds::pass_dead_bytes), like pass16s's.
// Why LDS-DMA and not the register ring of the compiled-in passes: a load in flight into a VGPR is invisible to hipcc, which is
// free to copy that register before the data has landed whenever the USER's functor raises the register pressure (seen at once:
// fp64, n = 50 — tools/isa_lint.py on the run-time compiler's output).  A library cannot lint code it has not seen; a prefetch
// that owns no register cannot be broken by it.
#pragma once
#ifndef TOA_ROW_SINGLE
#define TOA_ROW_SINGLE 1   // accumulate passes use ONE region of the stage (fetch, wait, compute: every lane an item) for every functor;
                           // 0 = the ring for functors that are not compute-bound (A/B: C4 as text 11.4 -> 10.8 ms with 1, profiles/r06_ab_log.md)
#endif
#ifndef TOA_ROW_XREGS
#define TOA_ROW_XREGS 0   // A/B arm: x as a register array (50 VGPRs at n = 50: the fused kernel then keeps ONE workgroup per compute unit)
#endif
#ifndef TOA_ROW_ABL
#define TOA_ROW_ABL 0   // ablation arms (tools/row_model_probe.py through $TOA_JIT_FLAGS): 1 no functor, 2 no Gram steps, 4 no [J | r] image
#endif

namespace toa {

// Geometry of a row model's super-step: a function of (sizeof(T), kN, kR, kD) only, evaluated by the device templates and by
// the host launchers (jit.hip, kernels.hpp) alike.
struct RowStageGeom {
  int items2;  // items per super-step, one per lane, when the stage is a RING of `nbuf` regions (the items of the next nbuf - 1
               // super-steps are on their way while this one is consumed): memory-bound passes — a functor with the user's own
               // Jacobian, every cost-only pass
  int nbuf;    // regions of that ring
  int items1;  // ... when it is ONE region (fetch, wait, compute): the accumulate pass of a compute-bound functor (AD), which
               // wants every lane busy more than it wants the prefetch (0 = the model has no such pass)
  int ps;      // item stride in the raw image, elements (>= kD)
  int rsp;     // row stride of the [J | r] image, elements (>= the packed row)
  int bytes;   // the wave's LDS stage
  int table_off;   // where the model's own `extra` bytes begin, behind the regions (AdRowFunctor on a manifold: the Jets of x (+) d)
  // Both images are read and written a ROW (an item) PER LANE: a stride of 16 bytes x an odd number makes every 16-byte access of
  // sixteen consecutive lanes hit sixteen different bank quads (ds_read_b128 / ds_write_b128 without conflicts) and keeps every
  // row 16-byte aligned.
  static __host__ __device__ constexpr int stride16(int elems, int sz) {
    int q = (elems * sz + 15) / 16;
    if (!(q & 1)) ++q;
    return q * 16 / sz;
  }
  // bytes of one region for `it` items: their raw image, then the [J | r] image over it
  static __host__ __device__ constexpr int region(int it, int ps, int rsp, int kR, int sz) {
    const int raw = it * ps * sz, img = ((it * kR + 3) & ~3) * rsp * sz;
    return (raw > img ? raw : img);
  }
  // What a wave's stage may take.  With `waves` = 2 workgroups of four waves on a compute unit a wave has 160 KiB / 8 = 20 KiB of
  // LDS, with 3 (twelve waves) 13.3 KiB; the part of its carve that is alive during a pass (x, g, the diagonal, the last step, the
  // memo's x, the state and option blocks) takes 5 x 64 scalars + ~0.5 KiB of that.
  static __host__ __device__ constexpr int budget(int sz, int waves) { return 160 * 1024 / (4 * waves) - 320 * sz - 512; }
  // Ring depth (TOA_ROW_NBUF) and resident workgroups per compute unit the geometry is sized for (TOA_ROW_WAVES): A/B arms of
  // the run-time build (profiles/r06_ab_log.md); jit.hip reads the same two values out of $TOA_JIT_FLAGS for its LDS sizing.
#ifndef TOA_ROW_NBUF
#define TOA_ROW_NBUF 2
#endif
#ifndef TOA_ROW_WAVES
#define TOA_ROW_WAVES 2
#endif
  static __host__ __device__ constexpr RowStageGeom make(int sz, int n, int kR, int kD, bool compute_bound, int nbuf = TOA_ROW_NBUF,
                                                        int waves = TOA_ROW_WAVES, int extra = 0) {
    const int rem = n & 15;
    const bool thin = n >= 16 && rem + 1 <= 4;
    const int nbm = thin ? (n >> 4) : (n + 16) / 16;
    const int rs = thin ? 16 * nbm + rem + 1 : nbm * ((n + nbm) / nbm);
    RowStageGeom g{};
    g.rsp = stride16(rs, sz);
    g.ps = kD > 0 ? stride16(kD, sz) : 0;
    const int itmax = 64 / kR;           // rows of a super-step <= 64
    const int bud = budget(sz, waves) - extra;
    g.nbuf = nbuf;
    int it = itmax;                      // the largest multiple of 4 (one Gram step = four rows) whose ring fits; else 2, 1
    if (it >= 4) it &= ~3;
    while (it > 1 && g.nbuf * region(it, g.ps, g.rsp, kR, sz) > bud) it = it > 4 ? it - 4 : it >> 1;
    g.items2 = it;
    g.bytes = g.nbuf * region(it, g.ps, g.rsp, kR, sz);
    g.items1 = 0;
    if (compute_bound) {
      it = itmax;
      if (it >= 4) it &= ~3;
      while (it > 1 && region(it, g.ps, g.rsp, kR, sz) > bud) it = it > 4 ? it - 4 : it >> 1;
      g.items1 = it;
      if (region(it, g.ps, g.rsp, kR, sz) > g.bytes) g.bytes = region(it, g.ps, g.rsp, kR, sz);
    }
    g.bytes = (g.bytes + 15) & ~15;
    g.table_off = g.bytes;
    g.bytes += (extra + 15) & ~15;
    return g;
  }
};

// Forward-mode AD as a manual-Jacobian functor: F::eval, written once over the scalar type (docs/API.md:21-35), is run on
// Jet<T, CW> once per CHUNK of CW parameters — seeds x_jet[j].v[j - c0] = 1 (optimize_autodiff.h:56-69) — and J.row = res.v
// (:127-148) lands chunk by chunk in the row's registers.  The chunk index is a compile-time constant, so the seeds are
// wave-uniform compares of the parameter index: scalar-unit work.  Cost-only passes run F::eval on plain T (grad == nullptr).
// one(b): T(1) if b else T(0), formed on the SCALAR unit for a wave-uniform b (a float select of uniform operands is still a
// v_cndmask_b32 to hipcc — five times the issue cost of an FMA on this hardware, profiles/r04_issue_probe.txt — so the bits are
// selected as an integer and read back as a scalar: the seed then enters the Jet arithmetic as the SGPR operand of a v_fmac)
__device__ __forceinline__ float seed_one(bool b, float) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(b ? 0x3f800000 : 0));
}
__device__ __forceinline__ double seed_one(bool b, double) {
  const unsigned hi = unsigned(__builtin_amdgcn_readfirstlane(b ? 0x3ff00000 : 0));
  return __builtin_bit_cast(double, (unsigned long long)hi << 32);
}
template <typename T, int CW>
struct ChunkSeededX {
  const T* xs;   // the parameters (the wave's carve)
  int c0;        // first parameter of the chunk
  __device__ __forceinline__ Jet<T, CW> operator[](int j) const {
    Jet<T, CW> r;
    r.a = xs[j];
#pragma unroll
    for (int s = 0; s < CW; ++s) r.v[s] = seed_one(j == c0 + s, T(0));
    return r;
  }
};
// Seeds of the tangent for one chunk, d = 0: d[a] as a Jet (optimize_autodiff.h:48-77) — c0 is a run-time value here (the chunk is
// the LANE that builds the table below; once per pass)
template <typename T, int CW>
struct ChunkSeededD {
  int c0;
  __device__ __forceinline__ Jet<T, CW> operator[](int a) const {
    Jet<T, CW> r;
    r.a = T(0);
#pragma unroll
    for (int s = 0; s < CW; ++s) r.v[s] = (a == c0 + s) ? T(1) : T(0);
    return r;
  }
};
// MANIFOLD == 2 (TOA_MANIFOLD_USER; round 6): the parameters live in the user's container of F::kX stored scalars and the
// derivative is taken through x (+) d at d = 0.  x (+) d does not depend on the item, so its Jets — per chunk, kX of them — are
// formed ONCE per accumulate pass (lane c builds chunk c, straight into LDS) and every lane's functor reads them from that table
// (the same address in every lane: a broadcast read).  Cost-only passes evaluate F on the stored scalars themselves.
template <typename T, typename F, int MANIFOLD = 0>
struct AdRowFunctor {
  static constexpr int kN = F::kN, kR = F::kR, kD = F::kD, kH = F::kH;
  static_assert(MANIFOLD != 1 || F::kN == 6, "TOA_MANIFOLD_SE3: one pose, six tangent dimensions");
  static constexpr int kX = MANIFOLD == 1 ? 12 : (MANIFOLD == 2 ? FunctorX<F>::value : F::kN);   // stored scalars of x
  template <class S, class XA_, class DA_, class XP_>
  static __device__ __forceinline__ void plus(const XA_& x, const DA_& d, XP_&& xp) { F::template plus<S>(x, d, xp); }
  static constexpr bool kManual = true;
  static constexpr bool kComputeBound = true;                  // (RowModel: one LDS region, every lane an item — the Jets are the bound)
  static constexpr bool kIndexedOperands = true;               // x[j] / p[j] with a RUNNING j: straight from LDS (a register array indexed
                                                               // by a loop counter is scratch memory)
#ifndef TOA_AD_CW
#define TOA_AD_CW 12
#endif
  static constexpr int kChunks = (kN + TOA_AD_CW - 1) / TOA_AD_CW;   // Jets of <= 12 partials, as balanced as kN allows
  static constexpr int kCW = (kN + kChunks - 1) / kChunks;
  static constexpr bool kTable = MANIFOLD != 0;
  static constexpr int kTableBytes = kTable ? ((kChunks * kX * (kCW + 1) * int(sizeof(T)) + 15) & ~15) : 0;
  using TabJet = Jet<T, kCW>;
  static __device__ __forceinline__ void build_table(const T* xs, TabJet* tab, const int lane) {
    if constexpr (MANIFOLD == 1) {   // ONE SE3 pose (R row-major, t): the right perturbation pose * exp(d) at d = 0 (optimize_autodiff.h:48-55, 73-77)
      if (lane == 0) {
        T xr[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) xr[i] = xs[i];
        TabJet xj[12];
        se3_seed_pose<T>(xr, xj);
#pragma unroll
        for (int i = 0; i < 12; ++i) tab[i] = xj[i];
      }
    } else {
      if (lane < kChunks) {
        const ChunkSeededD<T, kCW> D{lane * kCW};
        F::template plus<TabJet>(xs, D, tab + lane * kX);
      }
    }
    wave_sync();
  }
  template <bool want_grad>
  static __device__ __forceinline__ void eval_manual_tab(const T* x, const TabJet* tab, const T* h, const T* p, T* r, T (*J)[kN]) {
    if constexpr (!want_grad) {
      F::template eval<T>(x, h, p, r);
    } else {
      static_for<kChunks>([&](auto cc) __attribute__((always_inline)) {
        constexpr int c0 = decltype(cc)::value * kCW;
        const TabJet* X = tab + decltype(cc)::value * kX;
        TabJet rr[kR];
        F::template eval<TabJet>(X, h, p, rr);
#pragma unroll
        for (int q = 0; q < kR; ++q) {
          if constexpr (c0 == 0) r[q] = rr[q].a;
#pragma unroll
          for (int s = 0; s < kCW; ++s)
            if (c0 + s < kN) J[q][c0 + s] = rr[q].v[s];
        }
      });
    }
  }
  template <bool want_grad>
  static __device__ __forceinline__ void eval_manual(const T* x, const T* h, const T* p, T* r, T (*J)[kN]) {
    if constexpr (!want_grad) {
      F::template eval<T>(x, h, p, r);
    } else {
      static_for<kChunks>([&](auto cc) __attribute__((always_inline)) {
        constexpr int c0 = decltype(cc)::value * kCW;
        ChunkSeededX<T, kCW> X{x, c0};
        Jet<T, kCW> rr[kR];
        F::template eval<Jet<T, kCW>>(X, h, p, rr);
#pragma unroll
        for (int q = 0; q < kR; ++q) {
          if constexpr (c0 == 0) r[q] = rr[q].a;
#pragma unroll
          for (int s = 0; s < kCW; ++s)
            if (c0 + s < kN) J[q][c0 + s] = rr[q].v[s];
        }
      });
    }
  }
};

template <typename F, typename = void>
struct FunctorTable { static constexpr int bytes = 0; };
template <typename F>
struct FunctorTable<F, std::enable_if_t<F::kTable>> { static constexpr int bytes = F::kTableBytes; };

// MANIFOLD: 0 = Euclidean, 1 = ONE SE3 pose (12 stored scalars, six tangent dimensions), 2 = the user's container (TOA_MANIFOLD_USER:
// F::kX stored scalars, F::plus; round 6).  A functor with
// its own Jacobian fills J over the TANGENT (kN columns) from the stored scalars; an AD functor goes through AdRowFunctor's table.
// ROBUST = false: without the M-estimator branch of the passes (see JetModel; the launchers pick it for plain L2 solves only).
template <typename T, int NBM, int THIN, typename F, int MANIFOLD = 0, bool ROBUST = true>
struct RowModel {
  using Scalar = T;
  static_assert(MANIFOLD >= 0 && MANIFOLD <= 2, "TOA_MANIFOLD_*");
  static_assert(MANIFOLD != 1 || F::kN == 6, "TOA_MANIFOLD_SE3: one pose, six tangent dimensions");
  static constexpr int kXdim = MANIFOLD == 1 ? 12 : (MANIFOLD == 2 ? FunctorX<F>::value : 0);
  static_assert(kXdim <= 64, "stored scalars of x: one per lane");
  static constexpr int kTableBytes = FunctorTable<F>::bytes;
  static constexpr int kN = F::kN, kR = F::kR, kD = F::kD;
  static constexpr int kNmax = THIN > 0 ? 16 * NBM + THIN - 1 : 16 * NBM - 1;
  static constexpr int kNpad = (kNmax + 7) & ~7;
  static_assert(FunctorManual<F>::value, "a row model evaluates eval_manual (wrap a residual functor in AdRowFunctor)");
  static_assert(kR >= 1 && kR <= 8, "residuals per item");
  static_assert(kN <= kNmax && (THIN == 0 || kN == kNmax), "functor and layout disagree");
  // the packed-row geometry of DenseRowLayout::make(kN, .), as constants
  static constexpr int kNmr = THIN ? 16 * NBM : kN;
  static constexpr int kRsm = THIN ? 16 * NBM : NBM * ((kN + NBM) / NBM);
  static constexpr int kRs = kRsm + THIN;
  static constexpr RowStageGeom kGeom = RowStageGeom::make(int(sizeof(T)), kN, kR, kD, FunctorComputeBound<F>::value || TOA_ROW_SINGLE, TOA_ROW_NBUF, TOA_ROW_WAVES, kTableBytes);
  static_assert(kGeom.items2 >= 1 && (kTableBytes == 0 || kGeom.items1 >= 4), "the Jets of x (+) d leave no room for the items in the wave's LDS stage");
  static constexpr int PS = kGeom.ps, RSP = kGeom.rsp;
  static constexpr size_t kStageBytes = size_t(kGeom.bytes);
  static_assert(RSP >= kRs, "geometry");
  // x and the item as REGISTER arrays when they are small (the functor's loops over them unroll: no LDS traffic in its
  // arithmetic); straight from LDS otherwise, and for functors that index them with a running index
  static constexpr bool kXRegs = TOA_ROW_XREGS && MANIFOLD == 0 && !FunctorIndexed<F>::value && kN * int(sizeof(T)) <= 256;
  static constexpr bool kPRegs = !FunctorIndexed<F>::value && kD * int(sizeof(T)) <= 256;

  __device__ __forceinline__ void plus_eq(WaveLds<T>& L, const T* dv, T sign, int n, int lane) const {
    if constexpr (MANIFOLD == 1) Se3Manifold<T>::plus_eq(L, dv, sign, n, lane);
    else if constexpr (MANIFOLD == 2) UserManifoldOf<T, F>::plus_eq(L, dv, sign, n, lane);
    else euclid_plus_eq(L, dv, sign, lane);
  }
  DenseRowGram<T, NBM, THIN> gram;
  const T* data;
  const T* d;            // the bound problem: [kH header scalars | items x kD]
  DenseRowLayout lay;
  int m;                 // residual ROWS of a problem: items x kR (an item's kR residuals are kR consecutive rows)
  int it0, it1;          // the items this model works on: all of the problem's, or one chunk of the row-split form
  int loss;              // TOA_LOSS_* on each ITEM's squared residual norm (toa_set_loss; robust_norms.h:20-26); 0 = plain L2
  T th2;
  int ninl;              // inlier residuals of the last pass; -1 = all of them (no loss)
  unsigned char* stage;  // this wave's LDS stage
  __device__ __forceinline__ void init(int n, int m_, const void* dp) {
    m = m_;
    lay = DenseRowLayout::make(n, m_);
    data = static_cast<const T*>(dp);
    it0 = 0; it1 = m_ / kR;
    loss = TOA_LOSS_L2; th2 = T(0); ninl = -1;
    stage = nullptr;
  }
  __device__ __forceinline__ void set_loss(int kind, double t2) { loss = kind; th2 = T(t2); }
  __device__ __forceinline__ void bind(long long p) {
    const size_t stride = FunctorPackedRows<F>::value ? lay.elems_per_problem() : F::kH + size_t(m / kR) * kD;
    d = data + size_t(p) * stride;
    it0 = 0; it1 = m / kR;
  }
  // row-split execution: rows [r0, r0 + rows) of problem p — r0 on an item boundary (a multiple of lcm(16, kR): jit.hip)
  __device__ __forceinline__ void bind_chunk(long long p, int r0, int rows, int) {
    bind(p);
    it0 = r0 / kR;
    it1 = min(m / kR, (r0 + rows) / kR);
    if (it1 < it0) it1 = it0;
  }

  // One 1 KiB piece (V) of the raw items of a super-step of IT items -> the region at LDS byte address `lds`: lane l's 16 bytes
  // land at lds + V * 1024 + l * 16.  The image is PADDED (item i at i * PS elements) while the DMA writes lane-linear, so the
  // padding is made on the SOURCE side: LDS byte o belongs to item o / (PS bytes) at byte o % (PS bytes) of it, and the lane
  // fetches that item's bytes from the item's place in memory (a 16-byte piece never straddles two items: PS bytes is a multiple
  // of 16; its tail may read the first bytes of the next item into the padding).  Bounds: the descriptor ends with the pass's
  // items — beyond them zeros arrive, touching no memory.  M0 (the DMA's LDS base) is written in the statement that uses it and
  // put back.
  static constexpr unsigned kItemBytes = unsigned(kD) * unsigned(sizeof(T)), kPsBytes = unsigned(PS) * unsigned(sizeof(T));
  template <int IT, int V>
  static __device__ __forceinline__ void dma_piece(const i32x4 rsrc, const int lane, const unsigned soff, const unsigned lds) {
    constexpr unsigned kRawBytes = unsigned(IT) * kPsBytes;
    const unsigned o = unsigned(V) * 1024u + unsigned(lane) * 16u;           // this lane's byte of the image
    const unsigned it = o / kPsBytes, w = o - it * kPsBytes;
    const unsigned voff = (w < kItemBytes) ? it * kItemBytes + w : 0x80000000u;   // (a piece of pure padding fetches nothing)
    const unsigned la = unsigned(__builtin_amdgcn_readfirstlane(int(lds + unsigned(V) * 1024u)));
    unsigned keep;
    if ((V + 1) * 1024u <= kRawBytes || o < kRawBytes)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(la), "s"(soff) : "memory");
  }
  template <int IT, int V0, int V1>   // pieces V0 <= V < V1
  static __device__ __forceinline__ void dma_issue(const i32x4 rsrc, const int lane, const unsigned soff, const unsigned lds) {
    if constexpr (V0 < V1) {
      dma_piece<IT, V0>(rsrc, lane, soff, lds);
      dma_issue<IT, V0 + 1, V1>(rsrc, lane, soff, lds);
    }
  }
  // 16-byte LDS accesses of a row that starts on a 16-byte boundary (the strides of RowStageGeom)
  static constexpr int EV = 16 / int(sizeof(T));
  typedef T TV __attribute__((ext_vector_type(16 / sizeof(T))));
  template <int N>
  static __device__ __forceinline__ void lds_read_row(T (&dst)[N ? N : 1], const T* src) {
    const TV* sv = reinterpret_cast<const TV*>(src);
#pragma unroll
    for (int q = 0; q < N / EV; ++q) {
      const TV t = sv[q];
#pragma unroll
      for (int e = 0; e < EV; ++e) dst[q * EV + e] = t[e];
    }
#pragma unroll
    for (int j = N / EV * EV; j < N; ++j) dst[j] = src[j];
  }
  // value of position `pos` of the packed row [J | r] (DenseRowLayout::pos_col / pos_b)
  template <int POS>
  static __device__ __forceinline__ T row_value(const T (&Jq)[kN], const T rq) {
    if constexpr (THIN > 0) {
      if constexpr (POS < kRsm) return Jq[POS];
      else if constexpr (POS - kRsm < THIN - 1) return Jq[kNmr + POS - kRsm];
      else return rq;
    } else {
      if constexpr (POS < kN) return Jq[POS];
      else if constexpr (POS == kRsm - 1) return rq;
      else return T(0);
    }
  }
  template <int Q>
  static __device__ __forceinline__ void lds_write_vecs(TV* dv, const T (&Jq)[kN], const T rq) {
    if constexpr (Q * EV < kRs) {
      TV t;
      static_for<EV>([&](auto ec) __attribute__((always_inline)) {
        constexpr int e = decltype(ec)::value;
        if constexpr (Q * EV + e < kRs) t[e] = row_value<Q * EV + e>(Jq, rq);
        else t[e] = T(0);
      });
      dv[Q] = t;
      lds_write_vecs<Q + 1>(dv, Jq, rq);
    }
  }
  static __device__ __forceinline__ void lds_write_row(T* row, const T (&Jq)[kN], const T rq) { lds_write_vecs<0>(reinterpret_cast<TV*>(row), Jq, rq); }
  static __device__ __forceinline__ void lds_zero_row(T* row) {
    TV* dv = reinterpret_cast<TV*>(row);
    TV z;
#pragma unroll
    for (int e = 0; e < EV; ++e) z[e] = T(0);
#pragma unroll
    for (int q = 0; q < (kRs + EV - 1) / EV; ++q) dv[q] = z;
  }

  // One pass over the bound items, IT of them per super-step, the stage used as NBUF regions.  WANT_H: the Gram of [J | r]
  // (K1) — returns the sum of the losses when a loss is set, 0 otherwise (the cost is then the Gram's (r, r) entry).
  // !WANT_H: cost only (K2).
  template <bool WANT_H, int IT, int NBUF>
  __device__ __forceinline__ T pass(const T* __restrict__ xs, const int lane_in) {
    constexpr int ROWS = IT * kR, SPS = (ROWS + 3) / 4;   // rows / Gram steps of a super-step
    constexpr int kRegion = RowStageGeom::region(IT, PS, RSP, kR, int(sizeof(T)));
    constexpr unsigned kSsBytes = unsigned(IT) * kItemBytes;     // a super-step's items in memory
    constexpr int PFV = (IT * int(kPsBytes) + 1023) / 1024;
    static_assert(ROWS <= 64 && kRegion % 16 == 0 && size_t(NBUF * kRegion) <= kStageBytes, "geometry");
    // A/B arm, measured and NOT the default (-DTOA_ROW_DMA_SPREAD=1): the next super-step's pieces issued BETWEEN this one's Gram
    // steps instead of in front of the functor (an LDS-DMA piece issued into a full queue stalls its wave; behind a step's
    // matrix-core instructions the queue has drained — MI355X_MICROARCH.md).  C4 as text, 12 500 problems: 13.56 -> 15.24 ms: the
    // pieces then have half a super-step to land instead of a whole one, and the wait in front of the next functor costs more
    // than the issue stalls saved.
#ifndef TOA_ROW_DMA_SPREAD
#define TOA_ROW_DMA_SPREAD 0
#endif
    constexpr bool kSpread = TOA_ROW_DMA_SPREAD && WANT_H && NBUF >= 2;
    int lane = lane_in;
    asm volatile("" : "+v"(lane));   // keep the per-lane addresses out of LICM's reach (see DenseRowGram::extract_g_diag_cost)
    const int k = lane >> 4, c = lane & 15;
    const bool robust = ROBUST && loss != TOA_LOSS_L2;   // wave-uniform
    const int nit = it1 - it0;
    const int nss = (nit + IT - 1) / IT;
    const T* const items = d + F::kH + size_t(it0) * kD;
    const i32x4 rsrc = make_rsrc(items, unsigned(nit) * unsigned(kD) * unsigned(sizeof(T)));
    unsigned char* const stg = static_cast<unsigned char*>(__builtin_assume_aligned(stage, 16));
    const unsigned lds0 = unsigned(reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned char*)(stg)));
    if (WANT_H) gram.clear();
    T csum = 0, inl = 0;
    // x: wave-uniform, once per pass
    T xl[kXRegs ? kN : 1];
    if constexpr (kXRegs) {
#pragma unroll
      for (int j = 0; j < kN; ++j) xl[j] = xs[j];
    }
    const T* const xp = kXRegs ? xl : xs;
    if constexpr (kTableBytes > 0 && WANT_H) F::build_table(xs, reinterpret_cast<typename F::TabJet*>(stg + kGeom.table_off), lane);
    if constexpr (NBUF >= 2) {   // prologue: the first NBUF - 1 super-steps start towards their regions
      static_for<NBUF - 1>([&](auto bc) __attribute__((always_inline)) {
        constexpr int b = decltype(bc)::value;
        dma_issue<IT, 0, PFV>(rsrc, lane, unsigned(__builtin_amdgcn_readfirstlane(int(unsigned(b) * kSsBytes))), lds0 + unsigned(b * kRegion));
      });
    }
    const bool active = THIN > 0 || c * NBM < kRsm;
    int cur = 0;   // byte offset of the region of super-step ss in the stage
    for (int ss = 0; ss < nss; ++ss) {
      unsigned nxt_soff = 0, nxt_lds = 0;
      if constexpr (NBUF >= 2) {
        // ---- this super-step's region has landed — the pieces of the NBUF - 2 younger super-steps may stay in flight (vector
        // memory operations retire in order) — and super-step ss + NBUF - 1 starts into the region of ss - 1, whose last readers
        // (its Gram steps) have their operands: lgkmcnt
        const int prv = cur == 0 ? (NBUF - 1) * kRegion : cur - kRegion;
        nxt_soff = unsigned(__builtin_amdgcn_readfirstlane(int(unsigned(ss + NBUF - 1) * kSsBytes)));
        nxt_lds = lds0 + unsigned(__builtin_amdgcn_readfirstlane(prv));
        static_assert((NBUF - 2) * PFV < 64, "vmcnt is a 6-bit counter");
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"((NBUF - 2) * PFV) : "memory");
        if constexpr (!kSpread) dma_issue<IT, 0, PFV>(rsrc, lane, nxt_soff, nxt_lds);
      } else {
        // ---- one region: fetch, wait (the SIMD's other waves cover the round trip of a functor that is the bound anyway)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        dma_issue<IT, 0, PFV>(rsrc, lane, unsigned(__builtin_amdgcn_readfirstlane(int(unsigned(ss) * kSsBytes))), lds0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      wave_sync();
      T* const S = reinterpret_cast<T*>(stg + cur);
      // operand addresses of the Gram steps: lane (k, c) reads row 4 s + k
      const T* const wrow = S + k * RSP + (active ? c * NBM : 0);
      const T* const vrow = S + k * RSP + kRsm;
      // ---- the functor, one item per lane
      const int item = ss * IT + lane;
      const bool valid = lane < IT && item < nit;
      T rv[kR];
      T Jv[kR][kN];
      T n2 = 0;
      if (valid) {
        const T* const pi = S + size_t(lane) * PS;
        T pl[kPRegs ? (kD ? kD : 1) : 1];
        if constexpr (kPRegs) lds_read_row<kD>(pl, pi);
        const T* const p = kPRegs ? pl : pi;
#if (TOA_ROW_ABL & 1)   // ablation: no functor — the item's first scalars stand in for the row
        (void)xp;
#pragma unroll
        for (int q = 0; q < kR; ++q) {
          rv[q] = p[kD - 1];
          if constexpr (WANT_H) {
#pragma unroll
            for (int a = 0; a < kN; ++a) Jv[q][a] = p[a % (kD ? kD : 1)];
          }
        }
#else
        if constexpr (kTableBytes > 0) {
          if constexpr (WANT_H) F::template eval_manual_tab<true>(xp, reinterpret_cast<const typename F::TabJet*>(stg + kGeom.table_off), d, p, rv, Jv);
          else F::template eval_manual_tab<false>(xp, nullptr, d, p, rv, static_cast<T(*)[kN]>(nullptr));
        } else {
          if constexpr (WANT_H) F::template eval_manual<true>(xp, d, p, rv, Jv);
          else F::template eval_manual<false>(xp, d, p, rv, static_cast<T(*)[kN]>(nullptr));
        }
#endif
#pragma unroll
        for (int q = 0; q < kR; ++q) n2 += rv[q] * rv[q];
      }
      T sq = T(1);
      if (robust) {   // the item's ||r||^2 through the M-estimator: cost += l, its rows of [J | r] scaled by sqrt(s)
        T l, sc;
        robust_norm(loss, n2, th2, l, sc);
        csum += valid ? l : T(0);
        inl += (valid && n2 <= th2) ? T(kR) : T(0);
        sq = r_sqrt(sc);
      } else if constexpr (!WANT_H) {
        csum += n2;
      }
      if constexpr (!WANT_H) wave_sync();   // (the region is overwritten two super-steps on: not before every lane has read its item)
      if constexpr (WANT_H) {
        // ---- [J | r] over the raw items (every lane has read its own by now): 16-byte stores, a row per lane; the lanes of a
        // last, partial super-step that have no item write zero rows
        wave_sync();
#if (TOA_ROW_ABL & 4)
        if (false) {
#else
        if (valid) {
#endif
#pragma unroll
          for (int q = 0; q < kR; ++q) {
            if (robust) {
              rv[q] *= sq;
#pragma unroll
              for (int a = 0; a < kN; ++a) Jv[q][a] *= sq;
            }
            lds_write_row(S + size_t(lane * kR + q) * RSP, Jv[q], rv[q]);
          }
        } else if (lane < IT) {
#pragma unroll
          for (int q = 0; q < kR; ++q) lds_zero_row(S + size_t(lane * kR + q) * RSP);
        }
        wave_sync();
        // ---- the super-step's Gram steps, their operands read a batch ahead of the matrix core
#ifndef TOA_ROW_KB
#define TOA_ROW_KB 4
#endif
        constexpr int kBatch = SPS < TOA_ROW_KB ? SPS : TOA_ROW_KB;
        constexpr int kNB = (SPS + kBatch - 1) / kBatch;
        static_for<kNB>([&](auto bc) __attribute__((always_inline)) {
          constexpr int b = decltype(bc)::value;
          constexpr int s0 = b * kBatch;
          constexpr int nb = SPS - s0 < kBatch ? SPS - s0 : kBatch;
          T w[nb][NBM], v[nb][THIN ? THIN : 1];
          static_for<nb>([&](auto sc) __attribute__((always_inline)) {
            constexpr int s = s0 + decltype(sc)::value;
            constexpr bool partial = 4 * s + 3 >= ROWS;   // (only when a super-step has fewer than four rows left: IT * kR < 4)
#pragma unroll
            for (int cb = 0; cb < NBM; ++cb) {
              const T t = wrow[s * 4 * RSP + cb];
              w[s - s0][cb] = (active && (!partial || 4 * s + k < ROWS)) ? t : T(0);
            }
#pragma unroll
            for (int j = 0; j < THIN; ++j) {
              const T t = vrow[s * 4 * RSP + j];
              v[s - s0][j] = (!partial || 4 * s + k < ROWS) ? t : T(0);
            }
          });
#if (TOA_ROW_ABL & 2)   // ablation: no Gram steps (the operands are read and summed into the cost accumulator)
          static_for<nb>([&](auto sc) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value;
#pragma unroll
            for (int cb = 0; cb < NBM; ++cb) csum += w[s][cb];
#pragma unroll
            for (int j = 0; j < THIN; ++j) csum += v[s][j];
          });
#else
          static_for<nb>([&](auto sc) __attribute__((always_inline)) {
            constexpr int s = s0 + decltype(sc)::value;
            // (the LAST step of every super-step waits the matrix pipe out inside its own asm statement — ~32 cycles in ~3 000:
            //  between two super-steps runs the user's functor, under whose register pressure hipcc may move an accumulator,
            //  and a run-time build cannot be linted for accumulator reads that overtake the matrix core)
            if constexpr (s + 1 == SPS) gram.template add_step<true>(w[s - s0], v[s - s0], 1);
            else gram.template add_step<false>(w[s - s0], v[s - s0], 0);
          });
#endif
          if constexpr (kSpread) {   // this batch's share of the next super-step's pieces, behind its matrix-core work
            constexpr int v0 = PFV * b / kNB, v1 = PFV * (b + 1) / kNB;
            dma_issue<IT, v0, v1>(rsrc, lane, nxt_soff, nxt_lds);
          }
        });
        wave_sync();
      }
      if constexpr (NBUF >= 2) cur = __builtin_amdgcn_readfirstlane(cur + kRegion == NBUF * kRegion ? 0 : cur + kRegion);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the piece past the end (zeros) has landed before anybody else writes there
    ninl = robust ? int(wave_allreduce_sum(inl)) : -1;   // exact in T: one count per residual
    if constexpr (WANT_H) {
      gram.finish_steps();
      return robust ? wave_allreduce_sum(csum) : T(0);
    }
    return wave_allreduce_sum(csum);
  }

  __device__ __forceinline__ void accumulate(WaveLds<T>& L, int n, int lane, T& cost, int& nres) {
    T cl;
    if constexpr (kGeom.items1 > 0) cl = pass<true, kGeom.items1, 1>(L.xs, lane);
    else cl = pass<true, kGeom.items2, kGeom.nbuf>(L.xs, lane);
    cost = gram.extract_g_diag_cost(L.g, L.hd, lay, n, lane, L.tmp);
    if (ROBUST && loss != TOA_LOSS_L2) cost = cl;   // sum of the robust losses, not the Gram's r^T r (which is scaled by s)
    nres = m;
  }
  __device__ __forceinline__ void evaluate(WaveLds<T>& L, int, int lane, T& cost, int& nres) {
    cost = pass<false, kGeom.items2, kGeom.nbuf>(L.xs, lane);
    nres = m;
  }
  template <typename O>
  __device__ __forceinline__ void write_sym(O* M, int LD, int n, int lane) const { gram.write_sym(M, LD, lay, n, lane); }
  // memo of the last accepted linearisation (lm_device.hpp)
  static constexpr bool kMemo = true;
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

}  // namespace toa
