// Device residual models, part 4: automatic differentiation on the device for narrow parameter blocks (JetModel, n <= 12: an
// item per lane, the Gram in registers), user manifolds as text, the functor traits, the built-in functors.  Ends by including
// row_model.hpp (13 <= n <= 63: a row per lane, staged into the MFMA Gram).
#pragma once
#include "models_se3.hpp"

namespace toa {

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
    if constexpr (kXdim > 32 || F::kN > 12) {
      // wide containers (row models, round 6): x, +-delta and x (+) delta as REGISTER arrays of one lane would be ~190 registers —
      // the factorisation workspace is dead by now (the step has been solved for; n >= 13: n (n | 1) >= 169 scalars), so delta
      // goes to M[0 .. kN), x (+) delta to M[64 ..), and the user's text indexes LDS
      T* dd = L.M;
      T* xn = L.M + 64;
      if (lane < F::kN) dd[lane] = sign * dv[lane];
      wave_sync();
      if (lane == 0) F::template plus<T>(static_cast<const T*>(L.xs), static_cast<const T*>(dd), xn);
      wave_sync();
      if (lane < kXdim) L.xs[lane] = xn[lane];
      wave_sync();
      return;
    }
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

// kPackedRows: the items are the packed rows of a TOA_MODEL_DENSE_ROW problem (n <= 15: [m4][n + 1], rows padded to a multiple of four)
template <typename F, typename = void>
struct FunctorPackedRows { static constexpr bool value = false; };
template <typename F>
struct FunctorPackedRows<F, std::enable_if_t<F::kPackedRows>> { static constexpr bool value = true; };

// ROBUST = false: a variant WITHOUT the M-estimator branch of the passes (toa_set_loss is ignored: the launchers pick it for plain L2
// solves only).  The estimators' fp64 exp / log / atan2 cost the fused kernel its occupancy for everybody: fp64 n = 6 with the branch
// 268 registers = one wave per SIMD, without 167 = three (profiles/r06_ab_log.md section 12).
template <typename T, typename F, int MANIFOLD = 0, bool ROBUST = true>
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
    const size_t stride = FunctorPackedRows<F>::value ? size_t((items + 3) & ~3) * F::kD : F::kH + size_t(items) * F::kD;
    d = data + size_t(p) * stride;
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
    const bool robust = ROBUST && loss != TOA_LOSS_L2;   // wave-uniform
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

// The DenseRow residual WITH its Jacobian row, an item = one packed row [a_i | b_i] of the n <= 15 layouts (DenseRowLayout: no block
// padding there, row stride n + 1): what JetModel (the Gram in registers, n <= 10 in fp32, n <= 5 in fp64) and RowModel (a row per lane
// staged into the MFMA Gram) run for narrow blocks of TOA_MODEL_DENSE_ROW (round 6) instead of sixteen lanes per row, whose pass costs
// ~26 vector instructions per four rows whatever their length.
template <typename T, int NN>
struct DenseRowPackedFunctor {
  static constexpr int kN = NN, kR = 1, kD = NN + 1, kH = 0;
  static constexpr bool kManual = true;
  static constexpr bool kPackedRows = true;   // the problem stride is the packed layout's (rows padded to a multiple of four)
  template <bool want_grad>
  static __device__ __forceinline__ void eval_manual(const T* x, const T*, const T* p, T* r, T (*J)[kN]) {
    T t = x[0] * p[0];
#pragma unroll
    for (int j = 1; j < NN; ++j) t = fma(x[j], p[j], t);
    T sn, cs;
    sincos_t(t, &sn, &cs);
    r[0] = (t + T(0.1) * sn) - p[NN];
    if constexpr (want_grad) {
      const T sc = T(1) + T(0.1) * cs;
#pragma unroll
      for (int j = 0; j < NN; ++j) J[0][j] = sc * p[j];
    }
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
