// One translation unit per (dtype, block count): compiled with -DTOA_INST_DT={0,1} -DTOA_INST_NBM={1..4},
// or -DTOA_INST_SOLVE -DTOA_INST_DT={0,1} for the K3 seam kernels.  See __graft_entry__.build().
#include "kernels.hpp"
using namespace toa;

#if TOA_INST_DT == 0
using InstT = float;
#else
using InstT = double;
#endif
#define TOA_CAT2(a, b, c) a##b##_##c
#define TOA_CAT(a, b, c) TOA_CAT2(a, b, c)

#ifdef TOA_INST_MISC
int TOA_CAT(toa_inst_misc_fused_, TOA_INST_DT, 0)(int model, int npad, toa_handle h, const FusedParams& prm) {
  if (model == TOA_MODEL_SQRT2) return launch_fused<Sqrt2Model<InstT>>(h, prm);
  if (model == TOA_MODEL_TESTFN) return launch_fused<TestFnModel<InstT>>(h, prm);
  if (model == TOA_MODEL_SE3_REPROJ)
    return h->tune.se3_reproj_header_l2 ? launch_fused<Se3ReprojModel<InstT, false>>(h, prm) : launch_fused<Se3ReprojModel<InstT>>(h, prm);
  if (model == TOA_MODEL_SE3_PRIOR) return launch_fused<Se3PriorModel<InstT>>(h, prm);
  // (plain L2 solves: the variant without the M-estimator branch — models_jet.hpp JetModel ROBUST)
  const bool robust = prm.loss != TOA_LOSS_L2;
  if (model == TOA_MODEL_CIRCLE_FIT)
    return robust ? launch_fused<JetModel<InstT, CircleFitFunctor<InstT>>>(h, prm) : launch_fused<JetModel<InstT, CircleFitFunctor<InstT>, 0, false>>(h, prm);
  if (model == TOA_MODEL_DENSE_ROW_AD6)
    return robust ? launch_fused<JetModel<InstT, DenseRowAdFunctor<InstT, 6>>>(h, prm) : launch_fused<JetModel<InstT, DenseRowAdFunctor<InstT, 6>, 0, false>>(h, prm);
  if (model == TOA_MODEL_MAHA_PRIOR) {
    switch (npad) {
      case 16: return launch_fused<MahaPriorModel<InstT, 16>>(h, prm);
      case 32: return launch_fused<MahaPriorModel<InstT, 32>>(h, prm);
      case 48: return launch_fused<MahaPriorModel<InstT, 48>>(h, prm);
      default: return launch_fused<MahaPriorModel<InstT, 64>>(h, prm);
    }
  }
  switch (npad) {
    case 16: return launch_fused<GaussianPriorModel<InstT, 16>>(h, prm);
    case 32: return launch_fused<GaussianPriorModel<InstT, 32>>(h, prm);
    case 48: return launch_fused<GaussianPriorModel<InstT, 48>>(h, prm);
    default: return launch_fused<GaussianPriorModel<InstT, 64>>(h, prm);
  }
}
int TOA_CAT(toa_inst_misc_wide_, TOA_INST_DT, 0)(int model, toa_handle h, const FusedParams& prm, int splits) {
  if (model == TOA_MODEL_SE3_REPROJ)
    return h->tune.se3_reproj_header_l2 ? launch_wide<Se3ReprojModel<InstT, false>, 16, Se3Manifold<InstT>>(h, prm, splits)
                                        : launch_wide<Se3ReprojModel<InstT>, 16, Se3Manifold<InstT>>(h, prm, splits);
  if (prm.mode == 0) return toa_fail(TOA_E_UNSUPPORTED, "row-split execution is available for DenseRow and SE3Reproj");
  // stepping form: every model, one chunk per problem
  using E = EuclidManifold<InstT>;
  switch (model) {
    case TOA_MODEL_SQRT2: return launch_stepping<Sqrt2Model<InstT>, 16, E>(h, prm);
    case TOA_MODEL_TESTFN: return launch_stepping<TestFnModel<InstT>, 16, E>(h, prm);
    case TOA_MODEL_SE3_PRIOR: return launch_stepping<Se3PriorModel<InstT>, 16, Se3Manifold<InstT>>(h, prm);
    case TOA_MODEL_CIRCLE_FIT: return launch_stepping<JetModel<InstT, CircleFitFunctor<InstT>>, 16, E>(h, prm);
    case TOA_MODEL_DENSE_ROW_AD6: return launch_stepping<JetModel<InstT, DenseRowAdFunctor<InstT, 6>>, 16, E>(h, prm);
    default: break;
  }
  const int npad = 16 * ((prm.n + 15) / 16);
  if (model == TOA_MODEL_MAHA_PRIOR) {
    switch (npad) {
      case 16: return launch_stepping<MahaPriorModel<InstT, 16>, 16, E>(h, prm);
      case 32: return launch_stepping<MahaPriorModel<InstT, 32>, 32, E>(h, prm);
      case 48: return launch_stepping<MahaPriorModel<InstT, 48>, 48, E>(h, prm);
      default: return launch_stepping<MahaPriorModel<InstT, 64>, 64, E>(h, prm);
    }
  }
  switch (npad) {  // TOA_MODEL_GAUSSIAN_PRIOR
    case 16: return launch_stepping<GaussianPriorModel<InstT, 16>, 16, E>(h, prm);
    case 32: return launch_stepping<GaussianPriorModel<InstT, 32>, 32, E>(h, prm);
    case 48: return launch_stepping<GaussianPriorModel<InstT, 48>, 48, E>(h, prm);
    default: return launch_stepping<GaussianPriorModel<InstT, 64>, 64, E>(h, prm);
  }
}
int TOA_CAT(toa_inst_misc_accumulate_, TOA_INST_DT, 0)(int model, int npad, toa_handle h, int n, int m, int64_t P,
                                                       const void* data, const void* x, int want_grad, void* g, void* H,
                                                       double* cost, int32_t* nres) {
  (void)npad;
  if (model == TOA_MODEL_SQRT2) return launch_accumulate<Sqrt2Model<InstT>>(h, n, m, P, data, x, want_grad, g, H, cost, nres);
  if (model == TOA_MODEL_TESTFN) return launch_accumulate<TestFnModel<InstT>>(h, n, m, P, data, x, want_grad, g, H, cost, nres);
  if (model == TOA_MODEL_SE3_REPROJ)
    return launch_accumulate<Se3ReprojModel<InstT>>(h, n, m, P, data, x, want_grad, g, H, cost, nres);
  if (model == TOA_MODEL_SE3_PRIOR)
    return launch_accumulate<Se3PriorModel<InstT>>(h, n, m, P, data, x, want_grad, g, H, cost, nres);
  if (model == TOA_MODEL_CIRCLE_FIT)
    return launch_accumulate<JetModel<InstT, CircleFitFunctor<InstT>>>(h, n, m, P, data, x, want_grad, g, H, cost, nres);
  if (model == TOA_MODEL_DENSE_ROW_AD6)
    return launch_accumulate<JetModel<InstT, DenseRowAdFunctor<InstT, 6>>>(h, n, m, P, data, x, want_grad, g, H, cost, nres);
  if (model == TOA_MODEL_MAHA_PRIOR) return launch_accumulate<MahaPriorModel<InstT, 16>>(h, n, m, P, data, x, want_grad, g, H, cost, nres);
  return launch_accumulate<GaussianPriorModel<InstT, 16>>(h, n, m, P, data, x, want_grad, g, H, cost, nres);
}
#elif defined(TOA_INST_JETROW)
// TOA_MODEL_DENSE_ROW_AD: the DenseRow residual written as r(x) only, differentiated on the device for WIDE parameter
// blocks (RowModel over AdRowFunctor, row_model.hpp): n = 12 (the C3 shape) and n = 50 (the C4 shape) are instantiated — a functor's parameter count is
// a compile-time constant, as in the reference's static-size Jets.
int TOA_CAT(toa_inst_jetrow_fused_, TOA_INST_DT, 0)(int n, toa_handle h, const FusedParams& prm) {
  // (toa_set_loss is refused for this family — capi.hip check_loss_supported — so its fused kernel is the variant without the M-estimator branch)
  if (n == 12) return launch_fused<RowModel<InstT, 1, 0, AdRowFunctor<InstT, DenseRowAdFunctor<InstT, 12>>, 0, false>>(h, prm);
  if (n == 50) return launch_fused<RowModel<InstT, 3, 3, AdRowFunctor<InstT, DenseRowAdFunctor<InstT, 50>>, 0, false>>(h, prm);
  return toa_fail(TOA_E_UNSUPPORTED, "TOA_MODEL_DENSE_ROW_AD is instantiated for n = 12 and n = 50");
}
int TOA_CAT(toa_inst_jetrow_wide_, TOA_INST_DT, 0)(int n, toa_handle h, const FusedParams& prm) {
  using E = EuclidManifold<InstT>;
  if (n == 12) return launch_stepping<RowModel<InstT, 1, 0, AdRowFunctor<InstT, DenseRowAdFunctor<InstT, 12>>>, 16, E>(h, prm);
  if (n == 50) return launch_stepping<RowModel<InstT, 3, 3, AdRowFunctor<InstT, DenseRowAdFunctor<InstT, 50>>>, 64, E>(h, prm);
  return toa_fail(TOA_E_UNSUPPORTED, "TOA_MODEL_DENSE_ROW_AD is instantiated for n = 12 and n = 50");
}
int TOA_CAT(toa_inst_jetrow_accumulate_, TOA_INST_DT, 0)(toa_handle h, int n, int m, int64_t P, const void* data, const void* x,
                                                         int want_grad, void* g, void* H, double* cost, int32_t* nres) {
  if (n == 12) return launch_accumulate<RowModel<InstT, 1, 0, AdRowFunctor<InstT, DenseRowAdFunctor<InstT, 12>>>>(h, n, m, P, data, x, want_grad, g, H, cost, nres);
  if (n == 50) return launch_accumulate<RowModel<InstT, 3, 3, AdRowFunctor<InstT, DenseRowAdFunctor<InstT, 50>>>>(h, n, m, P, data, x, want_grad, g, H, cost, nres);
  return toa_fail(TOA_E_UNSUPPORTED, "TOA_MODEL_DENSE_ROW_AD is instantiated for n = 12 and n = 50");
}
#elif defined(TOA_INST_NARROW)
// TOA_MODEL_DENSE_ROW on the narrow routes (round 6; the packed rows of the n <= 15 layouts — and of the thin layouts — ARE items [a_i | b_i]):
//   JetModel over DenseRowPackedFunctor   an item per lane, the (n + 1)(n + 2) / 2 Gram in registers: fp32 n <= 10, fp64 n <= 5
//   RowModel over DenseRowPackedFunctor   a row per lane staged into the MFMA Gram: fp32 n = 11; with an M-estimator on the handle
//                                         (toa_set_loss) also the BASELINE shapes n = 12, 50 (fp64: n = 6, 12, 50) — the loss then runs
//                                         INSIDE the fused kernel instead of the launch-per-iteration form
// against sixteen lanes per row: fp32 n = 6 x 1000: 3.07 -> 1.46 ms per 2 GB of rows, fp64 n = 4: 3.30 -> 2.02 (profiles/r06_ab_log.md
// sections 7, 8, 11).  A functor's parameter count is a compile-time constant, as for TOA_MODEL_DENSE_ROW_AD.
#if TOA_INST_DT == 0
#define TOA_NARROW_CASES(JET, ROW) \
  switch (n) {                     \
    case 1: JET(1); case 2: JET(2); case 3: JET(3); case 4: JET(4); case 5: JET(5); case 6: JET(6); case 7: JET(7); case 8: JET(8); case 9: JET(9); \
    case 10: JET(10); case 11: ROW(1, 0, 11); case 12: ROW(1, 0, 12); case 50: ROW(3, 3, 50); \
    default: break;                \
  }
#else
// fp64: the estimators' exp / log / atan2 in double precision cost JetModel its occupancy (n = 6: 268 registers = one wave per SIMD), so plain L2
// solves run the variant without that branch (167 registers: three waves per SIMD) — n <= 6 —, solves with an M-estimator the full one
#define TOA_NARROW_CASES(JET, ROW) \
  switch (n) {                     \
    case 1: JET(1); case 2: JET(2); case 3: JET(3); case 4: JET(4); case 5: JET(5); case 6: if (!robust) { JET(6); } else { ROW(1, 0, 6); } \
    case 12: ROW(1, 0, 12); case 50: ROW(3, 3, 50); \
    default: break;                \
  }
#endif
// (two translation units per dtype — -DTOA_NARROW_PART=0: the JetModel fused kernels, 1: the RowModel ones, the seam and the dispatch — so that
//  neither is the long pole of the build)
int TOA_CAT(toa_inst_narrow_jet_fused_, TOA_INST_DT, 0)(int n, toa_handle h, const FusedParams& prm);
#if TOA_NARROW_PART == 0
int TOA_CAT(toa_inst_narrow_jet_fused_, TOA_INST_DT, 0)(int n, toa_handle h, const FusedParams& prm) {
  const bool robust = prm.loss != TOA_LOSS_L2;
  (void)robust;
#if TOA_INST_DT == 0
#define TOA_NJ(N) return launch_fused<JetModel<InstT, DenseRowPackedFunctor<InstT, N>>>(h, prm)
#else
#define TOA_NJ(N) \
  { if (robust) return launch_fused<JetModel<InstT, DenseRowPackedFunctor<InstT, N>>>(h, prm); \
    return launch_fused<JetModel<InstT, DenseRowPackedFunctor<InstT, N>, 0, false>>(h, prm); }
#endif
#define TOA_NONE(NB, TH, N) break
  TOA_NARROW_CASES(TOA_NJ, TOA_NONE)
#undef TOA_NJ
#undef TOA_NONE
  return toa_fail(TOA_E_ARG, "DenseRow narrow route: no JetModel instance for this n");
}
#else
int TOA_CAT(toa_inst_narrow_fused_, TOA_INST_DT, 0)(int n, toa_handle h, const FusedParams& prm) {
  const bool robust = prm.loss != TOA_LOSS_L2;
  (void)robust;
#define TOA_NJ(N) return TOA_CAT(toa_inst_narrow_jet_fused_, TOA_INST_DT, 0)(n, h, prm)
#define TOA_NF(NB, TH, N) return launch_fused<RowModel<InstT, NB, TH, DenseRowPackedFunctor<InstT, N>>>(h, prm)
  TOA_NARROW_CASES(TOA_NJ, TOA_NF)
#undef TOA_NJ
#undef TOA_NF
  return toa_fail(TOA_E_ARG, "DenseRow narrow route: no instance for this n");
}
int TOA_CAT(toa_inst_narrow_accumulate_, TOA_INST_DT, 0)(toa_handle h, int n, int m, int64_t P, const void* data, const void* x, int want_grad, void* g,
                                                         void* H, double* cost, int32_t* nres) {
  const bool robust = false;   // (the seam takes this route for plain L2 only: capi.hip)
  (void)robust;
#define TOA_NJ(N) return launch_accumulate<JetModel<InstT, DenseRowPackedFunctor<InstT, N>>>(h, n, m, P, data, x, want_grad, g, H, cost, nres)
#define TOA_NA(NB, TH, N) return launch_accumulate<RowModel<InstT, NB, TH, DenseRowPackedFunctor<InstT, N>>>(h, n, m, P, data, x, want_grad, g, H, cost, nres)
  TOA_NARROW_CASES(TOA_NJ, TOA_NA)
#undef TOA_NJ
#undef TOA_NA
  return toa_fail(TOA_E_ARG, "DenseRow narrow route: no instance for this n");
}
#endif   // TOA_NARROW_PART
#elif defined(TOA_INST_SOLVE)
int TOA_CAT(toa_inst_solve_, TOA_INST_DT, 0)(int npad, toa_handle h, int n, int64_t P, const void* H, const void* g,
                                             double scale, void* dx, int32_t* ok) {
  switch (npad) {
    case 16: return launch_solve<InstT, 16>(h, n, P, H, g, scale, dx, ok);
    case 32: return launch_solve<InstT, 32>(h, n, P, H, g, scale, dx, ok);
    case 48: return launch_solve<InstT, 48>(h, n, P, H, g, scale, dx, ok);
    default: return launch_solve<InstT, 64>(h, n, P, H, g, scale, dx, ok);
  }
}
int TOA_CAT(toa_inst_inv_cov_, TOA_INST_DT, 0)(int npad, toa_handle h, int n, int64_t P, const void* H, void* C, int32_t* ok) {
  switch (npad) {
    case 16: return launch_inv_cov<InstT, 16>(h, n, P, H, C, ok);
    case 32: return launch_inv_cov<InstT, 32>(h, n, P, H, C, ok);
    case 48: return launch_inv_cov<InstT, 48>(h, n, P, H, C, ok);
    default: return launch_inv_cov<InstT, 64>(h, n, P, H, C, ok);
  }
}
#else
int TOA_CAT(toa_inst_fused_, TOA_INST_DT, TOA_INST_NBM)(int thin, toa_handle h, const FusedParams& prm) {
  // the fused kernel runs the COOP variant of the model (ticketed row chunks: kernels.hpp CoopCtl) wherever one exists —
  // everywhere but the fp64 n <= 15 shapes, whose pass works in 64-row super-batches
  // ... and the three layouts where the variant costs a resident wave per SIMD (tools/kernel_regs.py against the plain
  // variant: f32 (1,1) 80 -> 84 registers, f32 (1,2) 92 -> 100, f64 (2,4) 256 -> 260)
  constexpr bool kF32 = sizeof(InstT) == 4;
  // fp64 n <= 15 (the 64-row super-batch layouts, C2 / C3): COOP selects the row-per-lane pass through LDS (pass16s), not the
  // cooperative form — that one exists (pass16's step0 / step1) but loses: same box, 10 000 problems, m = 500: off 67.4 M it/s,
  // K = 2: 64.7, K = 4: 63.3 (profiles/r03_ab_log.md).
  constexpr bool kCoop0 = true;
  constexpr bool kCoop1 = !(kF32 && TOA_INST_NBM == 1);
  constexpr bool kCoop2 = !(kF32 && TOA_INST_NBM == 1);
  constexpr bool kCoop4 = !(!kF32 && TOA_INST_NBM == 2);
  switch (thin) {
    case 0: return launch_fused<DenseRowModel<InstT, TOA_INST_NBM, 0, false, kCoop0>>(h, prm);
#if TOA_INST_NBM <= 3
    case 1: return launch_fused<DenseRowModel<InstT, TOA_INST_NBM, 1, false, kCoop1>>(h, prm);
    case 2: return launch_fused<DenseRowModel<InstT, TOA_INST_NBM, 2, false, kCoop2>>(h, prm);
    case 3: return launch_fused<DenseRowModel<InstT, TOA_INST_NBM, 3, false, true>>(h, prm);
    case 4: return launch_fused<DenseRowModel<InstT, TOA_INST_NBM, 4, false, kCoop4>>(h, prm);
#endif
    default: return toa_fail(TOA_E_ARG, "bad thin-tail width");
  }
}
int TOA_CAT(toa_inst_wide_, TOA_INST_DT, TOA_INST_NBM)(int thin, toa_handle h, const FusedParams& prm, int splits) {
  constexpr int kNp0 = 16 * TOA_INST_NBM;  // NPAD of the solve: see DenseRowModel::kNpad
  switch (thin) {
    case 0: return launch_wide<DenseRowModel<InstT, TOA_INST_NBM, 0>, kNp0, EuclidManifold<InstT>>(h, prm, splits);
#if TOA_INST_NBM <= 3
    case 1: return launch_wide<DenseRowModel<InstT, TOA_INST_NBM, 1>, kNp0 + 16, EuclidManifold<InstT>>(h, prm, splits);
    case 2: return launch_wide<DenseRowModel<InstT, TOA_INST_NBM, 2>, kNp0 + 16, EuclidManifold<InstT>>(h, prm, splits);
    case 3: return launch_wide<DenseRowModel<InstT, TOA_INST_NBM, 3>, kNp0 + 16, EuclidManifold<InstT>>(h, prm, splits);
    case 4: return launch_wide<DenseRowModel<InstT, TOA_INST_NBM, 4>, kNp0 + 16, EuclidManifold<InstT>>(h, prm, splits);
#endif
    default: return toa_fail(TOA_E_ARG, "bad thin-tail width");
  }
}
int TOA_CAT(toa_inst_accumulate_, TOA_INST_DT, TOA_INST_NBM)(int thin, toa_handle h, int n, int m, int64_t P,
                                                             const void* data, const void* x, int want_grad, void* g,
                                                             void* H, double* cost, int32_t* nres) {
  switch (thin) {
    case 0: return launch_accumulate<DenseRowModel<InstT, TOA_INST_NBM, 0>>(h, n, m, P, data, x, want_grad, g, H, cost, nres);
#if TOA_INST_NBM <= 3
    case 1: return launch_accumulate<DenseRowModel<InstT, TOA_INST_NBM, 1>>(h, n, m, P, data, x, want_grad, g, H, cost, nres);
    case 2: return launch_accumulate<DenseRowModel<InstT, TOA_INST_NBM, 2>>(h, n, m, P, data, x, want_grad, g, H, cost, nres);
    case 3: return launch_accumulate<DenseRowModel<InstT, TOA_INST_NBM, 3>>(h, n, m, P, data, x, want_grad, g, H, cost, nres);
    case 4: return launch_accumulate<DenseRowModel<InstT, TOA_INST_NBM, 4>>(h, n, m, P, data, x, want_grad, g, H, cost, nres);
#endif
    default: return toa_fail(TOA_E_ARG, "bad thin-tail width");
  }
}
#endif
