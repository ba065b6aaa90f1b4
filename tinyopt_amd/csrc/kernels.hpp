// Kernels of the LM hot path + their host launchers (templates).  Included by capi.hip (C-ABI,
// dispatch) and by inst.hip, which is compiled once per (dtype, block count) so that the ~40
// instantiations of the fused kernel build in parallel (see __graft_entry__.build()).
//
// Kernel inventory:
//   lm_fused_kernel        K1+K2+K3+K4: whole LM solves, one wavefront per problem at a time,
//                          dynamic problem queue, no inter-wave or host synchronisation.
//   accumulate_kernel      K1/K2 seam: the Accumulate callback for a batch (g, H, cost out).
//   solve_damped_kernel    K3 seam: damping + LDL^T solve for a batch.
//
// Round 6: one header per kernel family —
//   models_dense.hpp     DenseRowModel (+ the cooperative tail's control blocks)
//   models_analytic.hpp  GaussianPrior / MahaPrior / TestFn / Sqrt2
//   models_se3.hpp       manifold policies, SE3 maps, Se3Prior / Se3Reproj
//   models_jet.hpp       JetModel (n <= 12), functor traits, built-in functors  -> row_model.hpp (13 <= n <= 63)
//   fused_kernels.hpp    lm_fused_kernel, accumulate / solve_damped / inv_cov seams
//   wide_kernels.hpp     row-split, stepping-state, team and persistent kernels
//   host_launch.hpp      handle, launchers (host only: left out of run-time builds)
#pragma once
#include <hip/hip_runtime.h>

#ifdef __HIPCC_RTC__   // run-time compilation (jit.hip): the headers are handed to hiprtc by NAME, embedded in the library
#include "tinyopt_amd.h"
#else
#include "../../include/tinyopt_amd.h"
#endif
#include "wide_kernels.hpp"
#ifndef __HIPCC_RTC__
#include "host_launch.hpp"
#endif
