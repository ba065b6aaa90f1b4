"""tinyopt_amd — MI355X-native batched Levenberg-Marquardt behind tinyopt's ``Optimize(x, cost, options)``.

Host-side mirror (Python flavour) of the reference's user API for the LM hot path
(include/tinyopt/optimize.h:16-77).  PyTorch is used for device memory, streams and
``torch.distributed`` only; all arithmetic runs in the hand-written HIP kernels of
``tinyopt_amd/csrc`` through the C-ABI of ``include/tinyopt_amd.h``.  No CPU fallback exists.
"""
from ._capi import (F32, F64, MODEL_DENSE_ROW, STOP_NAMES, ToaError, ToaOptions, ToaResults, load)  # noqa: F401
from .api import (Context, BundleAdjustmentLists, JitResidual, JitModel, DenseRow, GaussianPrior, Sqrt2, SE3Reproj, CircleFit, DenseRowAD6, DenseRowAD, DenseRowNatural, BundleAdjustment, TestFn, MahaPrior, SE3Prior, Options, Output, Optimize, Optimizer, StopReason, accumulate, solve_damped, inv_cov, robust_norm, LOSS_KINDS)  # noqa: F401
from .dist import Communicator, gather_native, gather_output, shard_range  # noqa: F401

__all__ = ["Context", "BundleAdjustmentLists", "JitResidual", "JitModel", "DenseRow", "GaussianPrior", "Sqrt2", "SE3Reproj", "CircleFit", "DenseRowAD6", "DenseRowAD", "DenseRowNatural", "BundleAdjustment", "TestFn", "MahaPrior", "SE3Prior", "Options", "Output", "Optimize", "Optimizer", "StopReason", "accumulate", "solve_damped", "inv_cov", "robust_norm", "LOSS_KINDS",
           "gather_output", "gather_native", "Communicator", "shard_range", "F32", "F64"]
