"""Host-side mirror of tinyopt's Optimize / Options / Output for batched device models.

Names and meanings follow the reference:
  Options   include/tinyopt/optimizers/options.h:18-156 (nested groups hessian / cost / lm kept)
  Output    include/tinyopt/output.h:26-145 (one entry per problem of the batch)
  Optimize  include/tinyopt/optimize.h:16-77  ``Optimize(x, cost, options)``; x is updated in place
"""
from __future__ import annotations

import ctypes as C
import enum

import numpy as np
from dataclasses import dataclass, field
from typing import Optional

import torch

from . import _capi
from ._capi import (F32, F64, MODEL_DENSE_ROW_AD, MODEL_DENSE_ROW_NATURAL, MODEL_TESTFN, MODEL_MAHA_PRIOR, MODEL_SE3_PRIOR, MODEL_CIRCLE_FIT, MODEL_DENSE_ROW, MODEL_DENSE_ROW_AD6, MODEL_GAUSSIAN_PRIOR, MODEL_SE3_REPROJ,
                    MODEL_SQRT2, ToaOptions, ToaResults, check)


class StopReason(enum.IntEnum):  # include/tinyopt/stop_reasons.h:14-43
    kOutOfMemory = -4
    kSolverFailed = -3
    kSystemHasNaNOrInf = -2
    kSkipped = -1
    kNone = 0
    kMinError = 1
    kMinRelError = 2
    kMinDeltaNorm = 3
    kMinGradNorm = 4
    kMaxIters = 5
    kMaxNoDecr = 6
    kMaxConsecNoDecr = 7
    kTimedOut = 8
    kUserStopped = 9


@dataclass
class _Hessian:  # options.h:58-67
    use_ldlt: bool = True
    H_is_full: bool = True
    check_min_H_diag: float = 0.0
    save_last: bool = True


@dataclass
class _CostScaling:  # options.h:75-80
    use_squared_norm: bool = True
    downscale_by_2: bool = False
    normalize: bool = False


@dataclass
class _LM:  # options.h:127-141
    damping_init: float = 1e-4
    damping_range: tuple = (1e-9, 1e9)
    good_factor: float = 1.0 / 3.0
    bad_factor: float = 2.0


@dataclass
class _Log:  # options.h:113-125 — the per-iteration log line of Optimizer_::Step (optimizer.h:463-516)
    enable: bool = False           # (the reference logs by default; a batched solve does not: one line per problem and iteration)
    e: str = "\u03b5\u00b2"          # symbol of the error in the line
    print_emoji: bool = True
    print_x: bool = False
    print_dx: bool = False
    print_inliers: bool = False
    print_t: bool = True
    problems: Optional[object] = None   # which problems of the batch are logged: None = problem 0, an iterable of indices, or "all"
    sink: Optional[object] = None       # callable(str); None = print (the reference's TINYOPT_LOG writes to std::cout, log.h:23)


@dataclass
class Options:
    """tinyopt::Options (options.h:18-156).  ``max_duration_ms``, ``stop_callback`` and ``stop_callback2`` are
    host-side controls: when any of them is set, ``Optimize`` runs the loop through the stepping form and evaluates
    them between iterations (optimizer.h:302-305, 529-534) — per problem: ``stop_callback(err, dx_norm2, grad_norm2)``
    / ``stop_callback2(err, dx, g)`` with numpy vectors.  ``log.enable`` (off by default here) prints the reference's
    per-iteration line for the chosen problems through the same stepping form (``log.print_max_stdev`` / ``print_J_jet`` /
    ``print_failure`` are not mirrored)."""
    LevenbergMarquardt = 0
    GaussNewton = 1
    solver_type: int = 0
    check_final_cost: bool = False
    use_step_quality_approx: bool = False
    grad_clipping: float = 0.0
    hessian: _Hessian = field(default_factory=_Hessian)
    cost: _CostScaling = field(default_factory=_CostScaling)
    max_iters: int = 50
    min_error: float = 1e-12
    min_rerr_dec: float = 1e-10
    min_step_norm2: float = 1e-14
    min_grad_norm2: float = 1e-18
    max_total_failures: int = 0
    max_consec_failures: int = 5
    max_duration_ms: float = 0.0               # options.h:96; 0 = no limit
    stop_callback: Optional[object] = None     # options.h:97-101  bool(err, |dx|^2, |g|^2)
    stop_callback2: Optional[object] = None    # options.h:102-106 bool(err, dx, g)
    lm: _LM = field(default_factory=_LM)
    log: _Log = field(default_factory=_Log)

    def has_host_controls(self) -> bool:
        return self.max_duration_ms > 0 or self.stop_callback is not None or self.stop_callback2 is not None or self.log.enable

    @staticmethod
    def benchmark() -> "Options":
        """benchmarks/options.h:10-27 CreateOptions()."""
        o = Options()
        o.max_iters = 10
        o.min_error = 0.0
        o.min_rerr_dec = 1e-12
        o.min_step_norm2 = 1e-16
        o.max_consec_failures = 3
        o.hessian.save_last = False
        return o

    def to_pod(self) -> ToaOptions:
        p = ToaOptions()
        p.solver_type = int(self.solver_type)
        p.max_iters = int(self.max_iters)
        p.min_error = self.min_error
        p.min_rerr_dec = self.min_rerr_dec
        p.min_step_norm2 = self.min_step_norm2
        p.min_grad_norm2 = self.min_grad_norm2
        p.max_total_failures = int(self.max_total_failures)
        p.max_consec_failures = int(self.max_consec_failures)
        p.damping_init = self.lm.damping_init
        p.damping_min, p.damping_max = self.lm.damping_range
        p.good_factor = self.lm.good_factor
        p.bad_factor = self.lm.bad_factor
        p.grad_clipping = self.grad_clipping
        p.check_min_H_diag = self.hessian.check_min_H_diag
        p.check_final_cost = int(self.check_final_cost)
        p.use_step_quality_approx = int(self.use_step_quality_approx)
        p.use_ldlt = int(self.hessian.use_ldlt)
        p.H_is_full = int(self.hessian.H_is_full)
        p.save_last = int(self.hessian.save_last)
        p.use_squared_norm = int(self.cost.use_squared_norm)
        p.downscale_by_2 = int(self.cost.downscale_by_2)
        p.normalize = int(self.cost.normalize)
        return p


def _dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return F32
    if dt == torch.float64:
        return F64
    raise ValueError("dtype must be torch.float32 or torch.float64")  # reference: Scalar is float/double


class Context:
    """Owns a C-ABI handle bound to one GPU and torch's current stream on it."""

    def __init__(self, device: Optional[int] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("tinyopt_amd needs a ROCm GPU (MI355X / gfx950); there is no CPU path")
        self.lib = _capi.load()
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.stream = torch.cuda.current_stream(self.device)
        h = C.c_void_p()
        check(self.lib.toa_create(C.byref(h), self.device, C.c_void_p(self.stream.cuda_stream)))
        self.h = h

    def info(self):
        cus, khz = C.c_int(), C.c_int()
        name = C.create_string_buffer(128)
        check(self.lib.toa_device_info(self.h, C.byref(cus), C.byref(khz), name, 128))
        return {"num_cus": cus.value, "clock_khz": khz.value, "name": name.value.decode()}

    def hbm_read_GBps(self, tensor, reps: int = 5) -> float:
        """Measured streaming-read rate over `tensor`'s bytes (the STREAM-like ceiling of SURVEY §8d)."""
        out = C.c_double()
        check(self.lib.toa_hbm_read_probe(self.h, C.c_void_p(tensor.data_ptr()), tensor.numel() * tensor.element_size(),
                                          int(reps), C.byref(out)))
        return out.value

    def llc_read_GBps(self, tensor, max_bytes: int = 192 << 20) -> float:
        """Measured re-read rate of a working set that fits the Infinity Cache (the first `max_bytes` of `tensor`)."""
        out = C.c_double()
        nbytes = min(tensor.numel() * tensor.element_size(), int(max_bytes))
        check(self.lib.toa_llc_read_probe(self.h, C.c_void_p(tensor.data_ptr()), nbytes, C.byref(out)))
        return out.value

    def set_tuning(self, **kw) -> None:
        """The A/B arms of the library (include/tinyopt_amd.h toa_tuning) as typed per-handle state: ``ctx.set_tuning(memo_off=1)``;
        no arguments = the library's own choices.  The product reads no environment variable."""
        t = _capi.ToaTuning()
        for k, v in kw.items():
            if k == "reserved" or not hasattr(t, k):
                raise ValueError(f"unknown tuning field {k!r}")
            setattr(t, k, int(v))
        check(self.lib.toa_set_tuning(self.h, C.byref(t) if kw else None))
        self._se3_l2 = int(kw.get("se3_reproj_header_l2", 0))   # (what _apply_loss last left in that field: it only calls when it changes)

    def get_tuning(self) -> dict:
        t = _capi.ToaTuning()
        check(self.lib.toa_get_tuning(self.h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in t._fields_ if k != "reserved"}

    def tuning(self, **kw):
        """``with ctx.tuning(coop_off=1): ...`` — the fields set for the block, the previous state restored after it."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            old = self.get_tuning()
            self.set_tuning(**{**old, **kw})
            try:
                yield self
            finally:
                self.set_tuning(**old) if any(old.values()) else self.set_tuning()
        return cm()

    def debug_timeline(self, path: Optional[str]) -> None:
        check(self.lib.toa_debug_timeline(self.h, path.encode() if path else None))

    def close(self):
        if getattr(self, "h", None):
            self.lib.toa_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device: Optional[int] = None) -> Context:
    dev = torch.cuda.current_device() if device is None else int(device)
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    if key not in _default_ctx:
        _default_ctx[key] = Context(dev)
    return _default_ctx[key]


def dense_row_layout(dtype: torch.dtype, n: int, m: int):
    lib = _capi.load()
    nb, thin, rs, m4 = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    nbytes = C.c_size_t()
    check(lib.toa_dense_row_layout(_dtype_code(dtype), n, m, C.byref(nb), C.byref(thin), C.byref(rs), C.byref(m4),
                                   C.byref(nbytes)))
    rsm = rs.value - thin.value
    nmr = 16 * nb.value if thin.value else n
    return {"nb": nb.value, "thin": thin.value, "row_stride": rs.value, "rows_padded": m4.value,
            "bytes_per_problem": nbytes.value,
            # physical positions inside a packed row (mirrors DenseRowLayout::pos_col / pos_b)
            "pos_cols": [j if j < nmr else rsm + (j - nmr) for j in range(n)],
            "pos_b": (rs.value - 1) if thin.value else (rsm - 1)}


class _LossMixin:
    """Models whose residual items can be wrapped in one of the reference's M-estimators (losses/robust_norms.h:32-316):
    ``model.with_loss("huber", th)`` is the device counterpart of calling ``losses::Huber(n2, th*th, true)`` on each
    residual's squared norm inside the cost functor (docs/API.md:396-411).  cost += l, the item's J^T J and J^T r are
    scaled by s = dl/dn2, and Output.final_inlier_ratio reports the residuals with n2 <= th^2 (cost.h:84-95)."""
    loss: Optional[str] = None
    th: float = 0.0

    def with_loss(self, loss: Optional[str], th: float = 0.0):
        import copy
        if loss is not None and loss not in LOSS_KINDS:
            raise ValueError(f"unknown loss {loss!r}; one of {sorted(LOSS_KINDS)}")
        m = copy.copy(self)          # shares the device data
        m.loss, m.th = (loss if loss not in (None, "l2") else None), float(th)
        return m


def _apply_loss(ctx: "Context", cost) -> None:
    """The handle carries the cost functor's M-estimator (toa_set_loss): set it from the model before every launch."""
    # SE3Reproj carries its loss in the data header: a model without one runs the kernels without the M-estimator branch (toa_tuning::
    # se3_reproj_header_l2; fp64: 308 -> 216 registers) — the field follows the model, a host call only when it changes
    want = 1 if getattr(cost, "header_l2", False) else 0
    if getattr(ctx, "_se3_l2", 0) != want:
        ctx.set_tuning(**{**ctx.get_tuning(), "se3_reproj_header_l2": want})
    kind = getattr(cost, "loss", None)
    if kind is None:
        check(ctx.lib.toa_set_loss(ctx.h, 0, 0.0))
    else:
        check(ctx.lib.toa_set_loss(ctx.h, LOSS_KINDS[kind], float(cost.th) * float(cost.th)))


class DenseRow(_LossMixin):
    """Device residual model  r_i(x) = a_i.x + 0.1 sin(a_i.x) - b_i  for a batch of P problems.

    Plays the role of the user's cost functor in ``Optimize(x, cost)`` (a Jet-invocable residual
    functor in the reference, e.g. benchmarks/dense.cpp:56,71-74) — here a tag object that selects
    the pre-instantiated device functor and owns the problem data in HBM (packed layout).
    """
    model_id = MODEL_DENSE_ROW

    def __init__(self, packed: torch.Tensor, n: int, m: int, P: int):
        self.packed, self.n, self.m, self.P = packed, int(n), int(m), int(P)
        self.dtype = packed.dtype

    @property
    def algorithmic_bytes_per_pass(self) -> int:
        """bytes one Accumulate/Evaluate pass must read per problem = m(n+1)sizeof(T)."""
        return self.m * (self.n + 1) * self.packed.element_size()

    @staticmethod
    def from_arrays(A: torch.Tensor, b: torch.Tensor, ctx: Optional[Context] = None) -> "DenseRow":
        """A: [P, m, n], b: [P, m] on the GPU (natural layout) -> packed layout."""
        ctx = ctx or default_context(A.device.index)
        P, m, n = A.shape
        assert b.shape == (P, m) and A.dtype == b.dtype and A.is_cuda and b.is_cuda
        A, b = A.contiguous(), b.contiguous()
        lay = dense_row_layout(A.dtype, n, m)
        packed = torch.empty(P * lay["rows_padded"] * lay["row_stride"], dtype=A.dtype, device=A.device)
        check(ctx.lib.toa_dense_row_pack(ctx.h, _dtype_code(A.dtype), n, m, P, A.data_ptr(), b.data_ptr(), packed.data_ptr()))
        return DenseRow(packed, n, m, P)

    @staticmethod
    def synthetic(P: int, n: int, m: int, dtype: torch.dtype, seed: int = 0x71940917, problem0: int = 0,
                  ctx: Optional[Context] = None):
        """Generate the SURVEY §8(d) batch directly in HBM.  Returns (model, x0, xstar)."""
        ctx = ctx or default_context()
        dev = torch.device("cuda", ctx.device)
        lay = dense_row_layout(dtype, n, m)
        packed = torch.empty(P * lay["rows_padded"] * lay["row_stride"], dtype=dtype, device=dev)
        x0 = torch.empty(P, n, dtype=dtype, device=dev)
        xstar = torch.empty(P, n, dtype=dtype, device=dev)
        check(ctx.lib.toa_dense_row_synth(ctx.h, _dtype_code(dtype), n, m, P, seed, problem0, packed.data_ptr(),
                                          x0.data_ptr(), xstar.data_ptr()))
        return DenseRow(packed, n, m, P), x0, xstar


class GaussianPrior:
    """Device residual model  r = (x - y) / sigma  (m = n) — the residual of the reference's published dense
    benchmark with the semantics of its manual Accumulate callback (benchmarks/dense.cpp:57-66):
    grad = J*res, H.diagonal() = sigma^-2, cost = res.squaredNorm() returned as a scalar (1 residual)."""
    model_id = MODEL_GAUSSIAN_PRIOR

    def __init__(self, y: torch.Tensor, sigma: torch.Tensor):
        assert y.shape == sigma.shape and y.dim() == 2 and y.is_cuda and y.dtype == sigma.dtype
        self.P, self.n = y.shape
        self.m = self.n
        self.dtype = y.dtype
        self.packed = torch.stack([y, sigma], dim=1).contiguous()  # [P][2][n]

    @property
    def algorithmic_bytes_per_pass(self) -> int:
        return 2 * self.n * self.packed.element_size()


class Sqrt2:
    """Device residual model  r = x*x - 2  (n = m = 1), tests/sqrt2.cpp:30-70."""
    model_id = MODEL_SQRT2

    def __init__(self, P: int, dtype: torch.dtype, device=None):
        self.P, self.n, self.m, self.dtype = int(P), 1, 1, dtype
        self.packed = torch.zeros(1, dtype=dtype, device=device or "cuda")  # no problem data

    @property
    def algorithmic_bytes_per_pass(self) -> int:
        return 0


class SE3Reproj:
    """Device residual model: pinhole reprojection of 3-D points under an SE3 pose.
    x is [P, 12] (rotation matrix row-major + translation); the tangent has n = 6 (upsilon, omega) and the update
    is the right-multiplicative  pose <- pose * exp(delta)  of include/tinyopt/3rdparty/traits/sophus.h:24-26.
    data: [P, 8 + 5*npts] = [f, cx, cy, loss, th2, 0,0,0 | x, y, z, u, v per point].

    loss / th: optional M-estimator on each point's squared reprojection error (the reference's
    losses/robust_norms.h:32-316): ``loss`` in LOSS_KINDS ("huber", "cauchy", ...), ``th`` the threshold in pixels
    (th2 = th*th is what the reference's functions take).  The inlier ratio lands in Output.final_inlier_ratio."""
    model_id = MODEL_SE3_REPROJ
    xdim = 12

    def __init__(self, data: torch.Tensor, npts: int, loss: Optional[str] = None, th: float = 0.0):
        assert data.dim() == 2 and data.shape[1] == 8 + 5 * npts and data.is_cuda
        self.P, self.n, self.m, self.dtype = data.shape[0], 6, 2 * int(npts), data.dtype
        self.packed = data.contiguous()
        if loss is not None:
            if self.packed.data_ptr() == data.data_ptr():
                self.packed = self.packed.clone()
            self.packed[:, 3] = float(LOSS_KINDS[loss])
            self.packed[:, 4] = float(th) * float(th)
        # (one read-back at construction: a header the caller filled in by hand may name a loss too)
        self.header_l2 = loss is None and not bool((self.packed[:, 3] != 0).any())

    @property
    def algorithmic_bytes_per_pass(self) -> int:
        return (self.m // 2) * 5 * self.packed.element_size()


class CircleFit(_LossMixin):
    """tests/circle.cpp:32-68 on the device, differentiated by forward-mode dual numbers (csrc/jet.hpp):
    x = (cx, cy, radius), one residual ||p - c||^2 - radius^2 per observed point.  obs: [P, m, 2]."""
    model_id = MODEL_CIRCLE_FIT

    def __init__(self, obs: torch.Tensor):
        assert obs.dim() == 3 and obs.shape[2] == 2 and obs.is_cuda
        self.P, self.m, self.n, self.dtype = obs.shape[0], obs.shape[1], 3, obs.dtype
        self.packed = obs.contiguous()

    @property
    def algorithmic_bytes_per_pass(self) -> int:
        return self.m * 2 * self.packed.element_size()


class BundleAdjustmentLists(_LossMixin):
    """Bundle adjustment with VISIBILITY LISTS (`toa_ba_lists_run`): tens to hundreds of SE3 cameras, each point observed by a
    few of them — what the reference would hand to Eigen's SimplicialLDLT (math.h:266-277; README.md:30,165-167).  Same x as
    ``BundleAdjustment`` ([P, 12 C + 3 N]); the observations are a list sorted by (point, camera):
    intr [P, 4] = (f, cx, cy, 0), obs_cam / obs_pt [P, M] int32, obs_uv [P, M, 2].  ``Options.max_duration_ms`` is honoured.
    ``.with_loss("huber", th)`` puts every observation's squared reprojection error through an M-estimator (round 4)."""
    model_id = None

    def __init__(self, intr: torch.Tensor, obs_cam: torch.Tensor, obs_pt: torch.Tensor, obs_uv: torch.Tensor, ncam: int, npts: int):
        assert intr.dim() == 2 and intr.shape[1] == 4 and intr.is_cuda
        P, M = obs_cam.shape
        assert obs_pt.shape == (P, M) and obs_uv.shape == (P, M, 2) and intr.shape[0] == P
        assert obs_cam.dtype == torch.int32 and obs_pt.dtype == torch.int32 and obs_uv.dtype == intr.dtype
        self.P, self.ncam, self.npts, self.nobs, self.dtype = P, int(ncam), int(npts), M, intr.dtype
        self.n, self.m, self.xdim = 6 * self.ncam + 3 * self.npts, 2 * M, 12 * self.ncam + 3 * self.npts
        self.intr, self.obs_cam, self.obs_pt, self.obs_uv = intr.contiguous(), obs_cam.contiguous(), obs_pt.contiguous(), obs_uv.contiguous()

    @staticmethod
    def from_dense(data: torch.Tensor, ncam: int, npts: int) -> "BundleAdjustmentLists":
        """From ``BundleAdjustment``'s dense layout [P, 8 + 3 C N] = [f cx cy 0.. | uv (C, N, 2) | vis (C, N)]; every scene must
        have the same number of visible observations."""
        P = data.shape[0]
        uv = data[:, 8:8 + 2 * ncam * npts].reshape(P, ncam, npts, 2)
        vis = data[:, 8 + 2 * ncam * npts:].reshape(P, ncam, npts) != 0
        cams, pts, uvs = [], [], []
        for p in range(P):
            v = vis[p].t().contiguous()                    # [N, C]: point-major => sorted by (point, camera)
            idx = v.nonzero(as_tuple=False)
            pts.append(idx[:, 0].to(torch.int32))
            cams.append(idx[:, 1].to(torch.int32))
            uvs.append(uv[p].permute(1, 0, 2)[v])
        if len({int(t.shape[0]) for t in pts}) != 1:
            raise ValueError("every scene must have the same number of observations")
        intr = torch.zeros(P, 4, dtype=data.dtype, device=data.device)
        intr[:, :3] = data[:, :3]
        return BundleAdjustmentLists(intr, torch.stack(cams), torch.stack(pts), torch.stack(uvs), ncam, npts)

    @property
    def algorithmic_bytes_per_pass(self) -> int:
        return self.nobs * (2 * self.obs_uv.element_size() + 8) + self.xdim * self.obs_uv.element_size()


class JitResidual:
    """A residual the library has never seen, compiled at run time: the device-side form of tinyopt's "pass any callable"
    (``Optimize(x, [](const auto& x) { return r(x); })``, optimize.h:16-33, optimizer.h:145-160, docs/API.md:21-35).

    ``body`` is the BODY of the residual as C++ source text, written like the reference's lambda and generic in its scalar
    type ``S`` (instantiated on Jet<T, n> for Accumulate — forward-mode AD, optimize_autodiff.h:91-166 — and on plain T for
    the cost-only form): ``x[j]`` parameter j (an S), ``p[k]`` the item's data scalars, ``h[k]`` the problem's header
    scalars, ``r[q]`` the item's residuals; every ceres::Jet function (sin, exp, pow, atan2, ...) is in scope.  Example, the
    circle fit of tests/circle.cpp:32-68::

        fit = ta.JitResidual("const S dx = p[0] - x[0]; const S dy = p[1] - x[1]; r[0] = dx*dx + dy*dy - x[2]*x[2];",
                             n=3, item_scalars=2, dtype=torch.float64)
        out = ta.Optimize(x, fit.bind(points), options)        # points: [P, items, 2] on the GPU

    hiprtc builds lm_fused_kernel / accumulate_kernel for JetModel<T, that functor> (2-3 s the first time; the code object is
    cached on disk — ``JitResidual.set_cache_dir`` — so a second construction of the same residual, in this or another process,
    takes milliseconds: ``from_cache``) and the code object is loaded into the process: no rebuild of libtinyopt_amd.so, and no
    source tree next to it (the kernel headers are embedded in the library).

    Round 4:
      * ``n`` up to 63 — beyond 12 parameters the functor is evaluated on chunked Jets in matrix-core operand order (the path
        of ``DenseRowAD``): one residual per item, Euclidean, no loss;
      * ``manifold="se3"`` — x is ONE pose [P, 12] (rotation matrix row-major + translation), n = 6; the body reads the pose
        through ``x[0..11]`` (Jets over the right perturbation, optimize_autodiff.h:48-77 + sophus.h:13-27) and may call
        ``se3_log<S, T>(R, t, xi)``; the update is pose <- pose * exp(delta).  tests/sophus.cpp:26-44 as a string::

            prior = ta.JitResidual(SE3_PRIOR_BODY, n=6, item_scalars=0, residuals_per_item=6, header_scalars=12, manifold="se3")

      * ``kind="accumulate"`` — a manual Accumulate callback (docs/API.md:37-57): the body fills ``r[q]`` and, inside
        ``if (want_grad) { ... }``, its own Jacobian rows ``J[q][a]`` (plain T, no AD)."""

    MANIFOLDS = {"euclid": 0, "se3": 1, "user": 2}
    KINDS = {"residual": 0, "accumulate": 1}

    def __init__(self, body: str, n: int, item_scalars: int, residuals_per_item: int = 1, header_scalars: int = 0,
                 dtype: torch.dtype = torch.float64, ctx: Optional["Context"] = None, manifold: str = "euclid", kind: str = "residual",
                 plus_body: Optional[str] = None, x_scalars: int = 0):
        """Round 5 — ``manifold="user"``: the caller's own parameter container (the reference's traits::params_trait<T>, traits.h:103-359).
        x is stored as ``x_scalars`` scalars per problem ([P, x_scalars]), ``n`` is the dimension of its tangent and ``plus_body``
        is the body of ``template <class S> void plus(const T* x, const S* d, S* xp)``: xp = x (+) d, written over the scalar type
        like the residual — the update and the roll-back run it on plain T, the differentiation on Jets seeded on d at d = 0.
        Round 6: at every n (beyond 12 parameters: up to 64 stored scalars; ``kind="accumulate"`` bodies fill J over the tangent)."""
        from ._capi import ToaJitSpec
        self.ctx = ctx or default_context()
        self.n, self.kR, self.kD, self.kH, self.dtype = int(n), int(residuals_per_item), int(item_scalars), int(header_scalars), dtype
        self.manifold, self.kind = manifold, kind
        self.xdim = 12 if manifold == "se3" else (int(x_scalars) if manifold == "user" else self.n)
        self._h = C.c_void_p()
        log = C.create_string_buffer(1 << 16)
        spec = ToaJitSpec()
        spec.dtype, spec.num_params, spec.residuals_per_item = _dtype_code(dtype), self.n, self.kR
        spec.scalars_per_item, spec.header_scalars = self.kD, self.kH
        spec.manifold, spec.kind = self.MANIFOLDS[manifold], self.KINDS[kind]
        if manifold == "user":
            if not plus_body or int(x_scalars) < 1:
                raise ValueError('manifold="user" needs plus_body (the body of x (+) d) and x_scalars')
            spec.x_scalars, spec.plus_body = int(x_scalars), plus_body.encode()
        rc = self.ctx.lib.toa_model_compile_ex(self.ctx.h, C.byref(spec), body.encode(), C.byref(self._h), log, len(log))
        self.compile_log = log.value.decode(errors="replace")
        check(rc)
        fc = C.c_int(0)
        check(self.ctx.lib.toa_jit_model_info(self._h, C.byref(fc), None))
        self.from_cache = bool(fc.value)

    def stats(self) -> dict:
        """What the run-time build of the fused kernel came out as: resident workgroups per compute unit, LDS per workgroup,
        vector registers per lane, scratch bytes per lane (0 = nothing spilled)."""
        v = [C.c_int(0) for _ in range(4)]
        check(self.ctx.lib.toa_jit_model_stats(self._h, *[C.byref(t) for t in v]))
        return dict(wg_per_cu=v[0].value, lds_bytes_per_wg=v[1].value, num_regs=v[2].value, scratch_bytes=v[3].value)

    @staticmethod
    def set_cache_dir(path: Optional[str], ctx: Optional["Context"] = None) -> None:
        """Where compiled code objects are kept (default: $XDG_CACHE_HOME/tinyopt_amd or ~/.cache/tinyopt_amd); "" = no cache."""
        check((ctx or default_context()).lib.toa_jit_set_cache_dir(None if path is None else path.encode()))   # None: back to the default

    def bind(self, data: Optional[torch.Tensor], header: Optional[torch.Tensor] = None, items: int = 1) -> "JitModel":
        """data: [P, items, item_scalars] (and header: [P, header_scalars]) on the GPU -> the ``cost`` of Optimize(x, cost).
        data = None for a functor with item_scalars = 0 (e.g. a pose prior: everything is in the header)."""
        return JitModel(self, data, header, items)

    def close(self) -> None:
        if self._h:
            self.ctx.lib.toa_model_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class JitModel(_LossMixin):
    """A JitResidual bound to its problem data ([P][header | items x item_scalars], the layout of the built-in Jet families)."""
    model_id = None

    def __init__(self, res: JitResidual, data: Optional[torch.Tensor], header: Optional[torch.Tensor] = None, items: int = 1):
        if data is None:   # a functor whose items carry no data of their own (item_scalars = 0): `items` evaluations of the header
            assert res.kD == 0 and header is not None
            data = torch.zeros(header.shape[0], int(items), 0, dtype=res.dtype, device=header.device)
        assert data.dim() == 3 and data.shape[2] == res.kD and data.is_cuda and data.dtype == res.dtype
        self.res = res
        self.P, self.items, self.n, self.dtype = data.shape[0], data.shape[1], res.n, res.dtype
        self.xdim = res.xdim
        self.m = self.items * res.kR
        flat = data.reshape(self.P, self.items * res.kD)   # (explicit: an empty batch has no size to infer)
        if res.kH:
            assert header is not None and header.shape == (self.P, res.kH) and header.dtype == res.dtype
            flat = torch.cat([header, flat], dim=1)
        self.packed = flat.contiguous()

    @property
    def algorithmic_bytes_per_pass(self) -> int:
        return self.packed.shape[1] * self.packed.element_size()


class SE3Prior:
    """SE3 pose prior — the reference's own manifold test (tests/sophus.cpp:26-44): residual(x) = log(prior_inv * x),
    differentiated on the device by dual numbers over the right perturbation (optimize_autodiff.h:48-77, sophus.h:24-26).
    prior_inv: [P, 12] (rotation matrix row-major + translation); x: [P, 12], updated by pose <- pose * exp(delta)."""
    model_id = MODEL_SE3_PRIOR
    xdim = 12

    def __init__(self, prior_inv: torch.Tensor):
        assert prior_inv.dim() == 2 and prior_inv.shape[1] == 12 and prior_inv.is_cuda
        self.P, self.n, self.m, self.dtype = prior_inv.shape[0], 6, 6, prior_inv.dtype
        self.packed = prior_inv.contiguous()

    @property
    def algorithmic_bytes_per_pass(self) -> int:
        return 12 * self.packed.element_size()


class MahaPrior:
    """Gaussian prior with a GENERAL covariance per problem: res = U (x - y) with U the upper Cholesky factor of the
    information matrix (the reference's MahaWhitenedInfoU, losses/mahalanobis.h:160-171; tests/cov.cpp:91-146).
    y: [P, n]; U: [P, n, n] upper triangular.  H = U^T U = cov^-1, so Output.Covariance() returns the prior covariance."""
    model_id = MODEL_MAHA_PRIOR

    def __init__(self, y: torch.Tensor, U: torch.Tensor):
        assert y.dim() == 2 and U.shape == (y.shape[0], y.shape[1], y.shape[1]) and y.is_cuda and U.dtype == y.dtype
        self.P, self.n = y.shape
        self.m, self.dtype = self.n, y.dtype
        self.packed = torch.cat([y, torch.triu(U).reshape(self.P, -1)], dim=1).contiguous()

    @staticmethod
    def from_covariance(y: torch.Tensor, cov: torch.Tensor):
        """`Lt = Cy.inverse().llt().matrixU()` (tests/cov.cpp:96) for a batch; model set-up, not the hot path."""
        info = torch.linalg.inv(cov.to(torch.float64))
        U = torch.linalg.cholesky(info).transpose(-1, -2).to(y.dtype)
        return MahaPrior(y, U.contiguous())

    @property
    def algorithmic_bytes_per_pass(self) -> int:
        return (self.n + self.n * self.n) * self.packed.element_size()


class TestFn:
    """The analytic functions of the reference's optimizer tests (tests/optimize_easy.cpp, tests/optimize_hard.cpp) as
    manual Accumulate callbacks with their exact Hessians: "rosenbrock", "plateau", "powell" (n = 4), "beale",
    "himmelblau".  x: [P, n] is a batch of starting points.  Exercises the bad-step / failed-solve / rollback
    branches of the LM loop on the device."""
    model_id = MODEL_TESTFN
    __test__ = False  # not a pytest class
    FUNCTIONS = {"rosenbrock": (0, 2, 1), "plateau": (1, 2, 1), "powell": (2, 4, 1), "beale": (3, 2, 3), "himmelblau": (4, 2, 2), "x_minus_2": (5, 1, 1)}

    def __init__(self, name: str, P: int, dtype=torch.float64, device="cuda"):
        fid, n, m = self.FUNCTIONS[name]
        self.name, self.P, self.n, self.m, self.dtype = name, int(P), n, m, dtype
        self.packed = torch.full((1,), float(fid), dtype=dtype, device=device)

    @property
    def algorithmic_bytes_per_pass(self) -> int:
        return 0


class DenseRowAD6(_LossMixin):
    """The DenseRow residual for n = 6 written the tinyopt way — residual only, Jacobian by device AD.
    A: [P, m, 6], b: [P, m] (natural layout)."""
    model_id = MODEL_DENSE_ROW_AD6

    def __init__(self, A: torch.Tensor, b: torch.Tensor):
        assert A.dim() == 3 and A.shape[2] == 6 and b.shape == A.shape[:2] and A.is_cuda
        self.P, self.m, self.n, self.dtype = A.shape[0], A.shape[1], 6, A.dtype
        self.packed = torch.cat([A, b[..., None]], dim=2).contiguous()

    @property
    def algorithmic_bytes_per_pass(self) -> int:
        return self.m * 7 * self.packed.element_size()


class DenseRowAD:
    """The DenseRow residual for a WIDE parameter block written the tinyopt way — residual only, the Jacobian by device
    AD in the matrix cores' operand layout (csrc/kernels.hpp JetRowModel, "chunked Jets").  n = 12 or n = 50.
    A: [P, m, n], b: [P, m] (natural layout)."""
    model_id = MODEL_DENSE_ROW_AD

    def __init__(self, A: torch.Tensor, b: torch.Tensor):
        assert A.dim() == 3 and A.shape[2] in (12, 50) and b.shape == A.shape[:2] and A.is_cuda
        self.P, self.m, self.n, self.dtype = A.shape[0], A.shape[1], A.shape[2], A.dtype
        self.packed = torch.cat([A, b[..., None]], dim=2).contiguous()

    @property
    def algorithmic_bytes_per_pass(self) -> int:
        return self.m * (self.n + 1) * self.packed.element_size()


class BundleAdjustment(_LossMixin):
    """C SE3 cameras x N 3-D points, pinhole reprojection residuals: the multi-pose problem behind
    config C5's single-pose block.  x: [P, 12*C + 3*N] = poses (rotation matrix row-major + translation) then points;
    data: [P, 8 + 3*C*N] = [f, cx, cy, 0... | uv (C, N, 2) | vis (C, N)].  Solved with the points eliminated (Schur
    complement on the reduced camera system, `toa_ba_run`); the reference would run Optimize on the full dense /
    SimplicialLDLT system (math.h:232-277).  Output.final_hessian is not produced.  ``.with_loss("cauchy", th)``: every
    observation's squared reprojection error through an M-estimator (losses/robust_norms.h:32-316; round 4)."""
    model_id = None

    def __init__(self, data: torch.Tensor, ncam: int, npts: int):
        assert data.dim() == 2 and data.shape[1] == 8 + 3 * ncam * npts and data.is_cuda
        self.P, self.ncam, self.npts, self.dtype = data.shape[0], int(ncam), int(npts), data.dtype
        self.n, self.m, self.xdim = 6 * self.ncam + 3 * self.npts, 2 * self.ncam * self.npts, 12 * self.ncam + 3 * self.npts
        self.packed = data.contiguous()

    @property
    def algorithmic_bytes_per_pass(self) -> int:
        return (3 * self.ncam * self.npts + 3 * self.npts) * self.packed.element_size()


class DenseRowNatural(_LossMixin):
    """The DenseRow family beyond one wavefront (n up to 4096): rows (a_i, b_i) in natural layout.
    64 <= n <= 128: the whole loop in one persistent kernel, a workgroup per problem, J^T J on the matrix cores
    without materialising J, blocked LDL^T by the four waves (csrc/large_fused.hip).  Beyond: J^T J through a batched rocBLAS
    GEMM, the damped solve through the workgroup LDL^T / rocSOLVER's batched Cholesky, the LM state machine of
    optimizer.h:242-539 in small kernels between them (csrc/large_n.hip).  A: [P, m, n], b: [P, m]; stored per problem as A
    then b."""
    model_id = MODEL_DENSE_ROW_NATURAL

    def __init__(self, A: torch.Tensor, b: torch.Tensor):
        assert A.dim() == 3 and b.shape == A.shape[:2] and A.is_cuda
        self.P, self.m, self.n, self.dtype = A.shape[0], A.shape[1], A.shape[2], A.dtype
        # per problem: A row-major [m][n] followed by b [m] (SURVEY §8d layout; rows stay 16-byte aligned when n % 4 == 0)
        self.packed = torch.cat([A.reshape(self.P, -1), b], dim=1).contiguous()

    @classmethod
    def from_arrays(cls, A: torch.Tensor, b: torch.Tensor) -> "DenseRowNatural":
        return cls(A, b)

    @property
    def algorithmic_bytes_per_pass(self) -> int:
        return self.m * (self.n + 1) * self.packed.element_size()


_MODELS = (BundleAdjustmentLists, JitModel, TestFn, MahaPrior, SE3Prior, DenseRow, GaussianPrior, Sqrt2, SE3Reproj, CircleFit, DenseRowAD6, DenseRowAD, DenseRowNatural, BundleAdjustment)


@dataclass
class Output:
    """tinyopt::Output (output.h:26-145), one row per problem; tensors live on the GPU.
    final_inlier_ratio = Output::final_cost.inlier_ratio (cost.h:84-95)."""
    stop_reason: torch.Tensor
    num_iters: torch.Tensor
    num_failures: torch.Tensor
    num_consec_failures: torch.Tensor
    final_cost: torch.Tensor
    final_num_residuals: torch.Tensor
    final_rerr_dec: torch.Tensor
    final_hessian: Optional[torch.Tensor] = None
    errs: Optional[torch.Tensor] = None
    deltas2: Optional[torch.Tensor] = None
    successes: Optional[torch.Tensor] = None
    counters: Optional[torch.Tensor] = None  # [acc passes streamed, eval passes, solves, problems, Builds served from the memo, 3 reserved]
    final_inlier_ratio: Optional[torch.Tensor] = None

    def Covariance(self, rescaled: bool = False):
        """Output::Covariance (output.h:80-94): inverse of the final undamped Hessian per problem; with
        rescaled=True multiplied by eps^2/(#eps - dims) where #residuals > dims.  Returns (C [P,n,n], ok [P]);
        ok == 0 where H is not invertible (std::nullopt in the reference)."""
        if self.final_hessian is None:
            raise ValueError("final_hessian was not saved (options.hessian.save_last)")
        C, ok = inv_cov(self.final_hessian)
        if rescaled:
            n = self.final_hessian.shape[1]
            nres = self.final_num_residuals.to(torch.float64)
            scale = torch.where(nres > n, self.final_cost * self.final_cost / (nres - n).clamp(min=1), torch.ones_like(nres))
            C = C * scale[:, None, None]
        return C, ok

    def Succeeded(self) -> torch.Tensor:  # output.h:30
        return self.stop_reason >= 0

    def Converged(self) -> torch.Tensor:  # output.h:33-35
        return (self.stop_reason >= 1) & (self.stop_reason < 5)


def _check_call(x: torch.Tensor, cost):
    if not isinstance(cost, _MODELS):
        raise TypeError("cost must be a device model (DenseRow, GaussianPrior, Sqrt2, ...); host callables cannot run "
                        "on the GPU path")
    if not x.is_cuda or not x.is_contiguous():
        raise ValueError("x must be a contiguous GPU tensor")
    P, xd = x.shape
    if xd != getattr(cost, "xdim", cost.n) or P != cost.P or x.dtype != cost.dtype:
        raise ValueError("x shape/dtype does not match the model")  # reference: std::invalid_argument


def _alloc_output(P: int, n: int, options: Options, history: bool, dev) -> Output:
    i32 = dict(dtype=torch.int32, device=dev)
    f64 = dict(dtype=torch.float64, device=dev)
    out = Output(
        stop_reason=torch.zeros(P, **i32), num_iters=torch.zeros(P, **i32), num_failures=torch.zeros(P, **i32),
        num_consec_failures=torch.zeros(P, **i32), final_cost=torch.zeros(P, **f64),
        final_num_residuals=torch.zeros(P, **i32), final_rerr_dec=torch.zeros(P, **f64),
        counters=torch.zeros(8, dtype=torch.int64, device=dev),
        final_inlier_ratio=torch.ones(P, dtype=torch.float32, device=dev))
    if options.hessian.save_last:
        out.final_hessian = torch.zeros(P, n, n, **f64)
    if history:
        hs = options.max_iters + 2
        out.errs = torch.zeros(P, hs, **f64)
        out.deltas2 = torch.zeros(P, hs, **f64)
        out.successes = torch.zeros(P, hs, dtype=torch.uint8, device=dev)
    return out


def _results_pod(out: Output) -> ToaResults:
    res = ToaResults()
    res.stop_reason = out.stop_reason.data_ptr()
    res.num_iters = out.num_iters.data_ptr()
    res.num_failures = out.num_failures.data_ptr()
    res.num_consec_failures = out.num_consec_failures.data_ptr()
    res.final_cost = out.final_cost.data_ptr()
    res.final_num_residuals = out.final_num_residuals.data_ptr()
    res.final_rerr_dec = out.final_rerr_dec.data_ptr()
    res.final_hessian = out.final_hessian.data_ptr() if out.final_hessian is not None else None
    res.errs = out.errs.data_ptr() if out.errs is not None else None
    res.deltas2 = out.deltas2.data_ptr() if out.deltas2 is not None else None
    res.successes = out.successes.data_ptr() if out.successes is not None else None
    res.hist_stride = out.errs.shape[1] if out.errs is not None else 0
    res.final_inlier_ratio = out.final_inlier_ratio.data_ptr() if out.final_inlier_ratio is not None else None
    return res


def Optimize(x: torch.Tensor, cost, options: Optional[Options] = None, *, history: bool = False,
             ctx: Optional[Context] = None, out: Optional[Output] = None, splits: Optional[int] = None,
             zero_counters: bool = True) -> Output:
    """``tinyopt::Optimize(x, cost, options)`` (optimize.h:16-77) for a batch of independent problems.

    x: [P, n] GPU tensor, updated IN PLACE (the reference takes x by non-const reference).
    cost: a device model (``DenseRow``, ...).  Returns the per-problem Output.  One kernel launch,
    asynchronous on torch's current stream — except ``DenseRowNatural`` beyond n = 128, a host loop over passes that returns
    when the solve is done (it enqueues two passes ahead of the stop counts where every stage is a kernel of this library).  ``out.counters`` is zeroed here and added to by the
    library (every path accumulates); ``zero_counters=False`` skips that fill launch for a caller whose ``out`` is fresh or who
    wants running totals (it is ~8 us of a 50 us single-problem solve; the C-ABI takes the counters as they are).
    """
    options = options or Options()
    _check_call(x, cost)
    P, n = x.shape[0], cost.n
    ctx = ctx or default_context(x.device.index)
    if isinstance(cost, BundleAdjustment):
        if options.has_host_controls() or splits is not None:
            raise ValueError("BundleAdjustment runs as one launch per solve: no stop callbacks / splits")
        pod = options.to_pod()
        pod.save_last = 0
        if out is None:
            import copy
            o2 = copy.deepcopy(options)
            o2.hessian.save_last = False
            out = _alloc_output(P, 1, o2, history, x.device)
        elif zero_counters:
            out.counters.zero_()
        res = _results_pod(out)
        _apply_loss(ctx, cost)
        check(ctx.lib.toa_ba_run(ctx.h, _dtype_code(x.dtype), cost.ncam, cost.npts, P, cost.packed.data_ptr(), x.data_ptr(),
                                 C.byref(pod), C.byref(res), out.counters.data_ptr()))
        return out
    if isinstance(cost, BundleAdjustmentLists):
        if options.stop_callback is not None or options.stop_callback2 is not None or splits is not None:
            raise ValueError("BundleAdjustmentLists: no stop callbacks / splits (max_duration_ms is honoured)")
        pod = options.to_pod()
        pod.save_last = 0
        if out is None:
            import copy
            o2 = copy.deepcopy(options)
            o2.hessian.save_last = False
            out = _alloc_output(P, 1, o2, history, x.device)
        elif zero_counters:
            out.counters.zero_()
        res = _results_pod(out)
        _apply_loss(ctx, cost)
        check(ctx.lib.toa_ba_lists_run(ctx.h, _dtype_code(x.dtype), cost.ncam, cost.npts, cost.nobs, P, cost.intr.data_ptr(),
                                       cost.obs_cam.data_ptr(), cost.obs_pt.data_ptr(), cost.obs_uv.data_ptr(), x.data_ptr(),
                                       C.byref(pod), C.byref(res), out.counters.data_ptr(), float(options.max_duration_ms or 0.0)))
        return out
    if isinstance(cost, JitModel):
        if options.has_host_controls():   # the stepping form of the run-time model (toa_jit_lm_begin / step / stop)
            if splits is not None or out is not None:
                raise ValueError("stop callbacks / max_duration_ms run through the stepping form: splits / out are not taken")
            return _optimize_with_host_controls(x, cost, options, history, ctx)
        pod = options.to_pod()
        if out is None:
            out = _alloc_output(P, n, options, history, x.device)
        elif zero_counters:
            out.counters.zero_()
        res = _results_pod(out)
        _apply_loss(ctx, cost)
        if splits is None:   # the library decides (row-split for few, huge problems: P * 4 <= #CUs and m >= 512)
            check(ctx.lib.toa_jit_lm_run(ctx.h, cost.res._h, cost.items, P, cost.packed.data_ptr(), x.data_ptr(), C.byref(pod),
                                         C.byref(res), out.counters.data_ptr()))
        else:                # explicit row-split execution with `splits` chunks per problem (0 = automatic count)
            check(ctx.lib.toa_jit_lm_run_split(ctx.h, cost.res._h, cost.items, P, cost.packed.data_ptr(), x.data_ptr(), C.byref(pod),
                                               C.byref(res), out.counters.data_ptr(), int(splits)))
        return out
    if options.has_host_controls():
        if splits is not None or out is not None:
            raise ValueError("stop callbacks / max_duration_ms run through the stepping form: splits / out are not taken")
        return _optimize_with_host_controls(x, cost, options, history, ctx)
    pod = options.to_pod()
    if out is None:
        out = _alloc_output(P, n, options, history, x.device)
    elif zero_counters:
        out.counters.zero_()
    res = _results_pod(out)
    _apply_loss(ctx, cost)
    if splits is None:   # the library decides (row-split for few, huge problems)
        check(ctx.lib.toa_lm_run(ctx.h, cost.model_id, _dtype_code(x.dtype), n, cost.m, P, cost.packed.data_ptr(),
                                 x.data_ptr(), C.byref(pod), C.byref(res), out.counters.data_ptr()))
    else:                # explicit row-split execution with `splits` chunks per problem (0 = automatic count)
        check(ctx.lib.toa_lm_run_split(ctx.h, cost.model_id, _dtype_code(x.dtype), n, cost.m, P, cost.packed.data_ptr(),
                                       x.data_ptr(), C.byref(pod), C.byref(res), out.counters.data_ptr(), int(splits)))
    return out


class Optimizer:
    """The reference's class / stepping form: ``lm::Optimizer<H_t> optimizer(options)`` then ``optimizer.Step(x, acc, out)``
    one loop pass at a time (optimizer.h:199,331-539), or ``optimizer(x, cost, max_iters)`` for a bounded run — for a
    batch of independent problems.  The per-problem LM state (damping, counters, last step, ...) lives on the device
    between steps; x is updated in place at every step; ``out`` fills in as problems finish."""

    def __init__(self, x: torch.Tensor, cost, options: Optional[Options] = None, *, history: bool = False,
                 ctx: Optional[Context] = None):
        self.options = options or Options()
        _check_call(x, cost)
        self.x, self.cost = x, cost
        self.ctx = ctx or default_context(x.device.index)
        self.pod = self.options.to_pod()
        P, n = x.shape[0], cost.n
        self.out = _alloc_output(P, n, self.options, history, x.device)
        self._res = _results_pod(self.out)
        nbytes = self.ctx.lib.toa_lm_state_bytes(_dtype_code(x.dtype), n, P)
        self._state = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=x.device)
        self._active = torch.zeros(1, dtype=torch.int32, device=x.device)
        _apply_loss(self.ctx, cost)
        self._jit = isinstance(cost, JitModel)
        if self._jit:
            check(self.ctx.lib.toa_jit_lm_begin(self.ctx.h, cost.res._h, cost.items, P, cost.packed.data_ptr(), x.data_ptr(),
                                                C.byref(self.pod), C.byref(self._res), self._state.data_ptr()))
        else:
            check(self.ctx.lib.toa_lm_begin(self.ctx.h, cost.model_id, _dtype_code(x.dtype), n, cost.m, P, cost.packed.data_ptr(),
                                            x.data_ptr(), C.byref(self.pod), C.byref(self._res), self._state.data_ptr()))

    def Step(self, sync: bool = True) -> Optional[int]:
        """One pass of the loop body for every running problem.  Returns how many are still running (host sync), or
        None with sync=False (stream-ordered, nothing read back)."""
        x, cost = self.x, self.cost
        self._active.zero_()
        _apply_loss(self.ctx, cost)
        if self._jit:
            check(self.ctx.lib.toa_jit_lm_step(self.ctx.h, cost.res._h, cost.items, x.shape[0], cost.packed.data_ptr(), x.data_ptr(),
                                               C.byref(self.pod), C.byref(self._res), self.out.counters.data_ptr(),
                                               self._state.data_ptr(), self._active.data_ptr()))
        else:
            check(self.ctx.lib.toa_lm_step(self.ctx.h, cost.model_id, _dtype_code(x.dtype), cost.n, cost.m, x.shape[0],
                                           cost.packed.data_ptr(), x.data_ptr(), C.byref(self.pod), C.byref(self._res),
                                           self.out.counters.data_ptr(), self._state.data_ptr(), self._active.data_ptr()))
        return int(self._active.item()) if sync else None

    def __call__(self, max_iters: Optional[int] = None) -> Output:
        """``optimizer(x, cost, max_iters)``: step until every problem has stopped (or max_iters passes were made)."""
        limit = self.options.max_iters + 2 if max_iters is None else int(max_iters)
        for _ in range(limit):
            if self.Step() == 0:
                break
        return self.out

    def step_info(self, vectors: bool = False):
        """What the reference hands its stop callbacks after an iteration (optimizer.h:529-534): per problem the cost
        ``err``, ``dx_norm2``, ``grad_norm2`` (float64 tensors [P]) and, with vectors=True, ``dx`` and ``g`` [P, n]."""
        x, cost = self.x, self.cost
        P, n = x.shape[0], cost.n
        f64 = dict(dtype=torch.float64, device=x.device)
        err, dx2, g2 = torch.zeros(P, **f64), torch.zeros(P, **f64), torch.zeros(P, **f64)
        dx = torch.zeros(P, n, dtype=x.dtype, device=x.device) if vectors else None
        g = torch.zeros(P, n, dtype=x.dtype, device=x.device) if vectors else None
        check(self.ctx.lib.toa_lm_step_info(self.ctx.h, _dtype_code(x.dtype), n, P, self._state.data_ptr(), err.data_ptr(),
                                            dx2.data_ptr(), g2.data_ptr(), dx.data_ptr() if vectors else None,
                                            g.data_ptr() if vectors else None))
        return err, dx2, g2, dx, g

    def step_log(self):
        """The rest of what the reference's log line prints (optimizer.h:463-516): per problem the damping lambda, the residual
        count of the iteration's cost and its inlier residuals."""
        x, P = self.x, self.x.shape[0]
        lam = torch.zeros(P, dtype=torch.float64, device=x.device)
        nres = torch.zeros(P, dtype=torch.int32, device=x.device)
        ninl = torch.zeros(P, dtype=torch.int32, device=x.device)
        check(self.ctx.lib.toa_lm_step_log(self.ctx.h, _dtype_code(x.dtype), self.cost.n, P, self._state.data_ptr(), lam.data_ptr(),
                                           nres.data_ptr(), ninl.data_ptr()))
        return lam, nres, ninl

    def stop(self, request: torch.Tensor) -> None:
        """End the still-running problems p with request[p] != 0 (int32, a StopReason such as kUserStopped / kTimedOut):
        their Output rows are finalised exactly as for a problem that stops by itself."""
        x, cost = self.x, self.cost
        request = request.to(device=x.device, dtype=torch.int32).contiguous()
        if self._jit:
            check(self.ctx.lib.toa_jit_lm_stop(self.ctx.h, cost.res._h, cost.items, x.shape[0], cost.packed.data_ptr(), x.data_ptr(),
                                               C.byref(self.pod), C.byref(self._res), self.out.counters.data_ptr(),
                                               self._state.data_ptr(), request.data_ptr()))
            return
        check(self.ctx.lib.toa_lm_stop(self.ctx.h, cost.model_id, _dtype_code(x.dtype), cost.n, cost.m, x.shape[0],
                                       cost.packed.data_ptr(), x.data_ptr(), C.byref(self.pod), C.byref(self._res),
                                       self.out.counters.data_ptr(), self._state.data_ptr(), request.data_ptr()))


def _optimize_with_host_controls(x, cost, options: Options, history: bool, ctx: Context) -> Output:
    """Optimize() when Options carries host-side stop controls: the loop of OptimizeAcc (optimizer.h:266-310) driven
    from the host over the stepping form.  After every pass: the callbacks see (err, |dx|^2, |g|^2) / (err, dx, g) of
    each problem that no numeric stop test has ended (the else-if chain of optimizer.h:519-534), and the accumulated
    wall time is checked against max_duration_ms (one clock for the batch) — kTimedOut overrides whatever reason a
    problem picked up in that same pass, as the unconditional assignment at optimizer.h:303-305 does."""
    import time
    opt = Optimizer(x, cost, options, history=history, ctx=ctx)
    out = opt.out
    P = x.shape[0]
    want_vec = options.stop_callback2 is not None
    duration_ms = 0.0
    running_before = torch.ones(P, dtype=torch.bool, device=x.device)
    lg = options.log
    if lg.enable:   # the problems whose iterations are logged, and what the line needs from one iteration to the next
        logged = [0] if lg.problems is None else (list(range(P)) if lg.problems == "all" else [int(q) for q in lg.problems])
        logged = [q for q in logged if 0 <= q < P]
        sink = lg.sink or print
        eps = 1e-4 if x.dtype == torch.float32 else float(np.float32(1e-7))    # FloatEpsilon<Scalar> (math.h:297-301)
        big = float(np.finfo(np.float32 if x.dtype == torch.float32 else np.float64).max)
        final_cost = {q: big for q in logged}
    for it in range(options.max_iters + 2):
        t0 = time.perf_counter()
        x_before = x.clone() if (lg.enable and lg.print_x) else None   # (the line shows the x the iteration STARTED from: it is formed inside Step)
        active = opt.Step()
        if lg.enable and logged:
            # optimizer.h:463-516, one line per logged problem that made this iteration (the same fields in the same order)
            err_t, dx2_t, g2_t, dxv_t, _ = opt.step_info(vectors=lg.print_dx)
            lam_t, nres_t, ninl_t = opt.step_log()
            torch.cuda.synchronize(x.device)
            took = (time.perf_counter() - t0) * 1e3
            rb = running_before.cpu().numpy()
            for q in logged:
                if not rb[q]:
                    continue
                err, dxn2, gn2 = float(err_t[q]), float(dx2_t[q]), float(g2_t[q])
                fc = final_cost[q]
                derr = err - fc
                good = derr < 0.0                                                        # :428-429
                rel = (fc - err) / fc if (fc > eps and fc < big) else 0.0                # :431-434
                line = ""
                if lg.print_emoji:
                    line += ("\u2139\ufe0f" if it == 0 else "\u2705") if (good or it == 0) else "\u274c"
                line += f"#{it} "
                if lg.print_x:
                    line += "x:[" + " ".join(f"{float(v):.6g}" for v in x_before[q].cpu()) + "] "
                line += f"{lg.e}:{err:.4e} n:{int(nres_t[q])} d{lg.e}:{0.0 if it == 0 else derr:+.2e} r{lg.e}:{rel:+.1e} "
                line += f"|\u03b4x|:{dxn2 ** 0.5:.2e} "
                if lg.print_dx:
                    line += "\u03b4x:[" + " ".join(f"{float(v):.6g}" for v in dxv_t[q].cpu()) + "] "
                if options.min_grad_norm2 > 0:                                            # has_grad_norm2 (:413-415)
                    line += f"|\u2207|:{gn2 ** 0.5:.2e} "
                if options.solver_type == Options.LevenbergMarquardt and float(lam_t[q]) > 0:
                    line += f"\u25cb:{1.0 / float(lam_t[q]):.2e} "                       # SolverLM::stateAsString (lm.h:150-154)
                if lg.print_inliers:
                    nr = max(int(nres_t[q]), 1)
                    line += f"in:{100.0 * int(ninl_t[q]) / nr:.2f}% ({int(ninl_t[q])}) "
                if lg.print_t:
                    duration_so_far = duration_ms + took
                    line += f"\u03c4:{duration_so_far:.2f} "
                sink(line)
                if good or it == 0:
                    final_cost[q] = err                                                   # :441-446
        running = out.stop_reason == int(StopReason.kNone)
        running &= running_before          # a row reports kNone until its problem stops
        req = torch.zeros(P, dtype=torch.int32)
        # The reference evaluates the callbacks inside Step on EVERY iteration (optimizer.h:529-534) and labels kMaxIters only
        # after the loop, when stop_reason is still kNone (:320-321): a problem that used up its iterations in this very pass
        # is consulted too, and a callback returning true makes it kUserStopped, not kMaxIters.
        just_max = running_before & (out.stop_reason == int(StopReason.kMaxIters))
        consult = running | just_max
        if bool(consult.any()) and (options.stop_callback is not None or options.stop_callback2 is not None):
            err, dx2, g2, dxv, gv = opt.step_info(vectors=want_vec)
            err_h, dx2_h, g2_h = err.cpu().numpy(), dx2.cpu().numpy(), g2.cpu().numpy()
            dx_h = dxv.cpu().numpy() if want_vec else None
            g_h = gv.cpu().numpy() if want_vec else None
            just_max_h = just_max.cpu().numpy()
            for p in torch.nonzero(consult).flatten().tolist():
                stop = False
                if options.stop_callback is not None:
                    stop = bool(options.stop_callback(float(err_h[p]), float(dx2_h[p]), float(g2_h[p])))
                if not stop and options.stop_callback2 is not None:   # stop_callback2(float(err), dx.cast<float>(), g.cast<float>())
                    stop = bool(options.stop_callback2(float(np.float32(err_h[p])), dx_h[p].astype("float32"), g_h[p].astype("float32")))
                if stop:
                    if just_max_h[p]:
                        out.stop_reason[p] = int(StopReason.kUserStopped)   # already finalised: only the label changes
                    else:
                        req[p] = int(StopReason.kUserStopped)
        torch.cuda.synchronize(x.device)
        duration_ms += (time.perf_counter() - t0) * 1e3
        timed_out = options.max_duration_ms > 0 and duration_ms > options.max_duration_ms
        if timed_out:
            req[running.cpu()] = int(StopReason.kTimedOut)
        if bool(req.any()):
            opt.stop(req)
        if timed_out:   # problems that stopped by themselves in this very pass: the reference overwrites their reason too
            just = running_before & ~running
            out.stop_reason[just] = int(StopReason.kTimedOut)
            break
        running_before = running & (req.to(x.device) == 0)
        if not bool(running_before.any()):
            break
    return out


def accumulate(cost, x: torch.Tensor, want_grad: bool = True, ctx: Optional[Context] = None):
    """The Accumulate callback ``acc(x, grad, H) -> Cost`` (docs/API.md:37-57) for a batch.
    Returns (g [P,n], H [P,n,n], cost [P] float64, nres [P]); g/H are None when want_grad is False."""
    ctx = ctx or default_context(x.device.index)
    P, n = x.shape[0], cost.n
    dev = x.device
    g = torch.zeros(P, n, dtype=x.dtype, device=dev) if want_grad else None
    H = torch.zeros(P, n, n, dtype=x.dtype, device=dev) if want_grad else None
    c = torch.zeros(P, dtype=torch.float64, device=dev)
    nres = torch.zeros(P, dtype=torch.int32, device=dev)
    _apply_loss(ctx, cost)
    if isinstance(cost, JitModel):
        check(ctx.lib.toa_jit_accumulate(ctx.h, cost.res._h, cost.items, P, cost.packed.data_ptr(), x.data_ptr(), int(want_grad),
                                         g.data_ptr() if want_grad else None, H.data_ptr() if want_grad else None,
                                         c.data_ptr(), nres.data_ptr()))
        return g, H, c, nres
    check(ctx.lib.toa_accumulate(ctx.h, cost.model_id, _dtype_code(x.dtype), n, cost.m, P, cost.packed.data_ptr(),
                                 x.data_ptr(), int(want_grad), g.data_ptr() if want_grad else None,
                                 H.data_ptr() if want_grad else None, c.data_ptr(), nres.data_ptr()))
    return g, H, c, nres


def solve_damped(H: torch.Tensor, g: torch.Tensor, scale: float = 1.0, ctx: Optional[Context] = None):
    """SolverLM damping (lm.h:108-117, H_ii *= scale) + SolverGN::Solve (gn.h:150-171) for a batch.
    Returns (dx [P,n], ok [P] int32)."""
    ctx = ctx or default_context(H.device.index)
    P, n = g.shape
    H, g = H.contiguous(), g.contiguous()
    dx = torch.zeros_like(g)
    ok = torch.zeros(P, dtype=torch.int32, device=g.device)
    check(ctx.lib.toa_solve_damped(ctx.h, _dtype_code(g.dtype), n, P, H.data_ptr(), g.data_ptr(), float(scale),
                                   dx.data_ptr(), ok.data_ptr()))
    return dx, ok


LOSS_KINDS = {"l2": 0, "truncated": 1, "huber": 2, "tukey": 3, "arctan": 4, "cauchy": 5, "geman_mcclure": 6,
              "blake_zisserman": 7}


def robust_norm(kind: str, n2: torch.Tensor, th2: float, ctx: Optional[Context] = None):
    """The reference's M-estimators in their ``Name(n2, th2, true)`` form (losses/robust_norms.h:32-316,
    docs/API.md:402-406): returns (loss, scale) tensors shaped like ``n2`` (squared norms, GPU, float32/64)."""
    ctx = ctx or default_context(n2.device.index)
    n2 = n2.contiguous()
    loss = torch.empty_like(n2)
    scale = torch.empty_like(n2)
    check(ctx.lib.toa_robust_norm(ctx.h, LOSS_KINDS[kind], _dtype_code(n2.dtype), n2.numel(), n2.data_ptr(), float(th2),
                                  loss.data_ptr(), scale.data_ptr()))
    return loss, scale


def inv_cov(H: torch.Tensor, ctx: Optional[Context] = None):
    """tinyopt::InvCov (math.h:41-91) for a batch: C = H^-1 via LDL^T with the reference's acceptance rule.
    Returns (C [P,n,n], ok [P] int32)."""
    ctx = ctx or default_context(H.device.index)
    H = H.contiguous()
    P, n, _ = H.shape
    Cm = torch.zeros_like(H)
    ok = torch.zeros(P, dtype=torch.int32, device=H.device)
    check(ctx.lib.toa_inv_cov(ctx.h, _dtype_code(H.dtype), n, P, H.data_ptr(), Cm.data_ptr(), ok.data_ptr()))
    return Cm, ok
