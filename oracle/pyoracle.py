"""TEST INFRASTRUCTURE ONLY — ctypes wrapper over oracle/_build/liblm_oracle.so (numpy in/out).

May be imported only by tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke().
Nothing under tinyopt_amd/ imports this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_build", "liblm_oracle.so")
PIN = os.path.join(_HERE, "_build", "pin_reference_tests")

sys.path.insert(0, os.path.dirname(_HERE))
from tinyopt_amd._capi import ToaOptions  # POD data contract only (no library load)  # noqa: E402

F32, F64 = 0, 1
_lib = None


def build(march: str | None = None, out_dir: str | None = None) -> str:
    """make -C oracle (g++).  `march`/`out_dir` let bench.py build a host-tuned copy elsewhere."""
    if march is None and out_dir is None:
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
        return LIB
    out_dir = out_dir or os.path.join(_HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "liblm_oracle_native.so")
    subprocess.run(["g++", "-std=c++17", "-O3", f"-march={march or 'native'}", "-fopenmp", "-fPIC", "-shared",
                    "-o", out, os.path.join(_HERE, "lm_oracle_capi.cpp")], check=True)
    return out


def load(path: str | None = None):
    global _lib
    if path is None and _lib is not None:
        return _lib
    p = path or LIB
    if not os.path.exists(p):
        build()
    lib = C.CDLL(p)
    vp = C.c_void_p
    lib.oracle_num_threads_max.restype = C.c_int
    lib.oracle_synth_dense_row.argtypes = [C.c_int, C.c_uint64, C.c_int64, C.c_int64, C.c_int, C.c_int, vp, vp, vp, vp]
    lib.oracle_synth_gaussian_prior.argtypes = [C.c_int, C.c_uint64, C.c_int64, C.c_int64, C.c_int, vp, vp, vp]
    lib.oracle_dense_row_accumulate.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, vp]
    lib.oracle_solve_damped.argtypes = [C.c_int, C.c_int64, C.c_int, vp, vp, C.c_double, vp, vp]
    lib.oracle_dense_row_lm.restype = C.c_double
    lib.oracle_dense_row_lm.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_int, vp, vp, vp, C.POINTER(ToaOptions),
                                        vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int]
    lib.oracle_gaussian_prior_lm.restype = C.c_double
    lib.oracle_gaussian_prior_lm.argtypes = [C.c_int, C.c_int64, C.c_int, vp, vp, vp, C.POINTER(ToaOptions), vp, vp, vp, vp, vp]
    lib.oracle_gaussian_prior_lm_hist.restype = C.c_double
    lib.oracle_gaussian_prior_lm_hist.argtypes = [C.c_int, C.c_int64, C.c_int, vp, vp, vp, C.POINTER(ToaOptions), vp, vp, vp, vp, vp,
                                                  vp, vp, vp, C.c_int]
    lib.oracle_sqrt2_lm.argtypes = [C.c_int, C.c_int64, vp, C.POINTER(ToaOptions), vp, vp, vp, vp, vp, vp, C.c_int]
    lib.oracle_se3_reproj_lm.argtypes = [C.c_int, C.c_int64, C.c_int, vp, vp, C.POINTER(ToaOptions), vp, vp, vp, vp, vp]
    lib.oracle_se3_prior_lm.argtypes = [C.c_int, C.c_int64, vp, vp, C.POINTER(ToaOptions), vp, vp, vp]
    lib.oracle_se3_prior_accumulate.argtypes = [C.c_int64, vp, vp, vp, vp, vp]
    lib.oracle_se3_log.argtypes = [C.c_int64, vp, vp]
    lib.oracle_maha_prior_lm.argtypes = [C.c_int, C.c_int64, C.c_int, vp, vp, C.POINTER(ToaOptions), vp, vp, vp, vp]
    lib.oracle_testfn_lm.argtypes = [C.c_int, C.c_int, C.c_int64, vp, C.POINTER(ToaOptions), vp, vp, vp, vp, vp, vp, vp, C.c_int]
    lib.oracle_testfn_accumulate.argtypes = [C.c_int, C.c_int, C.c_int64, vp, vp, vp, vp]
    lib.oracle_robust_norm.argtypes = [C.c_int, C.c_int, C.c_int64, vp, C.c_double, vp, vp]
    lib.oracle_se3_reproj_accumulate.argtypes = [C.c_int, C.c_int64, C.c_int, vp, vp, vp, vp, vp]
    lib.oracle_se3_plus.argtypes = [C.c_int, C.c_int64, vp, vp]
    lib.oracle_set_loss.argtypes = [C.c_int, C.c_double, vp]
    lib.oracle_ba_lm.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_int, vp, vp, C.POINTER(ToaOptions), vp, vp, vp, vp, vp, vp, vp, vp, C.c_int]
    lib.oracle_circle_fit_lm.argtypes = [C.c_int, C.c_int64, C.c_int, vp, vp, C.POINTER(ToaOptions), vp, vp, vp]
    if path is None:
        _lib = lib
    return lib


def _code(dtype) -> int:
    return F32 if np.dtype(dtype) == np.float32 else F64


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def synth_dense_row(P, n, m, dtype, seed=0x71940917, problem0=0):
    lib = load()
    A = np.empty((P, m, n), dtype)
    b = np.empty((P, m), dtype)
    x0 = np.empty((P, n), dtype)
    xs = np.empty((P, n), np.float64)
    lib.oracle_synth_dense_row(_code(dtype), seed, problem0, P, n, m, _p(A), _p(b), _p(x0), _p(xs))
    return A, b, x0, xs


def dense_row_accumulate(A, b, x, want_grad=True, loss=None, th2=0.0):
    """(g, H, cost, nres); with a loss also the inlier ratio as a fifth element."""
    lib = load()
    P, m, n = A.shape
    if loss is not None:
        inl = np.ones(P, np.float32)
        lib.oracle_set_loss(LOSS_KINDS[loss], float(th2), _p(inl))
        try:
            r = dense_row_accumulate(A, b, x, want_grad)
        finally:
            lib.oracle_set_loss(0, 0.0, None)
        return r + (inl,)
    g = np.zeros((P, n), A.dtype)
    H = np.zeros((P, n, n), A.dtype)
    cost = np.zeros(P, np.float64)
    nres = np.zeros(P, np.int32)
    lib.oracle_dense_row_accumulate(_code(A.dtype), P, n, m, _p(A), _p(b), _p(np.ascontiguousarray(x)), int(want_grad),
                                    _p(g), _p(H), _p(cost), _p(nres))
    return g, H, cost, nres


def solve_damped(H, g, scale=1.0):
    lib = load()
    P, n = g.shape
    dx = np.zeros_like(g)
    ok = np.zeros(P, np.int32)
    lib.oracle_solve_damped(_code(g.dtype), P, n, _p(np.ascontiguousarray(H)), _p(np.ascontiguousarray(g)), float(scale),
                            _p(dx), _p(ok))
    return dx, ok


def dense_row_lm(A, b, x0, pod: ToaOptions, history=False, nthreads=1, lib=None, loss=None, th2=0.0):
    """Returns dict(x, stop, iters, fails, cost, rerr, H, errs, deltas2, succ, seconds[, inlier_ratio]).
    loss / th2: an M-estimator on every residual (LOSS_KINDS name; squared threshold)."""
    lib = lib or load()
    P, m, n = A.shape
    if loss is not None:
        inl = np.ones(P, np.float32)
        lib.oracle_set_loss(LOSS_KINDS[loss], float(th2), _p(inl))
        try:
            r = dense_row_lm(A, b, x0, pod, history=history, nthreads=1, lib=lib)
        finally:
            lib.oracle_set_loss(0, 0.0, None)
        r["inlier_ratio"] = inl
        return r
    x = np.array(x0, copy=True)
    stop = np.zeros(P, np.int32)
    iters = np.zeros(P, np.int32)
    fails = np.zeros(P, np.int32)
    cost = np.zeros(P, np.float64)
    rerr = np.zeros(P, np.float64)
    Hf = np.zeros((P, n, n), np.float64) if pod.save_last else None
    hs = pod.max_iters + 2
    errs = np.zeros((P, hs), np.float64) if history else None
    d2 = np.zeros((P, hs), np.float64) if history else None
    succ = np.zeros((P, hs), np.uint8) if history else None
    secs = lib.oracle_dense_row_lm(_code(A.dtype), P, n, m, _p(A), _p(b), _p(x), C.byref(pod), _p(stop), _p(iters),
                                   _p(fails), _p(cost), _p(rerr), _p(Hf), _p(errs), _p(d2), _p(succ), hs, nthreads)
    return dict(x=x, stop=stop, iters=iters, fails=fails, cost=cost, rerr=rerr, H=Hf, errs=errs, deltas2=d2, succ=succ,
                seconds=secs)


def synth_gaussian_prior(P, n, dtype, seed=0x71940917, problem0=0):
    lib = load()
    y = np.empty((P, n), dtype)
    sigma = np.empty((P, n), dtype)
    x0 = np.empty((P, n), dtype)
    lib.oracle_synth_gaussian_prior(_code(dtype), seed, problem0, P, n, _p(y), _p(sigma), _p(x0))
    return y, sigma, x0


def gaussian_prior_lm(y, sigma, x0, pod: ToaOptions, history=False):
    """benchmarks/dense.cpp manual-callback semantics; returns dict(x, stop, iters, fails, cost, H, seconds
    [, errs, deltas2, succ])."""
    lib = load()
    P, n = y.shape
    x = np.array(x0, copy=True)
    stop = np.zeros(P, np.int32)
    iters = np.zeros(P, np.int32)
    fails = np.zeros(P, np.int32)
    cost = np.zeros(P, np.float64)
    Hf = np.zeros((P, n, n), np.float64) if pod.save_last else None
    hs = pod.max_iters + 2
    errs = np.zeros((P, hs), np.float64) if history else None
    d2 = np.zeros((P, hs), np.float64) if history else None
    succ = np.zeros((P, hs), np.uint8) if history else None
    secs = lib.oracle_gaussian_prior_lm_hist(_code(y.dtype), P, n, _p(np.ascontiguousarray(y)), _p(np.ascontiguousarray(sigma)),
                                             _p(x), C.byref(pod), _p(stop), _p(iters), _p(fails), _p(cost), _p(Hf),
                                             _p(errs), _p(d2), _p(succ), hs)
    return dict(x=x, stop=stop, iters=iters, fails=fails, cost=cost, H=Hf, seconds=secs, errs=errs, deltas2=d2, succ=succ)


def sqrt2_lm(x0, pod: ToaOptions, history=True):
    lib = load()
    x = np.array(x0, copy=True)
    P = x.shape[0]
    stop = np.zeros(P, np.int32)
    iters = np.zeros(P, np.int32)
    cost = np.zeros(P, np.float64)
    hs = pod.max_iters + 2
    errs = np.zeros((P, hs), np.float64)
    d2 = np.zeros((P, hs), np.float64)
    succ = np.zeros((P, hs), np.uint8)
    lib.oracle_sqrt2_lm(_code(x.dtype), P, _p(x), C.byref(pod), _p(stop), _p(iters), _p(cost), _p(errs), _p(d2), _p(succ), hs)
    return dict(x=x, stop=stop, iters=iters, cost=cost, errs=errs, deltas2=d2, succ=succ)


def se3_plus(poses, delta):
    """pose * exp(delta) (sophus.h:24-26) for a batch; poses [P,12] (R row-major, t), delta [P,6] (upsilon, omega)."""
    lib = load()
    out = np.array(poses, copy=True)
    lib.oracle_se3_plus(_code(out.dtype), out.shape[0], _p(out), _p(np.ascontiguousarray(delta.astype(out.dtype))))
    return out


def synth_se3_reproj(P, npts, dtype, seed=0x71940917):
    """T* = exp(xi*), xi* ~ 0.3 U(-1,1)^6; points in a 4 x 4 x [4,8] m frustum in the camera frame;
    pinhole f=500, c=(320,240); pixel noise 0.5 U(-1,1); T0 = T* exp(0.05 U(-1,1)^6).
    Returns (data [P, 8+5*npts], pose0 [P,12], pose_star [P,12])."""
    rng = np.random.default_rng(seed)
    ident = np.tile(np.concatenate([np.eye(3).ravel(), np.zeros(3)]), (P, 1))
    pstar = se3_plus(ident.astype(np.float64), 0.3 * rng.uniform(-1, 1, (P, 6)))
    p0 = se3_plus(pstar, 0.05 * rng.uniform(-1, 1, (P, 6)))
    R = pstar[:, :9].reshape(P, 3, 3)
    t = pstar[:, 9:]
    pc = np.stack([rng.uniform(-2, 2, (P, npts)), rng.uniform(-2, 2, (P, npts)), rng.uniform(4, 8, (P, npts))], -1)
    pw = np.einsum("pji,pnj->pni", R, pc - t[:, None, :])          # p_w = R^T (p_c - t)
    f, cx, cy = 500.0, 320.0, 240.0
    uv = np.stack([f * pc[..., 0] / pc[..., 2] + cx, f * pc[..., 1] / pc[..., 2] + cy], -1) + 0.5 * rng.uniform(-1, 1, (P, npts, 2))
    data = np.zeros((P, 8 + 5 * npts))
    data[:, 0], data[:, 1], data[:, 2] = f, cx, cy
    data[:, 8:] = np.concatenate([pw, uv], -1).reshape(P, -1)
    return data.astype(dtype), p0.astype(dtype), pstar.astype(dtype)


def se3_reproj_accumulate(data, poses, npts):
    lib = load()
    P = poses.shape[0]
    g = np.zeros((P, 6), poses.dtype)
    H = np.zeros((P, 6, 6), poses.dtype)
    cost = np.zeros(P, np.float64)
    lib.oracle_se3_reproj_accumulate(_code(poses.dtype), P, npts, _p(np.ascontiguousarray(data)), _p(np.ascontiguousarray(poses)),
                                     _p(g), _p(H), _p(cost))
    return g, H, cost


def se3_reproj_lm(data, pose0, npts, pod: ToaOptions):
    lib = load()
    P = pose0.shape[0]
    x = np.array(pose0, copy=True)
    stop = np.zeros(P, np.int32)
    iters = np.zeros(P, np.int32)
    cost = np.zeros(P, np.float64)
    Hf = np.zeros((P, 6, 6), np.float64) if pod.save_last else None
    inl = np.zeros(P, np.float32)
    lib.oracle_se3_reproj_lm(_code(x.dtype), P, npts, _p(np.ascontiguousarray(data)), _p(x), C.byref(pod), _p(stop), _p(iters),
                             _p(cost), _p(Hf), _p(inl))
    return dict(x=x, stop=stop, iters=iters, cost=cost, H=Hf, inlier_ratio=inl)


LOSS_KINDS = {"l2": 0, "truncated": 1, "huber": 2, "tukey": 3, "arctan": 4, "cauchy": 5, "geman_mcclure": 6,
              "blake_zisserman": 7}


def robust_norm(kind, n2, th2):
    """(loss, scale) of the reference's M-estimators (losses/robust_norms.h) for an array of squared norms."""
    lib = load()
    n2 = np.ascontiguousarray(n2)
    loss = np.empty_like(n2)
    scale = np.empty_like(n2)
    k = LOSS_KINDS[kind] if isinstance(kind, str) else int(kind)
    lib.oracle_robust_norm(k, _code(n2.dtype), n2.size, _p(n2), float(th2), _p(loss), _p(scale))
    return loss, scale


def se3_set_loss(data, kind, th2):
    """Write the robust-loss header slots of SE3Reproj data ([3] = kind, [4] = th2); returns a copy."""
    out = np.array(data, copy=True)
    out[:, 3] = LOSS_KINDS[kind] if isinstance(kind, str) else int(kind)
    out[:, 4] = th2
    return out


def se3_add_outliers(data, npts, frac, seed=7, amplitude=80.0):
    """Replace a fraction of the observed pixels by gross outliers (uniform +-amplitude px); returns (copy, mask)."""
    rng = np.random.default_rng(seed)
    out = np.array(data, copy=True)
    P = out.shape[0]
    pts = out[:, 8:].reshape(P, npts, 5)
    mask = rng.uniform(size=(P, npts)) < frac
    noise = rng.uniform(-amplitude, amplitude, (P, npts, 2)).astype(out.dtype)
    pts[..., 3:5] += noise * mask[..., None]
    out[:, 8:] = pts.reshape(P, -1)
    return out, mask


def circle_fit_lm(obs, x0, pod: ToaOptions, loss=None, th2=0.0):
    """tests/circle.cpp:32-68 for a batch; obs [P, npts, 2], x0 [P, 3].  loss / th2: M-estimator per residual."""
    lib = load()
    P, npts, _ = obs.shape
    if loss is not None:
        inl = np.ones(P, np.float32)
        lib.oracle_set_loss(LOSS_KINDS[loss], float(th2), _p(inl))
        try:
            r = circle_fit_lm(obs, x0, pod)
        finally:
            lib.oracle_set_loss(0, 0.0, None)
        r["inlier_ratio"] = inl
        return r
    x = np.array(x0, copy=True)
    stop = np.zeros(P, np.int32)
    iters = np.zeros(P, np.int32)
    cost = np.zeros(P, np.float64)
    lib.oracle_circle_fit_lm(_code(x.dtype), P, npts, _p(np.ascontiguousarray(obs)), _p(x), C.byref(pod), _p(stop), _p(iters),
                             _p(cost))
    return dict(x=x, stop=stop, iters=iters, cost=cost)


def se3_log(poses):
    """xi = (upsilon, omega) = log(pose) for [P, 12] poses (fp64)."""
    lib = load()
    poses = np.ascontiguousarray(poses, np.float64)
    xi = np.zeros((poses.shape[0], 6))
    lib.oracle_se3_log(poses.shape[0], _p(poses), _p(xi))
    return xi


def se3_compose(a, b):
    """a * b for [P, 12] poses."""
    Ra, ta = a[:, :9].reshape(-1, 3, 3), a[:, 9:]
    Rb, tb = b[:, :9].reshape(-1, 3, 3), b[:, 9:]
    R = np.einsum("pij,pjk->pik", Ra, Rb)
    t = np.einsum("pij,pj->pi", Ra, tb) + ta
    return np.concatenate([R.reshape(-1, 9), t], axis=1)


def se3_prior_accumulate(prior_inv, poses):
    lib = load()
    P = poses.shape[0]
    g = np.zeros((P, 6)); H = np.zeros((P, 6, 6)); cost = np.zeros(P)
    lib.oracle_se3_prior_accumulate(P, _p(np.ascontiguousarray(prior_inv, np.float64)), _p(np.ascontiguousarray(poses, np.float64)),
                                    _p(g), _p(H), _p(cost))
    return g, H, cost


def se3_prior_lm(prior_inv, pose0, pod: ToaOptions):
    """tests/sophus.cpp:26-44 for a batch: residual log(prior_inv * x)."""
    lib = load()
    x = np.array(pose0, copy=True)
    P = x.shape[0]
    stop = np.zeros(P, np.int32); iters = np.zeros(P, np.int32); cost = np.zeros(P, np.float64)
    lib.oracle_se3_prior_lm(_code(x.dtype), P, _p(np.ascontiguousarray(prior_inv.astype(x.dtype))), _p(x), C.byref(pod), _p(stop),
                            _p(iters), _p(cost))
    return dict(x=x, stop=stop, iters=iters, cost=cost)


def maha_prior_data(y, cov):
    """[P, n + n*n] = y then U (row-major), U = upper Cholesky factor of cov^-1 (tests/cov.cpp:96: `Cy.inverse().llt().matrixU()`)."""
    y = np.asarray(y)
    P, n = y.shape
    U = np.stack([np.linalg.cholesky(np.linalg.inv(np.asarray(c, np.float64))).T for c in cov])
    return np.concatenate([y, U.reshape(P, n * n)], axis=1).astype(y.dtype)


def maha_prior_lm(data, x0, pod: ToaOptions):
    lib = load()
    x = np.array(x0, copy=True)
    P, n = x.shape
    stop = np.zeros(P, np.int32); iters = np.zeros(P, np.int32); cost = np.zeros(P, np.float64)
    Hf = np.zeros((P, n, n), np.float64) if pod.save_last else None
    lib.oracle_maha_prior_lm(_code(x.dtype), P, n, _p(np.ascontiguousarray(data)), _p(x), C.byref(pod), _p(stop), _p(iters),
                             _p(cost), _p(Hf))
    return dict(x=x, stop=stop, iters=iters, cost=cost, H=Hf)


TESTFNS = {"rosenbrock": 0, "plateau": 1, "powell": 2, "beale": 3, "himmelblau": 4, "x_minus_2": 5}


def testfn_options(name):
    """The reference test's own options for each function (tests/optimize_easy.cpp, optimize_hard.cpp) as a
    dict of (attribute path, value) for tinyopt_amd.Options."""
    return {
        "rosenbrock": {"max_iters": 200, "min_rerr_dec": 0.0, "max_consec_failures": 20},
        "plateau": {"lm.damping_init": 1e-6},
        "powell": {"max_iters": 200, "max_consec_failures": 0, "min_error": 1e-30, "min_rerr_dec": 1e-30, "lm.damping_init": 1e-1},
        "beale": {"max_iters": 200, "max_consec_failures": 0, "min_error": 1e-30, "lm.damping_init": 1e-3},
        "himmelblau": {"max_iters": 200, "max_consec_failures": 0, "min_error": 1e-30, "lm.damping_init": 1e-4},
    }[name]


def testfn_lm(name, x0, pod: ToaOptions, hist_stride=0):
    """LM on one of the reference's analytic test functions for a batch of starts x0 [P, n]."""
    lib = load()
    x = np.array(x0, copy=True)
    P = x.shape[0]
    stop = np.zeros(P, np.int32); iters = np.zeros(P, np.int32); fails = np.zeros(P, np.int32)
    cost = np.zeros(P, np.float64)
    errs = np.zeros((P, hist_stride)) if hist_stride else None
    d2 = np.zeros((P, hist_stride)) if hist_stride else None
    succ = np.zeros((P, hist_stride), np.uint8) if hist_stride else None
    lib.oracle_testfn_lm(TESTFNS[name], _code(x.dtype), P, _p(x), C.byref(pod), _p(stop), _p(iters), _p(fails), _p(cost),
                         _p(errs), _p(d2), _p(succ), int(hist_stride))
    return dict(x=x, stop=stop, iters=iters, fails=fails, cost=cost, errs=errs, deltas2=d2, succ=succ)


def testfn_accumulate(name, x):
    lib = load()
    x = np.ascontiguousarray(x)
    P, n = x.shape
    g = np.zeros((P, n), x.dtype); H = np.zeros((P, n, n), x.dtype); cost = np.zeros(P, np.float64)
    lib.oracle_testfn_accumulate(TESTFNS[name], _code(x.dtype), P, _p(x), _p(g), _p(H), _p(cost))
    return g, H, cost


def run_pin_tests() -> subprocess.CompletedProcess:
    if not os.path.exists(PIN):
        build()
    return subprocess.run([PIN], capture_output=True, text=True)


def synth_ba(P, ncam, npts, dtype, seed=0x71940917, noise_px=0.5, pose_pert=0.02, point_pert=0.05, invisible=0.0):
    """Synthetic bundle-adjustment scenes: ncam cameras on an arc looking at a cloud of npts points around the origin,
    pinhole f = 500, c = (320, 240), pixel noise noise_px * U(-1, 1); the start is the planted scene perturbed by
    exp(pose_pert * U(-1,1)^6) on every pose and point_pert * U(-1,1)^3 on every point.  `invisible`: fraction of the
    observations dropped (vis = 0).  Returns (data [P, 8 + 3*ncam*npts], x0 [P, 12*ncam + 3*npts], xstar)."""
    rng = np.random.default_rng(seed)
    f, cx, cy = 500.0, 320.0, 240.0
    data = np.zeros((P, 8 + 3 * ncam * npts))
    x0 = np.zeros((P, 12 * ncam + 3 * npts))
    xs = np.zeros_like(x0)
    ident = np.concatenate([np.eye(3).ravel(), np.zeros(3)])
    for p in range(P):
        pts = rng.uniform(-1, 1, (npts, 3)) * np.array([1.5, 1.0, 1.0])
        poses = np.zeros((ncam, 12))
        for c in range(ncam):
            ang = (c - (ncam - 1) / 2) * 0.25
            # camera at distance 6 from the origin, rotated about the y axis by `ang`, plus a small random twist
            Ry = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
            base = np.concatenate([Ry.ravel(), np.array([0.0, 0.0, 6.0])])[None, :]
            poses[c] = se3_plus(base, 0.05 * rng.uniform(-1, 1, (1, 6)))[0]
        R = poses[:, :9].reshape(ncam, 3, 3)
        pc = np.einsum("cij,nj->cni", R, pts) + poses[:, None, 9:]
        uv = np.stack([f * pc[..., 0] / pc[..., 2] + cx, f * pc[..., 1] / pc[..., 2] + cy], -1)
        uv = uv + noise_px * rng.uniform(-1, 1, uv.shape)
        vis = (rng.uniform(size=(ncam, npts)) >= invisible).astype(np.float64)
        vis[:2] = 1.0                                   # every point is seen by at least two cameras
        data[p, 0], data[p, 1], data[p, 2] = f, cx, cy
        data[p, 8:8 + 2 * ncam * npts] = uv.ravel()
        data[p, 8 + 2 * ncam * npts:] = vis.ravel()
        xs[p, :12 * ncam] = poses.ravel()
        xs[p, 12 * ncam:] = pts.ravel()
        x0[p, :12 * ncam] = se3_plus(poses, pose_pert * rng.uniform(-1, 1, (ncam, 6))).ravel()
        x0[p, 12 * ncam:] = (pts + point_pert * rng.uniform(-1, 1, pts.shape)).ravel()
    return data.astype(dtype), x0.astype(dtype), xs.astype(dtype)


def ba_lm(data, x0, ncam, npts, pod: ToaOptions, history=True, lib=None, loss=None, th2=0.0):
    """Bundle adjustment solved the reference's way: dense (6C + 3N)^2 Hessian + dense LDL^T (oracle/ba.hpp).
    loss / th2: an M-estimator on every observation's squared norm (losses/robust_norms.h:20-26); adds "inlier_ratio"."""
    lib = lib or load()
    if loss is not None:
        inl = np.ones(np.asarray(x0).shape[0], np.float32)
        lib.oracle_set_loss(LOSS_KINDS[loss], float(th2), _p(inl))
        try:
            r = ba_lm(data, x0, ncam, npts, pod, history=history, lib=lib)
        finally:
            lib.oracle_set_loss(0, 0.0, None)
        r["inlier_ratio"] = inl
        return r
    x = np.array(x0, copy=True)
    P = x.shape[0]
    stop = np.zeros(P, np.int32); iters = np.zeros(P, np.int32); fails = np.zeros(P, np.int32)
    cost = np.zeros(P, np.float64); nres = np.zeros(P, np.int32)
    hs = pod.max_iters + 2
    errs = np.zeros((P, hs)) if history else None
    d2 = np.zeros((P, hs)) if history else None
    succ = np.zeros((P, hs), np.uint8) if history else None
    lib.oracle_ba_lm(_code(x.dtype), P, ncam, npts, _p(np.ascontiguousarray(data)), _p(x), C.byref(pod), _p(stop), _p(iters),
                     _p(fails), _p(cost), _p(nres), _p(errs), _p(d2), _p(succ), hs)
    return dict(x=x, stop=stop, iters=iters, fails=fails, cost=cost, nres=nres, errs=errs, deltas2=d2, succ=succ)
