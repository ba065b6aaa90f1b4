"""Synthetic inputs of the bench workloads that are not DenseRow (which has `toa_dense_row_synth` on the device): the SE3
reprojection block of BASELINE config C5 and bundle-adjustment scenes, dense-mask and visibility-list form.
Product-side generators (numpy, float64 arithmetic, cast at the end): `bench.py --no-cpu` needs nothing from oracle/."""
from __future__ import annotations

import numpy as np


def se3_exp(d: np.ndarray):
    """exp of tangent vectors d [..., 6] = (upsilon, omega) (Sophus order) -> (R [..., 3, 3], t [..., 3])."""
    u, w = d[..., :3], d[..., 3:]
    th2 = (w * w).sum(-1)
    th = np.sqrt(th2)
    small = th2 < 1e-10
    ths = np.where(small, 1.0, th)
    A = np.where(small, 1 - th2 / 6, np.sin(ths) / ths)
    B = np.where(small, 0.5 - th2 / 24, (1 - np.cos(ths)) / np.where(small, 1.0, th2))
    Cc = np.where(small, 1 / 6 - th2 / 120, (ths - np.sin(ths)) / np.where(small, 1.0, th2 * ths))
    K = np.zeros(d.shape[:-1] + (3, 3))
    K[..., 0, 1], K[..., 0, 2], K[..., 1, 0] = -w[..., 2], w[..., 1], w[..., 2]
    K[..., 1, 2], K[..., 2, 0], K[..., 2, 1] = -w[..., 0], -w[..., 1], w[..., 0]
    K2 = K @ K
    eye = np.eye(3)
    R = eye + A[..., None, None] * K + B[..., None, None] * K2
    V = eye + B[..., None, None] * K + Cc[..., None, None] * K2
    return R, (V @ u[..., None])[..., 0]


def se3_plus(pose: np.ndarray, d: np.ndarray) -> np.ndarray:
    """pose * exp(d) (3rdparty/traits/sophus.h:24-26) for poses [..., 12] = (R row-major, t)."""
    R = pose[..., :9].reshape(pose.shape[:-1] + (3, 3))
    t = pose[..., 9:]
    Rd, td = se3_exp(d)
    Rn = R @ Rd
    tn = (R @ td[..., None])[..., 0] + t
    return np.concatenate([Rn.reshape(pose.shape[:-1] + (9,)), tn], -1)


def synth_se3_reproj(P: int, npts: int, dtype=np.float64, seed: int = 0x71940917):
    """BASELINE config C5: T* = exp(xi*), xi* ~ 0.3 U(-1,1)^6; points in a 4 x 4 x [4, 8] m frustum in the camera
    frame; pinhole f = 500, c = (320, 240); pixel noise 0.5 U(-1,1); T0 = T* exp(0.05 U(-1,1)^6).
    Returns (data [P, 8 + 5 npts] = [f cx cy 0.. | x y z u v per point], pose0 [P, 12], pose_star [P, 12])."""
    rng = np.random.default_rng(seed)
    ident = np.tile(np.concatenate([np.eye(3).ravel(), np.zeros(3)]), (P, 1))
    pstar = se3_plus(ident, 0.3 * rng.uniform(-1, 1, (P, 6)))
    p0 = se3_plus(pstar, 0.05 * rng.uniform(-1, 1, (P, 6)))
    R, t = pstar[:, :9].reshape(P, 3, 3), pstar[:, 9:]
    pc = np.stack([rng.uniform(-2, 2, (P, npts)), rng.uniform(-2, 2, (P, npts)), rng.uniform(4, 8, (P, npts))], -1)
    pw = np.einsum("pji,pnj->pni", R, pc - t[:, None, :])
    f, cx, cy = 500.0, 320.0, 240.0
    uv = np.stack([f * pc[..., 0] / pc[..., 2] + cx, f * pc[..., 1] / pc[..., 2] + cy], -1) + 0.5 * rng.uniform(-1, 1, (P, npts, 2))
    data = np.zeros((P, 8 + 5 * npts))
    data[:, 0], data[:, 1], data[:, 2] = f, cx, cy
    data[:, 8:] = np.concatenate([pw, uv], -1).reshape(P, -1)
    return data.astype(dtype), p0.astype(dtype), pstar.astype(dtype)


def _ba_geometry(rng, ncam: int, npts: int, noise_px: float):
    pts = rng.uniform(-1, 1, (npts, 3)) * np.array([1.5, 1.0, 1.0])
    ang = (np.arange(ncam) - (ncam - 1) / 2) * 0.25
    base = np.zeros((ncam, 12))
    base[:, 0], base[:, 2], base[:, 4], base[:, 6], base[:, 8] = np.cos(ang), np.sin(ang), 1.0, -np.sin(ang), np.cos(ang)
    base[:, 11] = 6.0                                   # cameras on a circle of radius 6 looking at the cloud
    poses = se3_plus(base, 0.05 * rng.uniform(-1, 1, (ncam, 6)))
    R = poses[:, :9].reshape(ncam, 3, 3)
    pc = np.einsum("cij,nj->cni", R, pts) + poses[:, None, 9:]
    uv = np.stack([500.0 * pc[..., 0] / pc[..., 2] + 320.0, 500.0 * pc[..., 1] / pc[..., 2] + 240.0], -1)
    return poses, pts, uv + noise_px * rng.uniform(-1, 1, uv.shape)


def synth_ba(P: int, ncam: int, npts: int, dtype=np.float64, seed: int = 0x71940917, noise_px: float = 0.5, pose_pert: float = 0.02,
             point_pert: float = 0.05):
    """Bundle-adjustment scenes in `BundleAdjustment`'s dense layout, every point seen by every camera.
    Returns (data [P, 8 + 3 C N], x0 [P, 12 C + 3 N], x_star)."""
    rng = np.random.default_rng(seed)
    data = np.zeros((P, 8 + 3 * ncam * npts))
    x0 = np.zeros((P, 12 * ncam + 3 * npts))
    xs = np.zeros_like(x0)
    for p in range(P):
        poses, pts, uv = _ba_geometry(rng, ncam, npts, noise_px)
        data[p, 0], data[p, 1], data[p, 2] = 500.0, 320.0, 240.0
        data[p, 8:8 + 2 * ncam * npts] = uv.ravel()
        data[p, 8 + 2 * ncam * npts:] = 1.0
        xs[p, :12 * ncam], xs[p, 12 * ncam:] = poses.ravel(), pts.ravel()
        x0[p, :12 * ncam] = se3_plus(poses, pose_pert * rng.uniform(-1, 1, (ncam, 6))).ravel()
        x0[p, 12 * ncam:] = (pts + point_pert * rng.uniform(-1, 1, pts.shape)).ravel()
    return data.astype(dtype), x0.astype(dtype), xs.astype(dtype)


def synth_ba_lists(P: int, ncam: int, npts: int, per_point: int, dtype=np.float64, seed: int = 0x71940917, noise_px: float = 0.5,
                   pose_pert: float = 0.02, point_pert: float = 0.05):
    """Scenes for `BundleAdjustmentLists`: every point observed by `per_point` cameras drawn at random.
    Returns (intr [P, 4], obs_cam [P, M] int32, obs_pt [P, M] int32, obs_uv [P, M, 2], x0 [P, 12 C + 3 N], x_star), M = npts * per_point,
    observations sorted by (point, camera)."""
    rng = np.random.default_rng(seed)
    M = npts * per_point
    intr = np.tile(np.array([500.0, 320.0, 240.0, 0.0]), (P, 1))
    oc, op, ouv = np.zeros((P, M), np.int32), np.zeros((P, M), np.int32), np.zeros((P, M, 2))
    x0 = np.zeros((P, 12 * ncam + 3 * npts))
    xs = np.zeros_like(x0)
    for p in range(P):
        poses, pts, uv = _ba_geometry(rng, ncam, npts, noise_px)
        cams = np.sort(np.argsort(rng.uniform(size=(npts, ncam)), axis=1)[:, :per_point], axis=1)     # [N, per_point], sorted
        op[p] = np.repeat(np.arange(npts, dtype=np.int32), per_point)
        oc[p] = cams.ravel().astype(np.int32)
        ouv[p] = uv[oc[p], op[p]]
        xs[p, :12 * ncam], xs[p, 12 * ncam:] = poses.ravel(), pts.ravel()
        x0[p, :12 * ncam] = se3_plus(poses, pose_pert * rng.uniform(-1, 1, (ncam, 6))).ravel()
        x0[p, 12 * ncam:] = (pts + point_pert * rng.uniform(-1, 1, pts.shape)).ravel()
    return intr.astype(dtype), oc, op, ouv.astype(dtype), x0.astype(dtype), xs.astype(dtype)
