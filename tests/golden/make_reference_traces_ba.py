#!/usr/bin/env python3
"""SECOND READING, part 3 (round 5; VERDICT r04 "weak" #1 / "next" #6): a small bundle adjustment the reference's way.

oracle/ba.hpp ("the loss applied the reference's way", hand-derived pose / point Jacobians, the dense (6C + 3N)^2 system) and both
device bundle-adjustment forms had ONE reading.  This is a second, independent one in plain Python:

    the parameter object      x = (C SE3 poses stored as R row-major | t, N 3-D points); PlusEq = pose * exp(delta) for poses
                              (3rdparty/traits/sophus.h:13-27 — Sophus' SE3::exp, restated from its published formulas), + for points
                              (traits.h:184-190); tangent order: cameras first (upsilon, omega each), then points
    the residual              pinhole reprojection  pi(R p + t) - uv,  pi(X, Y, Z) = (f X / Z + cx, f Y / Z + cy)
    its derivatives           NOT hand-derived: forward-mode dual numbers through x (+) delta at delta = 0, the way the reference
                              differentiates (diff/optimize_autodiff.h:48-77: Jets seeded on the tangent) — a derivation that shares
                              nothing with oracle/ba.hpp's closed-form [R | -R [p]x] blocks
    the M-estimator           per OBSERVATION: n2 = |r|^2 of its two residuals through rho (make_reference_traces_robust.py's restatement
                              of robust_norms.h), cost += l, J^T J and J^T r scaled by s, both residuals inliers when n2 <= th2
    the linear algebra        the full dense (6C + 3N)^2 Hessian and the pivoted LDL^T of SolveLDLT (math.h:232-240) — part 1's
    the loop                  part 1's optimize() (optimizer.h / lm.h / gn.h), with the manifold's PlusEq

Nothing under oracle/ or tinyopt_amd/ is imported.  Emits tests/golden/reference_traces_ba.json: per case the scene (data in the
layout of oracle/ba.hpp / toa_ba_run: [f cx cy 0 0 0 0 0 | uv: C x N x 2 | vis: C x N]), the start, the options, and the trajectory
(cost, |dx|^2, accept flag per pass; final cost, StopReason, iterations, inlier ratio).  x itself is NOT compared tightly by the
consumers: the gauge of a bundle adjustment is free (only the damping fixes 7 directions).  tests/test_cpu_oracle.py holds
oracle/ba.hpp to these traces; tests/test_gpu_traces.py both device forms.

Run:  python tests/golden/make_reference_traces_ba.py     (numpy only; a minute of pure-Python linear algebra)"""
import json
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_reference_traces as base            # noqa: E402
import make_reference_traces_robust as robust   # noqa: E402


# ---- dual numbers: value + first derivatives w.r.t. K tangent directions (what ceres::Jet<T, N> is) -----------------------------
class Dual:
    __slots__ = ("a", "v")

    def __init__(self, a, v):
        self.a, self.v = a, v

    @staticmethod
    def lift(x, K):
        return x if isinstance(x, Dual) else Dual(x, [0.0] * K)

    def _bin(self, o):
        return o if isinstance(o, Dual) else Dual(o, [0.0] * len(self.v))

    def __add__(self, o):
        o = self._bin(o)
        return Dual(self.a + o.a, [p + q for p, q in zip(self.v, o.v)])
    __radd__ = __add__

    def __sub__(self, o):
        o = self._bin(o)
        return Dual(self.a - o.a, [p - q for p, q in zip(self.v, o.v)])

    def __rsub__(self, o):
        return self._bin(o) - self

    def __neg__(self):
        return Dual(-self.a, [-p for p in self.v])

    def __mul__(self, o):
        o = self._bin(o)
        return Dual(self.a * o.a, [self.a * q + p * o.a for p, q in zip(self.v, o.v)])
    __rmul__ = __mul__

    def __truediv__(self, o):
        o = self._bin(o)
        inv = 1.0 / o.a
        q = self.a * inv
        return Dual(q, [(p - q * r) * inv for p, r in zip(self.v, o.v)])

    def __rtruediv__(self, o):
        return self._bin(o) / self


def dsin(x):
    return Dual(math.sin(x.a), [math.cos(x.a) * p for p in x.v]) if isinstance(x, Dual) else math.sin(x)


def dcos(x):
    return Dual(math.cos(x.a), [-math.sin(x.a) * p for p in x.v]) if isinstance(x, Dual) else math.cos(x)


def dsqrt(x):
    if isinstance(x, Dual):
        r = math.sqrt(x.a)
        return Dual(r, [0.5 * p / r for p in x.v])
    return math.sqrt(x)


def val(x):
    return x.a if isinstance(x, Dual) else x


# ---- Sophus SE3::exp (tangent = (upsilon, omega)), over floats or duals ------------------------------------------------------------
def se3_exp(d):
    """(R [3][3], t [3]) of exp(d).  theta^2 below 1e-10 (in particular AT 0, where the derivative is taken): the Taylor branches
    A = 1 - t2/6, B = 1/2 - t2/24, C = 1/6 - t2/120 (no sqrt of zero: the dual number stays finite)."""
    ux, uy, uz, wx, wy, wz = d
    t2 = wx * wx + wy * wy + wz * wz
    if val(t2) < 1e-10:
        A = 1.0 - t2 / 6.0
        B = 0.5 - t2 / 24.0
        Cc = 1.0 / 6.0 - t2 / 120.0
    else:
        th = dsqrt(t2)
        A = dsin(th) / th
        B = (1.0 - dcos(th)) / t2
        Cc = (th - dsin(th)) / (t2 * th)
    R = [[1.0 - B * (wy * wy + wz * wz), B * wx * wy - A * wz, A * wy + B * wx * wz],
         [A * wz + B * wx * wy, 1.0 - B * (wx * wx + wz * wz), B * wy * wz - A * wx],
         [B * wx * wz - A * wy, A * wx + B * wy * wz, 1.0 - B * (wx * wx + wy * wy)]]
    u = [ux, uy, uz]
    w = [wx, wy, wz]
    c1 = [w[1] * u[2] - w[2] * u[1], w[2] * u[0] - w[0] * u[2], w[0] * u[1] - w[1] * u[0]]        # omega x upsilon
    c2 = [w[1] * c1[2] - w[2] * c1[1], w[2] * c1[0] - w[0] * c1[2], w[0] * c1[1] - w[1] * c1[0]]  # omega x (omega x upsilon)
    t = [u[i] + B * c1[i] + Cc * c2[i] for i in range(3)]                                          # V upsilon
    return R, t


def pose_times_exp(P, d):
    """pose * exp(d): rotation R_p R_d, translation R_p t_d + t_p.  P: 12 floats (R row-major, t)."""
    Rd, td = se3_exp(d)
    out = [None] * 12
    for i in range(3):
        for j in range(3):
            out[3 * i + j] = P[3 * i] * Rd[0][j] + P[3 * i + 1] * Rd[1][j] + P[3 * i + 2] * Rd[2][j]
        out[9 + i] = P[3 * i] * td[0] + P[3 * i + 1] * td[1] + P[3 * i + 2] * td[2] + P[9 + i]
    return out


def reproject(P, q, f, cx, cy):
    X = P[0] * q[0] + P[1] * q[1] + P[2] * q[2] + P[9]
    Y = P[3] * q[0] + P[4] * q[1] + P[5] * q[2] + P[10]
    Z = P[6] * q[0] + P[7] * q[1] + P[8] * q[2] + P[11]
    return f * X / Z + cx, f * Y / Z + cy


def make_cost(C, N, f, cx, cy, uv, vis, kind, th2, tally):
    n = 6 * C + 3 * N
    rho = robust.RHO[kind] if kind else (lambda n2, t2: (n2, 1.0))
    K = 9
    seeds = [Dual(0.0, [1.0 if k == a else 0.0 for k in range(K)]) for a in range(K)]

    def fn(x, want):
        cost, nres, ninl = 0.0, 0, 0
        g = H = None
        if want:
            g = [0.0] * n
            H = [[0.0] * n for _ in range(n)]
        for c in range(C):
            P = x[12 * c:12 * c + 12]
            for j in range(N):
                if vis[c][j] == 0.0:
                    continue
                q = x[12 * C + 3 * j:12 * C + 3 * j + 3]
                if want:   # the residual on x (+) delta with duals seeded on delta (6 pose + 3 point directions) at delta = 0
                    Pd = pose_times_exp(P, seeds[:6])
                    qd = [q[k] + seeds[6 + k] for k in range(3)]
                    pu, pv = reproject(Pd, qd, f, cx, cy)
                    r = [pu.a - uv[c][j][0], pv.a - uv[c][j][1]]
                    J = [pu.v, pv.v]
                else:
                    pu, pv = reproject(P, q, f, cx, cy)
                    r = [pu - uv[c][j][0], pv - uv[c][j][1]]
                n2 = r[0] * r[0] + r[1] * r[1]
                l, s = rho(n2, th2)
                cost += l
                nres += 2
                if (not kind) or n2 <= th2:
                    ninl += 2
                if want:
                    idx = [6 * c + k for k in range(6)] + [6 * C + 3 * j + k for k in range(3)]
                    for row in range(2):
                        for a_ in range(9):
                            sJ = s * J[row][a_]
                            g[idx[a_]] += sJ * r[row]
                            for b_ in range(9):
                                H[idx[b_]][idx[a_]] += sJ * J[row][b_]
        tally.append((ninl, nres))
        return cost, g, H, nres
    return fn


def make_plus(C, N):
    def plus(x, dx, sign):
        out = list(x)
        for c in range(C):
            out[12 * c:12 * c + 12] = pose_times_exp(x[12 * c:12 * c + 12], [sign * v for v in dx[6 * c:6 * c + 6]])
        for k in range(3 * N):
            out[12 * C + k] = x[12 * C + k] + sign * dx[6 * C + k]
        return out
    return plus


def scene(seed, C, N, invisible, outliers, pose_pert, point_pert):
    """Cameras on an arc at distance 6 looking at a cloud of points around the origin, f = 500, c = (320, 240), 0.5 px of uniform
    noise, a few gross outliers among the observations; the start = the planted scene perturbed on the manifold."""
    rng = np.random.default_rng(seed)
    f, cx, cy = 500.0, 320.0, 240.0
    pts = (rng.uniform(-1, 1, (N, 3)) * np.array([1.5, 1.0, 1.0])).tolist()
    poses = []
    for c in range(C):
        ang = (c - (C - 1) / 2) * 0.25
        Pb = [math.cos(ang), 0.0, math.sin(ang), 0.0, 1.0, 0.0, -math.sin(ang), 0.0, math.cos(ang), 0.0, 0.0, 6.0]
        poses.append(pose_times_exp(Pb, (0.05 * rng.uniform(-1, 1, 6)).tolist()))
    uv = [[None] * N for _ in range(C)]
    vis = [[1.0] * N for _ in range(C)]
    for c in range(C):
        for j in range(N):
            u, v = reproject(poses[c], pts[j], f, cx, cy)
            uv[c][j] = [round(u + 0.5 * rng.uniform(-1, 1), 6), round(v + 0.5 * rng.uniform(-1, 1), 6)]
            if c >= 2 and rng.uniform() < invisible:
                vis[c][j] = 0.0
    for _ in range(outliers):
        c, j = int(rng.integers(C)), int(rng.integers(N))
        uv[c][j][0] += round(float(rng.choice([-1.0, 1.0]) * rng.uniform(15, 40)), 6)
    x0 = []
    for c in range(C):
        x0 += pose_times_exp(poses[c], (pose_pert * rng.uniform(-1, 1, 6)).tolist())
    for j in range(N):
        x0 += [pts[j][k] + point_pert * rng.uniform(-1, 1) for k in range(3)]
    x0 = [round(float(v), 9) for v in x0]
    data = [f, cx, cy, 0.0, 0.0, 0.0, 0.0, 0.0] + [v for c in range(C) for j in range(N) for v in uv[c][j]] + [v for c in range(C) for v in vis[c]]
    return f, cx, cy, uv, vis, x0, data


OPT = dict(min_rerr_dec=1e-6, max_iters=30)
CASES = [
    # C, N, invisible, outliers, loss, th (pixels), pose / point perturbation, options
    (2, 8, 0.0, 0, None, 0.0, 0.02, 0.05, OPT),
    (3, 10, 0.25, 0, None, 0.0, 0.03, 0.08, OPT),
    (3, 9, 0.0, 0, None, 0.0, 0.02, 0.05, dict(OPT, damping_init=1e-6)),      # (GaussNewton proper fails at once: the gauge makes H singular)
    (4, 10, 0.3, 0, None, 0.0, 0.02, 0.05, dict(OPT, damping_init=1e-2)),
    (3, 10, 0.0, 3, "huber", 3.0, 0.02, 0.05, OPT),
    (3, 10, 0.2, 3, "cauchy", 3.0, 0.02, 0.05, OPT),
    (2, 9, 0.0, 2, "tukey", 6.0, 0.02, 0.04, OPT),
    (4, 8, 0.2, 3, "geman_mcclure", 4.0, 0.015, 0.04, OPT),
    (3, 8, 0.0, 2, "arctan", 3.0, 0.02, 0.05, dict(OPT, use_step_quality_approx=True)),
    (3, 9, 0.0, 2, "truncated", 5.0, 0.015, 0.04, OPT),
]


def main():
    cases = []
    for k, (C, N, inv, outl, kind, th, pp, qp, kw) in enumerate(CASES):
        f, cx, cy, uv, vis, x0, data = scene(9000 + k, C, N, inv, outl, pp, qp)
        th2 = th * th
        opt = base.Options(**kw)
        tally = []
        name = f"ba_{k}"
        base.FUNCS[name] = make_cost(C, N, f, cx, cy, uv, vis, kind, th2, tally)
        plus = make_plus(C, N)
        out = base.optimize(name, x0, opt, plus=plus, ndim=6 * C + 3 * N)
        evals = list(tally)
        robust_ok = out["min_margin"] >= 1e-9 and len(evals) == len(out["errs"])
        for eps in (1e-13, -1e-13):
            ctr = [0]

            def perturb(c_, eps=eps, ctr=ctr):
                ctr[0] += 1
                return c_ * (1.0 + eps * (1 if ctr[0] % 2 else -1))
            if base.decisions(base.optimize(name, x0, opt, perturb=perturb, plus=plus, ndim=6 * C + 3 * N)) != base.decisions(out):
                robust_ok = False
        nrej = sum(1 for s in out["successes"][1:] if not s)
        print(f"C={C} N={N} loss={kind} stop={out['stop']} iters={out['num_iters']} rejected={nrej} margin={out['min_margin']:.2e} robust={robust_ok} "
              f"cost {(out['errs'] or [float('nan')])[0]:.4g} -> {out['final_cost']:.4g}")
        if not robust_ok:
            continue
        acc = [i for i, s in enumerate(out["successes"]) if s or i == 0]
        ninl, nres = evals[acc[-1]]
        o = {kk: (v if not isinstance(v, tuple) else list(v)) for kk, v in vars(opt).items()}
        cases.append(dict(function="bundle_adjustment", ncam=C, npts=N, loss=kind, th2=th2, data=data, x0=x0, options=o, dtype="float64",
                          comment=f"C={C} N={N} invisible={inv} outliers={outl} loss={kind}",
                          errs=out["errs"], deltas2=out["deltas2"], successes=[int(s) for s in out["successes"]],
                          stop_reason=out["stop"], num_iters=out["num_iters"], num_failures=out["num_failures"],
                          num_consec_failures=out["num_consec"], final_cost=out["final_cost"], final_num_residuals=nres,
                          final_inlier_ratio=ninl / nres, x=out["x"]))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_traces_ba.json")
    with open(path, "w") as fjs:
        json.dump(dict(generator="tests/golden/make_reference_traces_ba.py (independent Python restatement, see its docstring)", cases=cases), fjs, indent=1)
    print(f"wrote {len(cases)} cases to {path}")


if __name__ == "__main__":
    main()
