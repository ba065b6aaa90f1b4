#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (oracle/lm_oracle.hpp).

The reference holds no golden vectors for the LM path (unseeded Eigen::Random
inputs, tolerance-to-analytic assertions) and cannot be built or imported here (C++20 + Eigen 3.4 +
Catch2, none in the image), so these fixtures are produced by the oracle AFTER it has been pinned to
the reference's known answers (oracle/pin_reference_tests.cpp).  They freeze (a) the oracle against
regressions and (b) small inputs/outputs the GPU parity tests replay without recomputing.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402
from tinyopt_amd.api import Options  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def sqrt2_traces():
    """sqrt2 from x0 in {1, -0.3, 3.2} (tests/sqrt2.cpp:106-112), fp64 AD-scalar semantics, default options
    with the test's max_iters=20 / max_consec_failures=0 (tests/sqrt2.cpp:22-28)."""
    o = Options()
    o.max_iters = 20
    o.max_consec_failures = 0
    for dt, tag in ((np.float64, "f64"), (np.float32, "f32")):
        x0 = np.array([1.0, -0.3, 3.2], dt)
        r = pyoracle.sqrt2_lm(x0, o.to_pod())
        np.savez(os.path.join(OUT, f"sqrt2_{tag}.npz"), x0=x0, x=r["x"], stop=r["stop"], iters=r["iters"], cost=r["cost"],
                 errs=r["errs"], deltas2=r["deltas2"], succ=r["succ"])


def dense_row_cases():
    """(g, H, cost) triples and full LM results for small DenseRow batches at the BASELINE shapes."""
    for tag, dt, n, m, P in (("c2_f64", np.float64, 6, 1000, 2), ("c3_f64", np.float64, 12, 500, 4),
                             ("c4_f32", np.float32, 50, 2000, 2), ("c4_f64", np.float64, 50, 2000, 1)):
        A, b, x0, xs = pyoracle.synth_dense_row(P, n, m, dt, seed=0x71940917)
        g, H, c, nres = pyoracle.dense_row_accumulate(A, b, x0)
        o = Options.benchmark()
        o.hessian.save_last = True
        r = pyoracle.dense_row_lm(A, b, x0, o.to_pod(), history=True)
        # inputs are regenerated from the seed by the tests (keeps the fixture small); a checksum pins them
        np.savez_compressed(os.path.join(OUT, f"dense_row_{tag}.npz"), n=n, m=m, P=P, seed=0x71940917,
                            A_sum=np.float64(A.astype(np.float64).sum()), b_sum=np.float64(b.astype(np.float64).sum()),
                            x0=x0, xstar=xs, g=g, H=H, cost=c, nres=nres, x=r["x"], stop=r["stop"], iters=r["iters"],
                            final_cost=r["cost"], final_H=r["H"], errs=r["errs"], deltas2=r["deltas2"], succ=r["succ"])


def ldlt_cases():
    rng = np.random.default_rng(20260928)
    Hs, gs = [], []
    for n in (1, 2, 3, 6, 12):
        J = rng.uniform(-1, 1, (3 * n + 1, n))
        Hs.append(np.pad(J.T @ J, ((0, 12 - n), (0, 12 - n))))
        gs.append(np.pad(rng.uniform(-1, 1, n), (0, 12 - n)))
    np.savez(os.path.join(OUT, "ldlt_spd.npz"), H=np.stack(Hs), g=np.stack(gs), n=np.array([1, 2, 3, 6, 12]),
             dx=np.stack([np.pad(pyoracle.solve_damped(H[None, :n, :n].copy(), g[None, :n].copy(), 1.0001)[0][0], (0, 12 - n))
                          for H, g, n in zip(Hs, gs, (1, 2, 3, 6, 12))]))


def robust_cases():
    """(loss, scale) of every M-estimator (losses/robust_norms.h) on a fixed grid of squared norms, th = 1.3 as in
    tests/robust_norms.cpp:66, plus a robust SE3 solve with planted outliers (Huber, 3 px)."""
    n2 = np.concatenate([np.array([0.0, 0.3, 0.5, 1.69, 2.3 * 2.3]), 10.0 ** np.linspace(-5, 1.5, 60)])
    th2 = 1.3 * 1.3
    out = {"n2": n2, "th2": np.float64(th2)}
    for kind in ("truncated", "huber", "tukey", "arctan", "cauchy", "geman_mcclure", "blake_zisserman"):
        loss, scale = pyoracle.robust_norm(kind, n2, th2)
        out[f"{kind}_loss"], out[f"{kind}_scale"] = loss, scale
    P, npts = 2, 200
    data, p0, pstar = pyoracle.synth_se3_reproj(P, npts, np.float64, seed=11)
    data, mask = pyoracle.se3_add_outliers(data, npts, 0.10, seed=5)
    data = pyoracle.se3_set_loss(data, "huber", 9.0)
    r = pyoracle.se3_reproj_lm(data, p0, npts, Options().to_pod())
    out.update(se3_data=data, se3_p0=p0, se3_pstar=pstar, se3_x=r["x"], se3_stop=r["stop"], se3_iters=r["iters"],
               se3_cost=r["cost"], se3_inlier_ratio=r["inlier_ratio"])
    np.savez_compressed(os.path.join(OUT, "robust_f64.npz"), **out)


def round2_cases():
    """Round-2 paths: bundle adjustment solved the reference's dense way (oracle/ba.hpp), a DenseRow batch with a Huber loss
    on every residual, and a natural-layout n = 72 batch (the workgroup-per-problem kernel's range).  Inputs are stored
    (small) or regenerated from their seed with a checksum."""
    out = {}
    C, N, P = 3, 20, 2
    data, x0, xs = pyoracle.synth_ba(P, C, N, np.float64, seed=77, invisible=0.2)
    r = pyoracle.ba_lm(data, x0, C, N, Options().to_pod(), history=True)
    out.update(ba_C=C, ba_N=N, ba_data=data, ba_x0=x0, ba_x=r["x"], ba_stop=r["stop"], ba_iters=r["iters"], ba_cost=r["cost"],
               ba_errs=r["errs"], ba_succ=r["succ"], ba_deltas2=r["deltas2"])
    A, b, x0d, xsd = pyoracle.synth_dense_row(3, 6, 120, np.float64, seed=21)
    b[:, ::10] += 3.0                                    # planted outliers
    th2 = 0.25
    g, H, c, nres, inl = pyoracle.dense_row_accumulate(A, b, x0d, loss="huber", th2=th2)
    rl = pyoracle.dense_row_lm(A, b, x0d, Options().to_pod(), history=True, loss="huber", th2=th2)
    out.update(hub_A=A, hub_b=b, hub_x0=x0d, hub_th2=np.float64(th2), hub_g=g, hub_H=H, hub_cost=c, hub_inl=inl, hub_x=rl["x"], hub_stop=rl["stop"],
               hub_iters=rl["iters"], hub_final_cost=rl["cost"], hub_errs=rl["errs"], hub_succ=rl["succ"], hub_deltas2=rl["deltas2"])
    n, m, Pn = 72, 150, 2
    An, bn, x0n, xsn = pyoracle.synth_dense_row(Pn, n, m, np.float64, seed=5)
    rn = pyoracle.dense_row_lm(An, bn, x0n, Options.benchmark().to_pod(), history=True)
    out.update(nat_n=n, nat_m=m, nat_P=Pn, nat_seed=5, nat_A_sum=np.float64(An.sum()), nat_x0=x0n, nat_xstar=xsn, nat_x=rn["x"],
               nat_stop=rn["stop"], nat_iters=rn["iters"], nat_cost=rn["cost"], nat_errs=rn["errs"], nat_succ=rn["succ"],
               nat_deltas2=rn["deltas2"])
    np.savez_compressed(os.path.join(OUT, "round2_f64.npz"), **out)


def round4_cases():
    """Round-3 / round-4 device paths that had been compared with the oracle LIVE only (VERDICT r03 "weak" #1): a bundle
    adjustment with visibility lists (16 cameras), the same with a Cauchy loss, the one-workgroup blocked Cholesky range of the
    natural-layout family (n = 200 and n = 384), and the circle fit that tests/test_gpu_jit.py supplies as source text."""
    out = {}
    C, N = 16, 48
    data, x0, _ = pyoracle.synth_ba(2, C, N, np.float64, seed=404, invisible=0.6)
    r = pyoracle.ba_lm(data, x0, C, N, Options().to_pod(), history=True)
    out.update(bl_C=C, bl_N=N, bl_data=data, bl_x0=x0, bl_x=r["x"], bl_stop=r["stop"], bl_iters=r["iters"], bl_cost=r["cost"],
               bl_errs=r["errs"], bl_succ=r["succ"], bl_deltas2=r["deltas2"], bl_fails=r["fails"])
    rng = np.random.default_rng(404)
    dr = data.copy()
    uv = dr[:, 8:8 + 2 * C * N].reshape(2, C, N, 2)
    uv += (rng.random((2, C, N, 1)) < 0.1) * rng.uniform(25.0, 40.0, (2, C, N, 2))
    th = 4.0
    rr = pyoracle.ba_lm(dr, x0, C, N, Options().to_pod(), history=True, loss="cauchy", th2=th * th)
    out.update(blr_data=dr, blr_th=np.float64(th), blr_x=rr["x"], blr_stop=rr["stop"], blr_iters=rr["iters"], blr_cost=rr["cost"],
               blr_errs=rr["errs"], blr_succ=rr["succ"], blr_deltas2=rr["deltas2"], blr_fails=rr["fails"], blr_inl=rr["inlier_ratio"])
    for tag, n, m, P, seed in (("n200_", 200, 420, 2, 11), ("n384_", 384, 800, 1, 12)):
        A, b, x0n, xsn = pyoracle.synth_dense_row(P, n, m, np.float64, seed=seed)
        rn = pyoracle.dense_row_lm(A, b, x0n, Options.benchmark().to_pod(), history=True)
        out.update({tag + "n": n, tag + "m": m, tag + "P": P, tag + "seed": seed, tag + "A_sum": np.float64(A.sum()), tag + "xstar": xsn,
                    tag + "x": rn["x"], tag + "stop": rn["stop"], tag + "iters": rn["iters"], tag + "cost": rn["cost"], tag + "errs": rn["errs"],
                    tag + "succ": rn["succ"], tag + "deltas2": rn["deltas2"], tag + "fails": rn["fails"]})
    P, npts = 5, 10
    rng = np.random.default_rng(3)
    ang = np.linspace(0, 2 * np.pi, npts)[None, :] + rng.uniform(0, 1, (P, 1))
    obs = np.stack([2 + 2 * np.cos(ang), 7 + 2 * np.sin(ang)], -1) + 1e-5 * rng.uniform(-1, 1, (P, npts, 2))
    xc = np.tile(np.array([0.0, 0.0, 1.0]), (P, 1))
    o = Options()
    o.lm.damping_init = 1e1                      # tests/circle.cpp:58
    rc = pyoracle.circle_fit_lm(obs, xc, o.to_pod())
    out.update(cf_obs=obs, cf_x0=xc, cf_x=rc["x"], cf_stop=rc["stop"], cf_iters=rc["iters"], cf_cost=rc["cost"])
    np.savez_compressed(os.path.join(OUT, "round4_f64.npz"), **out)


def round6_cases():
    """The fixtures SURVEY §8(c) lists that the directory still lacked (VERDICT r05 "missing" #5): GaussianPrior (g, H, cost)
    triples and LM results at the sizes of the reference's published table (benchmarks/dense.cpp: n in {3, 6, 12, 33, 50}, m = n),
    DenseRow at the C2 / C3 shapes in fp32, SE3 exp / log / reprojection-Jacobian samples, and a DenseRow batch at n = 20 for the
    run-time row models of round 6 (the residual and its Jacobian row supplied as text)."""
    out = {}
    for n in (3, 6, 12, 33, 50):
        y, sigma, x0 = pyoracle.synth_gaussian_prior(3, n, np.float64, seed=600 + n)
        r = (x0 - y) / sigma                                      # benchmarks/dense.cpp:57-66: res = (x - y) / sigma
        g = r / sigma                                             # grad = J * res, J = diag(1 / sigma)
        hd = 1.0 / (sigma * sigma)                                # H.diagonal() = sigma^-2
        c = (r * r).sum(1)                                        # returns res.squaredNorm() (a scalar: Cost(v, 1), cost.h:22)
        o = Options.benchmark()
        o.hessian.save_last = True
        lm = pyoracle.gaussian_prior_lm(y, sigma, x0, o.to_pod(), history=True)
        out.update({f"gp{n}_y": y, f"gp{n}_sigma": sigma, f"gp{n}_x0": x0, f"gp{n}_g": g, f"gp{n}_Hdiag": hd, f"gp{n}_cost": c,
                    f"gp{n}_x": lm["x"], f"gp{n}_stop": lm["stop"], f"gp{n}_iters": lm["iters"], f"gp{n}_final_cost": lm["cost"],
                    f"gp{n}_final_H": lm["H"], f"gp{n}_errs": lm["errs"]})
    for tag, n, m, P in (("c2f32_", 6, 1000, 2), ("c3f32_", 12, 500, 4)):
        A, b, x0, xs = pyoracle.synth_dense_row(P, n, m, np.float32, seed=0x71940917)
        g, H, c, nres = pyoracle.dense_row_accumulate(A, b, x0)
        rl = pyoracle.dense_row_lm(A, b, x0, Options.benchmark().to_pod(), history=True)
        out.update({tag + "n": n, tag + "m": m, tag + "P": P, tag + "A_sum": np.float64(A.astype(np.float64).sum()), tag + "x0": x0,
                    tag + "g": g, tag + "H": H, tag + "cost": c, tag + "x": rl["x"], tag + "stop": rl["stop"], tag + "iters": rl["iters"],
                    tag + "final_cost": rl["cost"], tag + "errs": rl["errs"], tag + "succ": rl["succ"], tag + "deltas2": rl["deltas2"],
                    tag + "fails": rl["fails"]})
    # SE3: exp (pose * exp(delta), sophus.h:24-26), log, and the reprojection (g, H, cost) of SURVEY §8(d) C5's residual at the start pose
    rng = np.random.default_rng(606)
    ident = np.tile(np.concatenate([np.eye(3).ravel(), np.zeros(3)]), (6, 1))
    delta = np.concatenate([0.4 * rng.uniform(-1, 1, (5, 6)), np.zeros((1, 6))])            # (the last one: the identity, theta = 0)
    delta[4, 3:] *= 1e-9                                                                  # a rotation below the small-angle threshold
    poses = pyoracle.se3_plus(ident, delta)
    out.update(se3_delta=delta, se3_exp=poses, se3_log=pyoracle.se3_log(poses))
    poses2 = pyoracle.se3_plus(poses, 0.1 * rng.uniform(-1, 1, (6, 6)))
    out.update(se3_exp2=poses2, se3_log2=pyoracle.se3_log(poses2))
    data, p0, pstar = pyoracle.synth_se3_reproj(3, 64, np.float64, seed=66)
    gs, Hs, cs = pyoracle.se3_reproj_accumulate(data, p0, 64)
    out.update(rp_data=data, rp_pose0=p0, rp_g=gs, rp_H=Hs, rp_cost=cs)
    gp, Hp, cp = pyoracle.se3_prior_accumulate(poses2, poses)                                # residual log(prior_inv * x) (tests/sophus.cpp:26-44)
    out.update(pr_prior_inv=poses2, pr_pose=poses, pr_g=gp, pr_H=Hp, pr_cost=cp)
    # DenseRow at n = 20, two problems: inputs stored (small), for the row models
    A, b, x0, xs = pyoracle.synth_dense_row(2, 20, 96, np.float64, seed=620)
    g, H, c, _ = pyoracle.dense_row_accumulate(A, b, x0)
    rl = pyoracle.dense_row_lm(A, b, x0, Options.benchmark().to_pod(), history=True)
    out.update(rm_A=A, rm_b=b, rm_x0=x0, rm_g=g, rm_H=H, rm_cost=c, rm_x=rl["x"], rm_stop=rl["stop"], rm_iters=rl["iters"],
               rm_final_cost=rl["cost"], rm_errs=rl["errs"], rm_succ=rl["succ"], rm_deltas2=rl["deltas2"], rm_fails=rl["fails"])
    np.savez_compressed(os.path.join(OUT, "round6.npz"), **out)


if __name__ == "__main__":
    if "--only-round6" in sys.argv:
        round6_cases()
        sys.exit(0)
    if "--only-round4" in sys.argv:
        round4_cases()
        sys.exit(0)
    if "--only-robust" in sys.argv:
        robust_cases()
        sys.exit(0)
    if "--only-round2" in sys.argv:
        round2_cases()
        sys.exit(0)
    sqrt2_traces()
    dense_row_cases()
    ldlt_cases()
    robust_cases()
    round2_cases()
    round4_cases()
    round6_cases()
    print("golden fixtures written to", OUT)
