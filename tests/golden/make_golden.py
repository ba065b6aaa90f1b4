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


if __name__ == "__main__":
    if "--only-robust" in sys.argv:
        robust_cases()
        sys.exit(0)
    sqrt2_traces()
    dense_row_cases()
    ldlt_cases()
    robust_cases()
    print("golden fixtures written to", OUT)
