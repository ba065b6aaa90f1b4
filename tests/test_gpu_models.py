"""GPU parity for the small device models: Sqrt2 (BASELINE config 1, tests/sqrt2.cpp) and GaussianPrior
(the reference's published dense benchmark, benchmarks/dense.cpp) against the oracle and the reference's
known answers."""
import os

import numpy as np
import pytest
import torch

from parity import check_trajectories, gpu_dict

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("dtype,tdt", [(np.float64, torch.float64), (np.float32, torch.float32)])
def test_sqrt2_known_answers(ta, oracle, dtype, tdt):
    """tests/sqrt2.cpp:106-112: x0 in {1, -0.3, 3.2}, max_iters 20, max_consec_failures 0 ->
    Succeeded && Converged && |x| == sqrt(2) +- 1e-5; trajectories equal to the oracle / golden file."""
    o = ta.Options()
    o.max_iters = 20
    o.max_consec_failures = 0
    x0 = np.array([1.0, -0.3, 3.2], dtype)
    ref = oracle.sqrt2_lm(x0, o.to_pod())
    gold = np.load(os.path.join(GOLD, f"sqrt2_{'f64' if dtype == np.float64 else 'f32'}.npz"))
    x = torch.from_numpy(x0.reshape(3, 1).copy()).cuda()
    out = ta.Optimize(x, ta.Sqrt2(3, tdt), o, history=True)
    torch.cuda.synchronize()
    xg = x.cpu().numpy().ravel()
    stop = out.stop_reason.cpu().numpy()
    assert (stop >= 0).all() and ((stop >= 1) & (stop < 5)).all()
    assert np.allclose(np.abs(xg), np.sqrt(2.0), atol=1e-5)
    assert np.array_equal(stop, ref["stop"]) and np.array_equal(out.num_iters.cpu().numpy(), ref["iters"])
    assert np.array_equal(stop, gold["stop"])
    # scalar arithmetic, no reductions: trajectories agree to the last bits (FMA contraction may differ)
    rt = 1e-12 if dtype == np.float64 else 1e-5
    assert np.allclose(xg, ref["x"], rtol=rt, atol=0)
    assert np.allclose(out.errs.cpu().numpy(), ref["errs"], rtol=1e3 * rt, atol=1e-300)
    assert np.allclose(out.deltas2.cpu().numpy(), ref["deltas2"], rtol=1e3 * rt, atol=1e-300)


def test_sqrt2_readme_trace(ta):
    """README.md:91-96: x = 1 -> 1.49995 -> 1.41667 -> 1.41422 -> 1.41421, |dx| 5.00e-1, 8.33e-2."""
    x = torch.ones(1, 1, dtype=torch.float64, device="cuda")
    out = ta.Optimize(x, ta.Sqrt2(1, torch.float64), ta.Options(), history=True)
    d2 = out.deltas2.cpu().numpy()[0]
    assert abs(np.sqrt(d2[0]) - 0.5) < 5e-4 and abs(np.sqrt(d2[1]) - 8.33e-2) < 5e-5
    assert abs(x.item() - np.sqrt(2)) < 1e-5


def test_sqrt2_without_ldlt(ta, oracle):
    """benchmarks/dense.cpp:28-51 scalar cases run with use_ldlt = false (gn.h:157-162 Dims == 1 branch)."""
    o = ta.Options.benchmark()
    o.hessian.use_ldlt = False
    rng = np.random.default_rng(3)
    x0 = rng.uniform(-1, 1, 64)
    ref = oracle.sqrt2_lm(x0, o.to_pod())
    x = torch.from_numpy(x0.reshape(-1, 1).copy()).cuda()
    out = ta.Optimize(x, ta.Sqrt2(64, torch.float64), o)
    torch.cuda.synchronize()
    assert np.array_equal(out.stop_reason.cpu().numpy(), ref["stop"])
    assert np.array_equal(out.num_iters.cpu().numpy(), ref["iters"])
    assert np.allclose(x.cpu().numpy().ravel(), ref["x"], rtol=1e-12, atol=0)


@pytest.mark.parametrize("dtype,tdt", [(np.float64, torch.float64), (np.float32, torch.float32)])
@pytest.mark.parametrize("n", [3, 6, 12, 33, 50])   # the published table's sizes (docs/benchmark-ceres-table.png)
def test_gaussian_prior_matches_oracle(ta, oracle, dtype, tdt, n):
    P = 24
    y, sigma, x0 = oracle.synth_gaussian_prior(P, n, dtype, seed=5)
    o = ta.Options.benchmark()
    o.hessian.save_last = True
    ref = oracle.gaussian_prior_lm(y, sigma, x0, o.to_pod(), history=True)
    model = ta.GaussianPrior(torch.from_numpy(y).cuda(), torch.from_numpy(sigma).cuda())
    x = torch.from_numpy(x0.copy()).cuda()
    out = ta.Optimize(x, model, o, history=True)
    torch.cuda.synchronize()
    xg = x.cpu().numpy()
    stop = out.stop_reason.cpu().numpy()
    assert (stop >= 0).all()
    tol = 1e-12 if dtype == np.float64 else 1e-5
    assert np.abs(xg - y).max() < (1e-9 if dtype == np.float64 else 1e-4)       # known answer: x -> y
    assert np.abs(xg - ref["x"]).max() < tol * 10
    # the whole trajectory against the oracle's.  This solve converges in one Gauss-Newton step, after which every
    # cost is a sum of round-off residues (~eps^2 of the first cost): the floor of THIS problem is absolute, so costs
    # below eps * (first cost) are clamped onto one floor value before the comparison (tests/parity.py).
    refd = dict(errs=ref["errs"].copy(), succ=ref["succ"], iters=ref["iters"], stop=ref["stop"], x=ref["x"],
                cost=ref["cost"].copy(), fails=ref["fails"], deltas2=ref["deltas2"])
    g = gpu_dict(out, x)
    scale = np.maximum(ref["errs"][:, :1], 1e-300)
    eps = 1e-15 if dtype == np.float64 else 1e-6
    for d in (g, refd):
        d["errs"] = np.where(d["errs"] / scale < eps, 0.0, d["errs"]) + eps * scale
        d["cost"] = np.where(d["cost"] / scale[:, 0] < eps, 0.0, d["cost"]) + eps * scale[:, 0]
    st = check_trajectories(g, refd, dtype, o.to_pod(), label=f"gaussian prior n={n}")
    assert st["full"] + st["ties"] == P, st
    # tests/cov.cpp:20-47: covariance from the final UNDAMPED Hessian recovers the prior stdevs
    Hf = out.final_hessian.cpu().numpy()
    idx = np.arange(n)
    assert np.allclose(np.sqrt(1.0 / Hf[:, idx, idx]), sigma, rtol=1e-7 if dtype == np.float64 else 1e-4)
    off = Hf.copy()
    off[:, idx, idx] = 0
    assert not off.any()
    assert (out.final_num_residuals.cpu().numpy() == 1).all()                 # scalar return => Cost(v, 1)


def test_gaussian_prior_accumulate_seam(ta, oracle):
    y, sigma, x0 = oracle.synth_gaussian_prior(5, 12, np.float64, seed=9)
    model = ta.GaussianPrior(torch.from_numpy(y).cuda(), torch.from_numpy(sigma).cuda())
    g, H, c, nres = ta.accumulate(model, torch.from_numpy(x0).cuda())
    res = (x0 - y) / sigma
    assert np.array_equal(g.cpu().numpy(), (1.0 / sigma) * res)               # same operation order: bit-identical
    Hn = H.cpu().numpy()
    assert np.array_equal(Hn[:, np.arange(12), np.arange(12)], (1.0 / sigma) * (1.0 / sigma))
    assert np.allclose(c.cpu().numpy(), (res * res).sum(1), rtol=1e-14)
    assert (nres.cpu().numpy() == 1).all()


def test_model_shape_errors(ta):
    y = torch.zeros(2, 3, dtype=torch.float64, device="cuda")
    model = ta.GaussianPrior(y, y + 1)
    with pytest.raises(ValueError):
        ta.Optimize(torch.zeros(2, 4, dtype=torch.float64, device="cuda"), model)
    model.m = 5                                                               # m != n -> invalid argument from the C-ABI
    with pytest.raises(ta.ToaError):
        ta.Optimize(torch.zeros(2, 3, dtype=torch.float64, device="cuda"), model)


def _ortho_err(poses):
    R = poses[:, :9].reshape(-1, 3, 3)
    return np.abs(np.einsum("pij,pkj->pik", R, R) - np.eye(3)).max()


@pytest.mark.parametrize("dtype,tdt", [(np.float64, torch.float64), (np.float32, torch.float32)])
def test_se3_reproj_accumulate_and_fd(ta, oracle, dtype, tdt):
    """(g, H, cost) vs the oracle, and g = J^T r vs a central finite difference over the RIGHT perturbation
    (pose * exp(+-eps e_a)), in the spirit of diff/gradient_check.h:96-98."""
    P, npts = 3, 333
    data, p0, _ = oracle.synth_se3_reproj(P, npts, dtype, seed=21)
    model = ta.SE3Reproj(torch.from_numpy(data).cuda(), npts)
    g, H, c, nres = ta.accumulate(model, torch.from_numpy(p0).cuda())
    g_ref, H_ref, c_ref = oracle.se3_reproj_accumulate(data, p0, npts)
    tol = 1e-10 if dtype == np.float64 else 2e-4
    scale = np.abs(g_ref).max()
    assert np.abs(g.cpu().numpy() - g_ref).max() < tol * scale
    assert np.abs(H.cpu().numpy() - H_ref).max() < tol * np.abs(H_ref).max()
    assert np.allclose(c.cpu().numpy(), c_ref, rtol=tol)
    assert (nres.cpu().numpy() == 2 * npts).all()
    if dtype == np.float64:
        eps = 1e-6
        for a in range(6):
            d = np.zeros((P, 6)); d[:, a] = eps
            cp = ta.accumulate(model, torch.from_numpy(oracle.se3_plus(p0, d)).cuda(), want_grad=False)[2].cpu().numpy()
            cm = ta.accumulate(model, torch.from_numpy(oracle.se3_plus(p0, -d)).cuda(), want_grad=False)[2].cpu().numpy()
            assert np.allclose(0.5 * (cp - cm) / (2 * eps), g.cpu().numpy()[:, a], rtol=1e-6)


@pytest.mark.parametrize("dtype,tdt,npts", [(np.float64, torch.float64, 25000), (np.float64, torch.float64, 100),
                                            (np.float32, torch.float32, 2000)])
def test_se3_reproj_lm(ta, oracle, dtype, tdt, npts):
    """BASELINE config 5 (25 000 points = 50 000 residuals, fp64) and smaller: planted pose recovered to the
    pixel-noise level, same StopReason / iterations as the oracle, poses stay on the manifold."""
    P = 2
    data, p0, pstar = oracle.synth_se3_reproj(P, npts, dtype, seed=4)
    o = ta.Options()
    ref = oracle.se3_reproj_lm(data, p0, npts, o.to_pod())
    model = ta.SE3Reproj(torch.from_numpy(data).cuda(), npts)
    x = torch.from_numpy(p0.copy()).cuda()
    out = ta.Optimize(x, model, o)
    torch.cuda.synchronize()
    xg = x.cpu().numpy()
    stop = out.stop_reason.cpu().numpy()
    if dtype == np.float64:
        assert (stop >= 1).all() and (stop < 5).all()                  # Succeeded && Converged (tests/sophus.cpp:42-43)
    else:
        assert (stop >= 0).all()   # fp32: the exit at the round-off floor may be kMaxConsecNoDecr (still a success)
    assert _ortho_err(xg) < (1e-12 if dtype == np.float64 else 1e-5)
    noise = 0.5 / 500.0 / np.sqrt(npts) * 50                            # generous: pixel noise / focal / sqrt(N)
    assert np.abs(xg - pstar).max() < max(noise, 2e-3 if dtype == np.float32 else 1e-4)
    if dtype == np.float64:
        assert np.abs(xg - ref["x"]).max() < 1e-9
        assert np.array_equal(stop, ref["stop"]) and np.array_equal(out.num_iters.cpu().numpy(), ref["iters"])
        assert np.allclose(out.final_cost.cpu().numpy(), ref["cost"], rtol=1e-10)
        assert np.allclose(out.final_hessian.cpu().numpy(), ref["H"], rtol=1e-9, atol=1e-6 * np.abs(ref["H"]).max())
    else:
        assert np.abs(xg - ref["x"]).max() < 5e-4


@pytest.mark.parametrize("dtype,tdt,npts,P", [(np.float64, torch.float64, 300, 700), (np.float32, torch.float32, 300, 700), (np.float64, torch.float64, 25000, 1)])
def test_se3_reproj_without_a_header_loss_runs_the_kernels_without_the_estimator_branch(ta, oracle, dtype, tdt, npts, P):
    """Round 6: a model whose data headers name no loss runs the kernels WITHOUT the M-estimator branch (toa_tuning::se3_reproj_header_l2, set by
    the host mirrors from the model: fp64 308 -> 216 registers).  Same arithmetic on the L2 path: the results are the bits of the full kernels — a
    batch (fused kernel) and BASELINE C5's single problem (row-split form) —; a model WITH a loss keeps the full kernels."""
    data, p0, _ = oracle.synth_se3_reproj(P, npts, dtype, seed=9)
    d = torch.from_numpy(data).cuda()
    model = ta.SE3Reproj(d, npts)
    assert model.header_l2
    o = ta.Options()
    ctx = ta.api.default_context()
    x = torch.from_numpy(p0.copy()).cuda()
    out = ta.Optimize(x, model, o, history=True)
    torch.cuda.synchronize()
    assert ctx.get_tuning()["se3_reproj_header_l2"] == 1
    model.header_l2 = False                       # the same bytes through the kernels with the branch
    x2 = torch.from_numpy(p0.copy()).cuda()
    out2 = ta.Optimize(x2, model, o, history=True)
    torch.cuda.synchronize()
    assert ctx.get_tuning()["se3_reproj_header_l2"] == 0
    assert torch.equal(x, x2) and torch.equal(out.num_iters, out2.num_iters) and torch.equal(out.errs, out2.errs) and torch.equal(out.final_cost, out2.final_cost)
    robust = ta.SE3Reproj(d, npts, loss="huber", th=3.0)
    assert not robust.header_l2
    x3 = torch.from_numpy(p0.copy()).cuda()
    out3 = ta.Optimize(x3, robust, o)
    torch.cuda.synchronize()
    assert ctx.get_tuning()["se3_reproj_header_l2"] == 0 and bool((out3.stop_reason > 0).all())
    byhand = d.clone(); byhand[:, 3] = 2.0; byhand[:, 4] = 9.0     # a header filled in by the caller is seen at construction
    assert not ta.SE3Reproj(byhand, npts).header_l2


@pytest.mark.parametrize("dtype,tdt", [(np.float64, torch.float64), (np.float32, torch.float32)])
def test_inv_cov_and_output_covariance(ta, oracle, dtype, tdt):
    """tinyopt::InvCov (math.h:41-57) + Output::Covariance (output.h:80-94), pinned like tests/cov.cpp:20-47:
    the covariance from the final undamped Hessian recovers the prior standard deviations."""
    rng = np.random.default_rng(1)
    for n in (1, 2, 7, 33, 50):
        J = rng.uniform(-1, 1, (5, 3 * n + 2, n))
        H = np.einsum("pij,pik->pjk", J, J).astype(dtype)
        C, ok = ta.inv_cov(torch.from_numpy(H).cuda())
        assert ok.cpu().numpy().all()
        I = np.einsum("pij,pjk->pik", H.astype(np.float64), C.cpu().numpy().astype(np.float64))
        assert np.abs(I - np.eye(n)).max() < (1e-9 if dtype == np.float64 else 5e-3)
    bad = np.stack([np.diag([1.0, -1.0, 2.0]), np.diag([1.0, 2.0, 3.0])]).astype(dtype)   # indefinite -> nullopt
    C, ok = ta.inv_cov(torch.from_numpy(bad).cuda())
    assert list(ok.cpu().numpy()) == [0, 1]
    # tests/cov.cpp: Gaussian prior, sigma = 4.2 -> sqrt(diag(Cov)) == sigma
    P, n = 6, 4
    y = rng.uniform(-10, 10, (P, n)).astype(dtype)
    sigma = np.full((P, n), 4.2, dtype)
    model = ta.GaussianPrior(torch.from_numpy(y).cuda(), torch.from_numpy(sigma).cuda())
    x = torch.zeros(P, n, dtype=tdt, device="cuda")
    out = ta.Optimize(x, model, ta.Options())
    assert out.Succeeded().all() and out.Converged().all()
    Cv, ok = out.Covariance()
    d = np.sqrt(np.diagonal(Cv.cpu().numpy(), axis1=1, axis2=2))
    assert ok.cpu().numpy().all() and np.abs(d - 4.2).max() < (1e-7 if dtype == np.float64 else 1e-4)


@pytest.mark.parametrize("dtype,tdt", [(np.float64, torch.float64), (np.float32, torch.float32)])
def test_maha_prior_general_covariance(ta, oracle, dtype, tdt):
    """tests/cov.cpp:91-146: a prior with the general covariance Cy = [[10,2],[2,4]] whitened by Lt = chol(Cy^-1).U;
    the solve lands on y and the covariance from the final Hessian equals Cy +-1e-5.  Plus random SPD covariances
    up to n = 50 against the oracle (MahaWhitenedInfoU, losses/mahalanobis.h:160-171)."""
    Cy = np.array([[[10.0, 2.0], [2.0, 4.0]]])
    y = np.array([[1.3, -0.7]], dtype)
    model = ta.MahaPrior.from_covariance(torch.from_numpy(y).cuda(), torch.from_numpy(Cy).cuda())
    x = torch.zeros(1, 2, dtype=tdt, device="cuda")
    out = ta.Optimize(x, model, ta.Options())
    torch.cuda.synchronize()
    assert bool(out.Succeeded().all()) and bool(out.Converged().all())            # REQUIRE(out.Succeeded/Converged)
    assert np.abs(x.cpu().numpy() - y).max() < (1e-8 if dtype == np.float64 else 1e-5)
    C, ok = out.Covariance()
    assert ok.cpu().numpy().all()
    assert np.abs(C.cpu().numpy() - Cy).max() < (1e-5 if dtype == np.float64 else 2e-3)   # REQUIRE((C - Cy) ~ 0, 1e-5)

    rng = np.random.default_rng(8)
    for n in (2, 6, 12, 33, 50):
        P = 6
        A = rng.uniform(-1, 1, (P, n + 3, n))
        cov = np.einsum("pij,pik->pjk", A, A) + 0.5 * np.eye(n)
        yy = rng.uniform(-2, 2, (P, n)).astype(dtype)
        data = oracle.maha_prior_data(yy, cov)
        x0 = np.zeros((P, n), dtype)
        o = ta.Options()
        ref = oracle.maha_prior_lm(data, x0, o.to_pod())
        m2 = ta.MahaPrior.from_covariance(torch.from_numpy(yy).cuda(), torch.from_numpy(cov).cuda())
        assert np.allclose(m2.packed.cpu().numpy(), data, rtol=1e-9 if dtype == np.float64 else 1e-5, atol=1e-12 if dtype == np.float64 else 1e-6)
        xg = torch.from_numpy(x0.copy()).cuda()
        out = ta.Optimize(xg, m2, o)
        torch.cuda.synchronize()
        assert (out.stop_reason.cpu().numpy() >= 0).all() and (ref["stop"] >= 0).all()
        tol = 1e-8 if dtype == np.float64 else 2e-3
        assert np.abs(xg.cpu().numpy() - ref["x"]).max() < tol
        assert np.abs(xg.cpu().numpy() - yy).max() < tol
        if dtype == np.float64:
            assert np.array_equal(out.stop_reason.cpu().numpy(), ref["stop"])
            assert np.array_equal(out.num_iters.cpu().numpy(), ref["iters"])
            assert np.allclose(out.final_hessian.cpu().numpy(), ref["H"], rtol=1e-9, atol=1e-12 * np.abs(ref["H"]).max())
            C, ok = out.Covariance()
            assert ok.cpu().numpy().all()
            assert np.allclose(C.cpu().numpy(), cov, rtol=1e-7, atol=1e-9 * np.abs(cov).max())


def test_se3_pose_prior_reference_test(ta, oracle):
    """tests/sophus.cpp:26-44: prior_inv = exp(random), pose = exp(random), residual log(prior_inv * x) differentiated by
    Jets over the right perturbation -> Succeeded && Converged && ||log(pose * prior_inv)|| < 1e-5.  The device
    Accumulate (Jet<T,6> through SE3 log) against the oracle's, then the whole solve against the oracle's trajectory."""
    rng = np.random.default_rng(44)
    P = 64
    ident = np.tile(np.concatenate([np.eye(3).ravel(), np.zeros(3)]), (P, 1))
    prior_inv = oracle.se3_plus(ident, 0.8 * rng.uniform(-1, 1, (P, 6)))
    pose0 = oracle.se3_plus(ident, 0.8 * rng.uniform(-1, 1, (P, 6)))
    pose0[0] = oracle.se3_plus(_inverse_pose(prior_inv[:1]), 1e-3 * rng.uniform(-1, 1, (1, 6)))[0]   # inside the small-angle branch
    model = ta.SE3Prior(torch.from_numpy(prior_inv).cuda())
    g, H, c, nres = ta.accumulate(model, torch.from_numpy(pose0).cuda())
    g_ref, H_ref, c_ref = oracle.se3_prior_accumulate(prior_inv, pose0)
    assert np.allclose(g.cpu().numpy(), g_ref, rtol=1e-10, atol=1e-12)
    assert np.allclose(H.cpu().numpy(), H_ref, rtol=1e-10, atol=1e-12)
    assert np.allclose(c.cpu().numpy(), c_ref, rtol=1e-12, atol=1e-300) and (nres.cpu().numpy() == 6).all()
    c0 = ta.accumulate(model, torch.from_numpy(pose0).cuda(), want_grad=False)[2]
    assert np.allclose(c0.cpu().numpy(), c_ref, rtol=1e-12, atol=1e-300)

    o = ta.Options()
    ref = oracle.se3_prior_lm(prior_inv, pose0, o.to_pod())
    x = torch.from_numpy(pose0.copy()).cuda()
    out = ta.Optimize(x, model, o)
    torch.cuda.synchronize()
    xg = x.cpu().numpy()
    assert bool(out.Succeeded().all()) and bool(out.Converged().all())          # REQUIRE(out.Succeeded / Converged)
    resid = np.linalg.norm(oracle.se3_log(oracle.se3_compose(xg, prior_inv)), axis=1)
    assert resid.max() < 1e-5                                                   # REQUIRE((pose * prior_inv).log().norm() ~ 0)
    assert _ortho_err(xg) < 1e-12
    assert np.array_equal(out.stop_reason.cpu().numpy(), ref["stop"])
    assert np.array_equal(out.num_iters.cpu().numpy(), ref["iters"])
    assert np.abs(xg - ref["x"]).max() < 1e-9


def _inverse_pose(p):
    R = p[:, :9].reshape(-1, 3, 3)
    Rt = np.transpose(R, (0, 2, 1))
    t = -np.einsum("pij,pj->pi", Rt, p[:, 9:])
    return np.concatenate([Rt.reshape(-1, 9), t], axis=1)
