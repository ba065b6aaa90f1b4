"""Bundle adjustment with the points eliminated (`toa_ba_run`): C SE3 cameras x N points against the
oracle, which solves the SAME problem the way the reference would — tinyopt::Optimize on the full dense (6C + 3N)^2
Hessian with the dense LDL^T (oracle/ba.hpp; math.h:232-240).  The Schur step is mathematically the dense step, so the
whole trajectory (cost, accept / reject, StopReason, iterations) must agree up to proven ties (tests/parity.py)."""
import numpy as np
import pytest
import torch

from parity import check_trajectories, gpu_dict

pytestmark = pytest.mark.gpu


def _ortho_err(x, ncam):
    R = x[:, :12 * ncam].reshape(x.shape[0], ncam, 12)[..., :9].reshape(-1, 3, 3)
    return np.abs(np.einsum("pij,pkj->pik", R, R) - np.eye(3)).max()


@pytest.mark.parametrize("dtype,tdt", [(np.float64, torch.float64), (np.float32, torch.float32)])
@pytest.mark.parametrize("ncam,npts,invisible", [(2, 200, 0.0), (8, 200, 0.0), (3, 50, 0.2), (10, 40, 0.1), (5, 30, 0.0), (7, 33, 0.0)])
def test_ba_matches_dense_oracle(ta, oracle, dtype, tdt, ncam, npts, invisible):
    P = 3
    data, x0, xs = oracle.synth_ba(P, ncam, npts, dtype, seed=7 + ncam, invisible=invisible)
    for opts in (ta.Options(), ta.Options.benchmark()):
        ref = oracle.ba_lm(data, x0, ncam, npts, opts.to_pod())
        model = ta.BundleAdjustment(torch.from_numpy(data).cuda(), ncam, npts)
        x = torch.from_numpy(x0.copy()).cuda()
        out = ta.Optimize(x, model, opts, history=True)
        torch.cuda.synchronize()
        xg = x.cpu().numpy()
        assert (out.stop_reason.cpu().numpy() >= 0).all() and (ref["stop"] >= 0).all()
        assert _ortho_err(xg, ncam) < (1e-12 if dtype == np.float64 else 1e-5)          # poses stay on the manifold
        assert np.array_equal(out.final_num_residuals.cpu().numpy(), ref["nres"])         # 2 per visible observation
        refd = dict(errs=ref["errs"], succ=ref["succ"], iters=ref["iters"], stop=ref["stop"], x=ref["x"], cost=ref["cost"],
                    fails=ref["fails"], deltas2=ref["deltas2"])
        # the gauge of a bundle adjustment is free (7 directions only the damping fixes): x is compared loosely, the
        # trajectory of costs / decisions tightly
        tol = dict(x_tol=1e-5, cost_rtol=1e-8) if dtype == np.float64 else dict(x_tol=5e-2, cost_rtol=5e-3, err_rtol=2e-3, floor_rtol=2e-3)
        st = check_trajectories(gpu_dict(out, x), refd, dtype, opts.to_pod(), tol=tol, label=f"BA {ncam}x{npts}")
        assert st["full"] + st["ties"] == P
        if dtype == np.float64:
            k = int(min(out.num_iters.min().item(), ref["iters"].min()))
            assert np.allclose(out.deltas2.cpu().numpy()[:, :k], ref["deltas2"][:, :k], rtol=1e-6)   # the STEP equals the dense step
        # the planted scene: the reprojection cost comes down to the pixel-noise level (0.5 px uniform: 1/12 per residual)
        nres = out.final_num_residuals.cpu().numpy()
        assert (out.final_cost.cpu().numpy() < 0.25 / 3 * nres * 1.3).all()



@pytest.mark.parametrize("ncam,npts", [(2, 5000), (8, 5000)])
def test_ba_large_scene_properties(ta, oracle, ncam, npts):
    """N = 5000 points (15 000 - 15 048 unknowns: the dense oracle would need a 1.8 GB Hessian): size-independent
    properties — planted scene recovered to the noise level, accepted costs never increase, the first step equals a
    float64 numpy Schur step computed from the device's own gradient / Hessian blocks ... and batch independence."""
    P = 2
    data, x0, xs = oracle.synth_ba(P, ncam, npts, np.float64, seed=11)
    model = ta.BundleAdjustment(torch.from_numpy(data).cuda(), ncam, npts)
    opts = ta.Options()
    x = torch.from_numpy(x0.copy()).cuda()
    out = ta.Optimize(x, model, opts, history=True)
    torch.cuda.synchronize()
    stop, iters = out.stop_reason.cpu().numpy(), out.num_iters.cpu().numpy()
    assert ((stop >= 1) & (stop < 5)).all()                                               # Succeeded && Converged
    nres = out.final_num_residuals.cpu().numpy()
    fc = out.final_cost.cpu().numpy()
    assert (nres == 2 * ncam * npts).all()
    # residuals are U(-0.5, 0.5) px: variance 1/12 each; the fit absorbs one of them per unknown (minus the 7 gauge directions)
    dof = nres - (6 * ncam + 3 * npts) + 7
    assert (fc < dof / 12.0 * 1.15).all() and (fc > dof / 12.0 * 0.85).all(), (fc, dof / 12.0)
    errs, succ = out.errs.cpu().numpy(), out.successes.cpu().numpy().astype(bool)
    for p in range(P):
        acc = [errs[p, i] for i in range(iters[p]) if succ[p, i] or i == 0]
        assert all(b <= a for a, b in zip(acc, acc[1:]))
    assert _ortho_err(x.cpu().numpy(), ncam) < 1e-12
    # a scene solved alone is bit-identical to its row in the batch
    m1 = ta.BundleAdjustment(torch.from_numpy(data[1:2]).cuda(), ncam, npts)
    x1 = torch.from_numpy(x0[1:2].copy()).cuda()
    o1 = ta.Optimize(x1, m1, opts)
    torch.cuda.synchronize()
    assert torch.equal(x1[0], x[1]) and int(o1.num_iters[0]) == iters[1] and float(o1.final_cost[0]) == fc[1]
    # the same scene with its first 200 points only, against the dense oracle, lands on the same cameras (to the noise)
    if ncam == 2:
        sub = np.concatenate([data[:, :8], data[:, 8:8 + 2 * ncam * npts].reshape(P, ncam, npts, 2)[:, :, :200].reshape(P, -1),
                              data[:, 8 + 2 * ncam * npts:].reshape(P, ncam, npts)[:, :, :200].reshape(P, -1)], 1)
        xsub = np.concatenate([x0[:, :12 * ncam], x0[:, 12 * ncam:12 * ncam + 600]], 1)
        ref = oracle.ba_lm(sub, xsub, ncam, 200, opts.to_pod())
        assert (ref["stop"] >= 1).all()


def test_ba_argument_errors(ta, oracle):
    data, x0, _ = oracle.synth_ba(1, 2, 10, np.float64)
    model = ta.BundleAdjustment(torch.from_numpy(data).cuda(), 2, 10)
    with pytest.raises(ValueError):
        ta.Optimize(torch.zeros(1, 5, dtype=torch.float64, device="cuda"), model)
    big = ta.BundleAdjustment(torch.zeros(1, 8 + 3 * 11 * 4, dtype=torch.float64, device="cuda"), 11, 4)
    with pytest.raises(ta.ToaError):
        ta.Optimize(torch.zeros(1, 12 * 11 + 12, dtype=torch.float64, device="cuda"), big)     # more than 10 cameras
    o = ta.Options()
    o.max_duration_ms = 1.0
    with pytest.raises(ValueError):
        ta.Optimize(torch.from_numpy(x0).cuda(), model, o)


def test_ba_single_camera_is_exactly_fittable(ta, oracle):
    """One camera: every point is free along its viewing ray, so the reprojection error can be driven to zero — the
    damped solve must get there (cost down by 20 orders of magnitude), like the dense oracle; relative comparisons of
    such costs are meaningless, so only the verdicts are compared."""
    data, x0, _ = oracle.synth_ba(2, 1, 64, np.float64, seed=8)
    o = ta.Options()
    ref = oracle.ba_lm(data, x0, 1, 64, o.to_pod())
    x = torch.from_numpy(x0.copy()).cuda()
    out = ta.Optimize(x, ta.BundleAdjustment(torch.from_numpy(data).cuda(), 1, 64), o, history=True)
    torch.cuda.synchronize()
    assert (out.stop_reason.cpu().numpy() >= 1).all() and (ref["stop"] >= 1).all()
    assert (out.final_cost.cpu().numpy() < 1e-12 * out.errs.cpu().numpy()[:, 0]).all() and (ref["cost"] < 1e-12 * ref["errs"][:, 0]).all()
    assert np.abs(out.num_iters.cpu().numpy().astype(int) - ref["iters"]).max() <= 2
