"""Device-side forward-mode AD (csrc/jet.hpp = ceres::Jet on the GPU, JetModel = OptimizeWithAutoDiff,
include/tinyopt/diff/optimize_autodiff.h:21-169): the user writes only r(x); J comes from dual numbers."""
import numpy as np
import pytest
import torch

from parity import check_trajectories, gpu_dict

pytestmark = pytest.mark.gpu


def _circle_obs(P, n, dtype, seed=0):
    """tests/circle.cpp:20-30: n points on a circle of radius 2 centred at (2, 7) + 1e-5 noise."""
    rng = np.random.default_rng(seed)
    ang = np.linspace(0, 2 * np.pi, n)[None, :] + rng.uniform(0, 1, (P, 1))
    obs = np.stack([2 + 2 * np.cos(ang), 7 + 2 * np.sin(ang)], -1) + 1e-5 * rng.uniform(-1, 1, (P, n, 2))
    return obs.astype(dtype)


@pytest.mark.parametrize("dtype,tdt", [(np.float64, torch.float64), (np.float32, torch.float32)])
def test_circle_fit_reference_known_answer(ta, oracle, dtype, tdt):
    """tests/circle.cpp:32-68: x0 = (0, 0, 1), lm.damping_init = 10 -> (2, 7, 2) +- 1e-5, Succeeded."""
    P, npts = 7, 10
    obs = _circle_obs(P, npts, dtype)
    x0 = np.tile(np.array([0, 0, 1], dtype), (P, 1))
    o = ta.Options()
    o.lm.damping_init = 1e1
    ref = oracle.circle_fit_lm(obs, x0, o.to_pod())
    x = torch.from_numpy(x0.copy()).cuda()
    out = ta.Optimize(x, ta.CircleFit(torch.from_numpy(obs).cuda()), o)
    torch.cuda.synchronize()
    xg = x.cpu().numpy()
    stop = out.stop_reason.cpu().numpy()
    assert (stop >= 0).all()
    tol = 1e-5 if dtype == np.float64 else 2e-4
    assert np.abs(xg[:, 0] - 2).max() < tol and np.abs(xg[:, 1] - 7).max() < tol and np.abs(np.abs(xg[:, 2]) - 2).max() < tol
    if dtype == np.float64:
        assert np.abs(xg - ref["x"]).max() < 1e-8
        assert np.array_equal(stop, ref["stop"]) and np.array_equal(out.num_iters.cpu().numpy(), ref["iters"])
        assert np.allclose(out.final_cost.cpu().numpy(), ref["cost"], rtol=1e-6, atol=1e-18)


def test_dense_row_ad_equals_analytic_path(ta, oracle):
    """The same residual through dual numbers (DenseRowAD6: r(x) only) and through the hand-derived MFMA path
    (DenseRow): g, H, cost and the whole LM trajectory agree to rounding — diff.cpp:89-111 'AD Jacobians'."""
    P, n, m = 9, 6, 130
    A, b, x0, xs = oracle.synth_dense_row(P, n, m, np.float64, seed=12)
    Ad, bd, xd = torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda(), torch.from_numpy(x0).cuda()
    ad = ta.DenseRowAD6(Ad, bd)
    an = ta.DenseRow.from_arrays(Ad, bd)
    g1, H1, c1, n1 = ta.accumulate(ad, xd)
    g2, H2, c2, n2 = ta.accumulate(an, xd)
    g_ref, H_ref, c_ref, _ = oracle.dense_row_accumulate(A, b, x0)
    for got, want in ((g1, g_ref), (H1, H_ref)):
        assert np.abs(got.cpu().numpy() - want).max() < 1e-10 * np.abs(want).max()
    assert np.allclose(c1.cpu().numpy(), c_ref, rtol=1e-12) and (n1.cpu().numpy() == m).all()
    assert np.allclose(g1.cpu().numpy(), g2.cpu().numpy(), rtol=0, atol=1e-10 * np.abs(g_ref).max())
    _, _, c1e, _ = ta.accumulate(ad, xd, want_grad=False)                      # functor evaluated on plain T
    assert np.allclose(c1e.cpu().numpy(), c_ref, rtol=1e-12)
    o = ta.Options()
    ref = oracle.dense_row_lm(A, b, x0, o.to_pod(), history=True)
    x = xd.clone()
    out = ta.Optimize(x, ad, o, history=True)
    torch.cuda.synchronize()
    assert np.abs(x.cpu().numpy() - ref["x"]).max() < 1e-8
    assert np.abs(x.cpu().numpy() - xs).max() < 5e-3
    refd = dict(errs=ref["errs"], succ=ref["succ"], iters=ref["iters"], stop=ref["stop"], x=ref["x"], cost=ref["cost"],
                fails=ref["fails"], deltas2=ref["deltas2"])
    st = check_trajectories(gpu_dict(out, x), refd, np.float64, o.to_pod(), label="DenseRowAD6")
    assert st["full"] + st["ties"] == P and all(j >= 2 for j in st["tie_iters"]), st


def test_ad_model_fp32_and_errors(ta, oracle):
    A, b, x0, xs = oracle.synth_dense_row(4, 6, 300, np.float32, seed=3)
    ad = ta.DenseRowAD6(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
    x = torch.from_numpy(x0.copy()).cuda()
    out = ta.Optimize(x, ad, ta.Options.benchmark())
    assert (out.stop_reason >= 0).all() and np.abs(x.cpu().numpy() - xs).max() < 5e-3
    with pytest.raises(ValueError):
        ta.Optimize(torch.zeros(4, 5, dtype=torch.float32, device="cuda"), ad)
