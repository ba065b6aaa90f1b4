"""GPU parity for the robust norms: the reference's M-estimators
(include/tinyopt/losses/robust_norms.h:32-316) alone — against the closed forms and derivative checks of
tests/robust_norms.cpp:53-115 and against the oracle — and inside K1 for the SE3 reprojection model
(per-point re-weighting, inlier ratio in Cost)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

KINDS = ["truncated", "huber", "tukey", "arctan", "cauchy", "geman_mcclure", "blake_zisserman"]


def _expected(kind, n2, th2):
    """LOSS_WRAPPER expected_code of tests/robust_norms.cpp:53-59."""
    th, n = np.sqrt(th2), np.sqrt(n2)
    return {
        "truncated": np.where(n > th, th2, n2),
        "huber": np.where(n > th, 2.0 * th * n - th2, n2),
        "tukey": np.where(n > th, th2, th2 * (1.0 - (1.0 - n2 / th2) ** 3)),
        "arctan": th * np.arctan2(n2, th),
        "cauchy": th2 * np.log(1.0 + n2 / th2),
        "geman_mcclure": n2 / (n2 + th2),
        "blake_zisserman": -np.log(np.exp(-n2) + np.exp(-th2)),
    }[kind]


@pytest.mark.parametrize("kind", KINDS)
def test_reference_known_answers(ta, kind):
    """th = 1.3; n2 = 0.5 ("Scalar"), 0.3 ("Scalar Inlier"), 2.3^2 ("Scalar Outlier"); the vector cases use
    ||(.1,-.2,-.3,.4)||^2 with th = 1.3 and th = 0.03.  Loss == closed form and scale == d loss / d n2, margin 1e-5."""
    xs = 0.1 ** 2 + 0.2 ** 2 + 0.3 ** 2 + 0.4 ** 2
    for th, n2s in ((1.3, [0.5, 0.3, 2.3 * 2.3, xs]), (0.03, [xs])):
        th2 = th * th
        n2 = np.array(n2s)
        h = 1e-6
        stack = torch.from_numpy(np.concatenate([n2, n2 + h, n2 - h])).cuda()
        l, s = ta.robust_norm(kind, stack, th2)
        l, s = l.cpu().numpy(), s.cpu().numpy()
        k = len(n2)
        assert np.abs(l[:k] - _expected(kind, n2, th2)).max() < 1e-5
        fd = (l[k:2 * k] - l[2 * k:]) / (2 * h)
        assert np.abs(s[:k] - fd).max() < 1e-5


@pytest.mark.parametrize("dtype,tdt", [(np.float64, torch.float64), (np.float32, torch.float32)])
@pytest.mark.parametrize("kind", KINDS + ["l2"])
def test_robust_norm_vs_oracle(ta, oracle, kind, dtype, tdt):
    rng = np.random.default_rng(5)
    n2 = np.concatenate([10.0 ** rng.uniform(-6, 1.5, 4000), [0.0, 1.69, 1.69 * (1 + 1e-7), 1.69 * (1 - 1e-7)]]).astype(dtype)
    th2 = 1.69
    l_ref, s_ref = oracle.robust_norm(kind, n2, th2)
    l, s = ta.robust_norm(kind, torch.from_numpy(n2).cuda(), th2)
    rt = 1e-13 if dtype == np.float64 else 3e-6
    # exp/log/atan2 of the device library vs libm: a few ulp; blake_zisserman's -log(...) cancels near 0
    at = (1e-15 if dtype == np.float64 else 1e-7) * (10 if kind == "blake_zisserman" else 1)
    assert np.allclose(l.cpu().numpy(), l_ref, rtol=rt, atol=at)
    assert np.allclose(s.cpu().numpy(), s_ref, rtol=rt, atol=at)


@pytest.mark.parametrize("kind", ["huber", "cauchy", "tukey", "truncated"])
def test_se3_robust_accumulate(ta, oracle, kind):
    """(g, H, cost) of the re-weighted Accumulate vs the oracle; gradient vs a finite difference of the ROBUST cost
    for the smooth estimators (the reference's own check of a loss Jacobian, tests/robust_norms.cpp:79-82)."""
    P, npts = 3, 400
    data, p0, _ = oracle.synth_se3_reproj(P, npts, np.float64, seed=33)
    data, _ = oracle.se3_add_outliers(data, npts, 0.15, seed=3)
    th = 6.0 if kind in ("huber", "cauchy") else 40.0   # redescending estimators: keep some points inside the threshold
    dref = oracle.se3_set_loss(data, kind, th * th)
    model = ta.SE3Reproj(torch.from_numpy(data).cuda(), npts, loss=kind, th=th)
    assert np.array_equal(model.packed.cpu().numpy(), dref)
    g, H, c, nres = ta.accumulate(model, torch.from_numpy(p0).cuda())
    g_ref, H_ref, c_ref = oracle.se3_reproj_accumulate(dref, p0, npts)
    assert np.abs(g_ref).max() > 0
    assert np.abs(g.cpu().numpy() - g_ref).max() < 1e-10 * np.abs(g_ref).max()
    assert np.abs(H.cpu().numpy() - H_ref).max() < 1e-10 * np.abs(H_ref).max()
    assert np.allclose(c.cpu().numpy(), c_ref, rtol=1e-11)
    c0 = ta.accumulate(model, torch.from_numpy(p0).cuda(), want_grad=False)[2]
    assert np.allclose(c0.cpu().numpy(), c_ref, rtol=1e-11)      # cost-only pass == cost of the full pass
    if kind in ("huber", "cauchy"):
        eps = 1e-6
        for a in range(6):
            d = np.zeros((P, 6)); d[:, a] = eps
            cp = ta.accumulate(model, torch.from_numpy(oracle.se3_plus(p0, d)).cuda(), want_grad=False)[2].cpu().numpy()
            cm = ta.accumulate(model, torch.from_numpy(oracle.se3_plus(p0, -d)).cuda(), want_grad=False)[2].cpu().numpy()
            assert np.allclose(0.5 * (cp - cm) / (2 * eps), g.cpu().numpy()[:, a], rtol=2e-5, atol=1e-4)


@pytest.mark.parametrize("kind", ["huber", "cauchy", "geman_mcclure"])
@pytest.mark.parametrize("P,npts,splits", [(4, 600, None), (1, 25000, None), (2, 3000, 7)])
def test_se3_robust_lm(ta, oracle, kind, P, npts, splits):
    """BA-style solve with 10 % gross outliers: same trajectory as the oracle (StopReason, iterations, cost, pose,
    inlier ratio) on the fused path, the automatic row-split path (P = 1, C5 size) and an explicit split; and the
    point of a robust norm: the planted pose is recovered far better than by plain least squares."""
    data, p0, pstar = oracle.synth_se3_reproj(P, npts, np.float64, seed=11)
    data, mask = oracle.se3_add_outliers(data, npts, 0.10, seed=5)
    th = 3.0 if kind != "geman_mcclure" else 5.0
    dref = oracle.se3_set_loss(data, kind, th * th)
    o = ta.Options()
    ref = oracle.se3_reproj_lm(dref, p0, npts, o.to_pod())
    model = ta.SE3Reproj(torch.from_numpy(data).cuda(), npts, loss=kind, th=th)
    x = torch.from_numpy(p0.copy()).cuda()
    out = ta.Optimize(x, model, o, splits=splits)
    torch.cuda.synchronize()
    xg = x.cpu().numpy()
    stop = out.stop_reason.cpu().numpy()
    assert (stop >= 0).all()
    assert np.array_equal(stop, ref["stop"]) and np.array_equal(out.num_iters.cpu().numpy(), ref["iters"])
    assert np.allclose(out.final_cost.cpu().numpy(), ref["cost"], rtol=1e-9)
    assert np.abs(xg - ref["x"]).max() < 1e-8
    assert np.allclose(out.final_inlier_ratio.cpu().numpy(), ref["inlier_ratio"], atol=1.5 / (2 * npts))
    # inliers found ~ points that were not corrupted
    assert np.abs(out.final_inlier_ratio.cpu().numpy() - (1 - mask.mean(axis=1))).max() < 0.05
    # plain L2 on the same contaminated data is pulled away by the outliers
    x2 = torch.from_numpy(p0.copy()).cuda()
    out2 = ta.Optimize(x2, ta.SE3Reproj(torch.from_numpy(data).cuda(), npts), o, splits=splits)
    torch.cuda.synchronize()
    assert (out2.final_inlier_ratio.cpu().numpy() == 1.0).all()
    err_robust = np.abs(xg - pstar).max()
    err_l2 = np.abs(x2.cpu().numpy() - pstar).max()
    assert err_robust < 0.2 * err_l2, (err_robust, err_l2)


def test_robust_golden_fixture(ta):
    """Replay tests/golden/robust_f64.npz (tests/golden/make_golden.py) through the HIP path."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "robust_f64.npz"))
    n2 = torch.from_numpy(g["n2"]).cuda()
    for kind in KINDS:
        l, s = ta.robust_norm(kind, n2, float(g["th2"]))
        assert np.allclose(l.cpu().numpy(), g[f"{kind}_loss"], rtol=1e-13, atol=1e-15)
        assert np.allclose(s.cpu().numpy(), g[f"{kind}_scale"], rtol=1e-13, atol=1e-15)
    x = torch.from_numpy(g["se3_p0"].copy()).cuda()
    out = ta.Optimize(x, ta.SE3Reproj(torch.from_numpy(g["se3_data"]).cuda(), 200), ta.Options())
    torch.cuda.synchronize()
    assert np.array_equal(out.stop_reason.cpu().numpy(), g["se3_stop"])
    assert np.array_equal(out.num_iters.cpu().numpy(), g["se3_iters"])
    assert np.allclose(out.final_cost.cpu().numpy(), g["se3_cost"], rtol=1e-9)
    assert np.abs(x.cpu().numpy() - g["se3_x"]).max() < 1e-8
    assert np.allclose(out.final_inlier_ratio.cpu().numpy(), g["se3_inlier_ratio"], atol=1e-6)


def test_robust_norm_argument_errors(ta):
    n2 = torch.ones(4, dtype=torch.float64, device="cuda")
    ctx = ta.api.default_context(0)
    with pytest.raises(ta.ToaError):
        ta.api.check(ctx.lib.toa_robust_norm(ctx.h, 42, 1, 4, n2.data_ptr(), 1.0, n2.data_ptr(), n2.data_ptr()))
