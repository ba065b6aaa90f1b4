"""Replay of tests/golden/round6.npz (tests/golden/make_golden.py --only-round6) through the device: the fixtures SURVEY §8(c) lists
that the directory lacked until round 6 — GaussianPrior at the sizes of the reference's published table, DenseRow C2 / C3 in fp32,
SE3 reprojection / pose-prior samples — and a DenseRow batch at n = 20 through the run-time row models (the residual and its
Jacobian row supplied as text; the residual alone, differentiated on the device)."""
import os

import numpy as np
import pytest
import torch

from parity import check_trajectories, gpu_dict

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "round6.npz")


@pytest.mark.parametrize("n", [3, 6, 12, 33, 50])
def test_gaussian_prior_fixture(ta, n):
    g = np.load(GOLD)
    y, sigma, x0 = g[f"gp{n}_y"], g[f"gp{n}_sigma"], g[f"gp{n}_x0"]
    model = ta.GaussianPrior(torch.from_numpy(y).cuda(), torch.from_numpy(sigma).cuda())
    gg, H, c, nres = ta.accumulate(model, torch.from_numpy(x0).cuda())
    torch.cuda.synchronize()
    assert np.allclose(gg.cpu().numpy(), g[f"gp{n}_g"], rtol=1e-12) and np.allclose(c.cpu().numpy(), g[f"gp{n}_cost"], rtol=1e-12)
    Hd = H.cpu().numpy()
    for p in range(y.shape[0]):
        assert np.allclose(np.diag(Hd[p]), g[f"gp{n}_Hdiag"][p], rtol=1e-12) and np.abs(Hd[p] - np.diag(np.diag(Hd[p]))).max() == 0
    assert (nres.cpu().numpy() == 1).all()                       # a scalar return is ONE residual (cost.h:22)
    o = ta.Options.benchmark()
    o.hessian.save_last = True
    x = torch.from_numpy(x0.copy()).cuda()
    out = ta.Optimize(x, model, o, history=True)
    torch.cuda.synchronize()
    assert np.array_equal(out.stop_reason.cpu().numpy(), g[f"gp{n}_stop"]) and np.array_equal(out.num_iters.cpu().numpy(), g[f"gp{n}_iters"])
    assert np.allclose(x.cpu().numpy(), g[f"gp{n}_x"], rtol=1e-10, atol=1e-12)
    assert np.allclose(out.final_hessian.cpu().numpy().reshape(-1, n, n), g[f"gp{n}_final_H"], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("tag", ["c2f32_", "c3f32_"])
def test_dense_row_fp32_fixture(ta, oracle, tag):
    g = np.load(GOLD)
    n, m, P = int(g[tag + "n"]), int(g[tag + "m"]), int(g[tag + "P"])
    A, b, x0, _ = oracle.synth_dense_row(P, n, m, np.float32, seed=0x71940917)     # inputs only: regenerated from the seed
    assert np.isclose(A.astype(np.float64).sum(), g[tag + "A_sum"], rtol=1e-12)
    model = ta.DenseRow.from_arrays(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
    gg, H, c, _ = ta.accumulate(model, torch.from_numpy(x0).cuda())
    torch.cuda.synchronize()
    for dev, ref in ((gg, g[tag + "g"]), (H, g[tag + "H"]), (c, g[tag + "cost"])):     # fp32 rel 1e-4 (math.h:297-301)
        assert np.abs(dev.cpu().numpy() - ref).max() <= 1e-4 * np.abs(ref).max()
    opts = ta.Options.benchmark()
    x = torch.from_numpy(x0.copy()).cuda()
    out = ta.Optimize(x, model, opts, history=True)
    torch.cuda.synchronize()
    ref = {k: g[tag + k] for k in ("errs", "succ", "iters", "stop", "x", "fails", "deltas2")}
    ref["cost"] = g[tag + "final_cost"]
    st = check_trajectories(gpu_dict(out, x), ref, np.float32, opts.to_pod(), label="DenseRow fp32 fixture " + tag)
    assert st["full"] + st["ties"] == P


def test_se3_fixtures(ta):
    g = np.load(GOLD)
    model = ta.SE3Reproj(torch.from_numpy(g["rp_data"]).cuda(), 64)
    gg, H, c, _ = ta.accumulate(model, torch.from_numpy(g["rp_pose0"]).cuda())
    torch.cuda.synchronize()
    assert np.allclose(gg.cpu().numpy(), g["rp_g"], rtol=1e-9) and np.allclose(H.cpu().numpy(), g["rp_H"], rtol=1e-9)
    assert np.allclose(c.cpu().numpy(), g["rp_cost"], rtol=1e-10)
    prior = ta.SE3Prior(torch.from_numpy(g["pr_prior_inv"]).cuda())
    gp, Hp, cp, _ = ta.accumulate(prior, torch.from_numpy(g["pr_pose"]).cuda())
    torch.cuda.synchronize()
    assert np.allclose(gp.cpu().numpy(), g["pr_g"], rtol=1e-9, atol=1e-12) and np.allclose(cp.cpu().numpy(), g["pr_cost"], rtol=1e-10)
    # exp: a pose-prior solve started at the identity ends on the inverse of prior_inv — reached through pose <- pose * exp(delta)
    ident = np.tile(np.concatenate([np.eye(3).ravel(), np.zeros(3)]), (6, 1))
    x = torch.from_numpy(ident.copy()).cuda()
    out = ta.Optimize(x, prior, ta.Options())
    torch.cuda.synchronize()
    assert bool((out.stop_reason >= 0).all()) and float(out.final_cost.max()) < 1e-10       # tests/sophus.cpp:44


@pytest.mark.parametrize("kind", ["accumulate", "residual"])
def test_row_model_fixture(ta, kind):
    from test_gpu_row_models import ad_body, manual_body
    g = np.load(GOLD)
    A, b, x0 = g["rm_A"], g["rm_b"], g["rm_x0"]
    n = A.shape[2]
    body = manual_body(n, fast_sincos=False) if kind == "accumulate" else ad_body(n)
    fit = ta.JitResidual(body, n=n, item_scalars=n + 1, dtype=torch.float64, kind=kind)
    model = fit.bind(torch.from_numpy(np.ascontiguousarray(np.concatenate([A, b[..., None]], -1))).cuda())
    gg, H, c, _ = ta.accumulate(model, torch.from_numpy(x0).cuda())
    torch.cuda.synchronize()
    assert np.allclose(gg.cpu().numpy(), g["rm_g"], rtol=1e-10, atol=1e-10 * np.abs(g["rm_g"]).max())
    assert np.allclose(H.cpu().numpy(), g["rm_H"], rtol=1e-10, atol=1e-10 * np.abs(g["rm_H"]).max())
    assert np.allclose(c.cpu().numpy(), g["rm_cost"], rtol=1e-10)
    opts = ta.Options.benchmark()
    x = torch.from_numpy(x0.copy()).cuda()
    out = ta.Optimize(x, model, opts, history=True)
    torch.cuda.synchronize()
    ref = {k: g["rm_" + k] for k in ("errs", "succ", "iters", "stop", "x", "fails", "deltas2")}
    ref["cost"] = g["rm_final_cost"]
    st = check_trajectories(gpu_dict(out, x), ref, np.float64, opts.to_pod(), label=f"row model fixture ({kind})")
    assert st["full"] + st["ties"] == A.shape[0]
