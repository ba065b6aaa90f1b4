"""Replay of tests/golden/round2_f64.npz (tests/golden/make_golden.py --only-round2) through the HIP paths of round 2:
bundle adjustment with the points eliminated, DenseRow with a Huber loss on every residual, and the natural-layout n = 72
batch of the workgroup-per-problem kernel — against the FROZEN oracle results, without recomputing them."""
import os

import numpy as np
import pytest
import torch

from parity import check_trajectories, gpu_dict

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "round2_f64.npz")


def _ref(g, pre):
    it = g[pre + "iters"]
    return dict(errs=g[pre + "errs"], succ=g[pre + "succ"], iters=it, stop=g[pre + "stop"], x=g[pre + "x"],
                cost=g[pre + ("final_cost" if pre == "hub_" else "cost")], fails=np.zeros_like(it), deltas2=g[pre + "deltas2"])


def test_bundle_adjustment_fixture(ta):
    g = np.load(GOLD)
    C, N = int(g["ba_C"]), int(g["ba_N"])
    opts = ta.Options()
    model = ta.BundleAdjustment(torch.from_numpy(g["ba_data"]).cuda(), C, N)
    x = torch.from_numpy(g["ba_x0"].copy()).cuda()
    out = ta.Optimize(x, model, opts, history=True)
    torch.cuda.synchronize()
    ref = _ref(g, "ba_")
    ref["fails"] = out.num_failures.cpu().numpy()   # not stored in the fixture
    st = check_trajectories(gpu_dict(out, x), ref, np.float64, opts.to_pod(), tol=dict(x_tol=1e-5, cost_rtol=1e-8), label="BA fixture")
    assert st["full"] + st["ties"] == x.shape[0]


def test_dense_row_huber_fixture(ta):
    g = np.load(GOLD)
    th = float(np.sqrt(g["hub_th2"]))
    model = ta.DenseRow.from_arrays(torch.from_numpy(g["hub_A"]).cuda(), torch.from_numpy(g["hub_b"]).cuda()).with_loss("huber", th)
    gg, H, c = ta.accumulate(model, torch.from_numpy(g["hub_x0"]).cuda())[:3]
    assert np.allclose(gg.cpu().numpy(), g["hub_g"], rtol=1e-10, atol=1e-10)
    assert np.allclose(H.cpu().numpy(), g["hub_H"], rtol=1e-10, atol=1e-10)
    assert np.allclose(c.cpu().numpy(), g["hub_cost"], rtol=1e-12)
    opts = ta.Options()
    x = torch.from_numpy(g["hub_x0"].copy()).cuda()
    out = ta.Optimize(x, model, opts, history=True)
    torch.cuda.synchronize()
    ref = _ref(g, "hub_")
    ref["fails"] = out.num_failures.cpu().numpy()
    st = check_trajectories(gpu_dict(out, x), ref, np.float64, opts.to_pod(), label="Huber fixture")
    assert st["full"] + st["ties"] == x.shape[0]


def test_natural_layout_fixture(ta, oracle):
    g = np.load(GOLD)
    n, m, P = int(g["nat_n"]), int(g["nat_m"]), int(g["nat_P"])
    A, b, x0, _ = oracle.synth_dense_row(P, n, m, np.float64, seed=int(g["nat_seed"]))   # inputs only: regenerated from the seed
    assert np.isclose(A.sum(), g["nat_A_sum"], rtol=1e-12) and np.array_equal(x0, g["nat_x0"])
    opts = ta.Options.benchmark()
    model = ta.DenseRowNatural(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
    x = torch.from_numpy(x0.copy()).cuda()
    out = ta.Optimize(x, model, opts, history=True)
    torch.cuda.synchronize()
    ref = _ref(g, "nat_")
    ref["fails"] = out.num_failures.cpu().numpy()
    st = check_trajectories(gpu_dict(out, x), ref, np.float64, opts.to_pod(), label="n = 72 fixture")
    assert st["full"] + st["ties"] == P
    assert np.abs(x.cpu().numpy() - g["nat_xstar"]).max() < 2e-2
