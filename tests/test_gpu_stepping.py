"""The reference's class / stepping form on the device (`lm::Optimizer<H_t> optimizer(options)`;
`optimizer.Step(x, acc, out)`; `optimizer(x, f, max_iters)` — include/tinyopt/optimizers/optimizer.h:199,331-539):
stepping to completion must reproduce `Optimize`, one loop pass per call, x updated in place at every step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _models(ta, oracle):
    A, b, x0, _ = oracle.synth_dense_row(24, 12, 500, np.float64, seed=3)
    yield "dense_row_f64", ta.DenseRow.from_arrays(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda()), x0, ta.Options.benchmark()
    A, b, x0, _ = oracle.synth_dense_row(16, 50, 300, np.float32, seed=4)
    yield "dense_row_f32", ta.DenseRow.from_arrays(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda()), x0, ta.Options.benchmark()
    rng = np.random.default_rng(2)
    starts = np.array([-1.2, 1.0])[None, :] + rng.uniform(-0.3, 0.3, (64, 2))
    o = ta.Options(); o.max_iters = 200; o.min_rerr_dec = 0.0; o.max_consec_failures = 20
    yield "rosenbrock", ta.TestFn("rosenbrock", 64), starts, o                     # rejected steps, rollbacks
    data, p0, _ = oracle.synth_se3_reproj(3, 400, np.float64, seed=4)
    yield "se3_reproj", ta.SE3Reproj(torch.from_numpy(data).cuda(), 400), p0, ta.Options()
    y = rng.uniform(-3, 3, (8, 6)); sg = rng.uniform(0.5, 1.5, (8, 6))
    yield "gaussian_prior", ta.GaussianPrior(torch.from_numpy(y).cuda(), torch.from_numpy(sg).cuda()), np.zeros((8, 6)), ta.Options()


def test_stepping_reproduces_optimize(ta, oracle):
    for name, model, x0, o in _models(ta, oracle):
        x_ref = torch.from_numpy(np.array(x0, copy=True)).cuda()
        ref = ta.Optimize(x_ref, model, o, history=True)
        x = torch.from_numpy(np.array(x0, copy=True)).cuda()
        opt = ta.Optimizer(x, model, o, history=True)
        steps, active_prev = 0, x.shape[0] + 1
        while True:
            active = opt.Step()
            steps += 1
            assert active <= active_prev, name                                   # problems only ever finish
            active_prev = active
            # every running problem has made exactly `steps` loop passes
            running = opt.out.stop_reason.cpu().numpy() == 0
            if running.any():
                assert (opt.out.num_iters.cpu().numpy()[running] == steps).all(), name
            if active == 0:
                break
            assert steps < o.max_iters + 3, name
        torch.cuda.synchronize()
        out = opt.out
        it_ref = ref.num_iters.cpu().numpy()
        # trajectories: identical wherever the reference rebuilds H every iteration; problems with eval-only iterations
        # (after a rejected step) re-form the same H at the rolled-back x, equal up to rounding
        st, st_ref = out.stop_reason.cpu().numpy(), ref.stop_reason.cpu().numpy()
        same = (st == st_ref) & (out.num_iters.cpu().numpy() == it_ref)
        assert same.all(), (name, same.mean())
        assert steps == it_ref.max(), name                                       # as many calls as the longest problem needs
        tol = 2e-3 if x.dtype == torch.float32 else 1e-9
        assert float((x - x_ref).abs().max()) < tol, name
        assert np.allclose(out.final_cost.cpu().numpy(), ref.final_cost.cpu().numpy(),
                           rtol=1e-3 if x.dtype == torch.float32 else 1e-9, atol=1e-12), name
        assert np.array_equal(out.num_failures.cpu().numpy(), ref.num_failures.cpu().numpy()), name
        k = it_ref.min()
        assert np.allclose(out.errs.cpu().numpy()[:, :k], ref.errs.cpu().numpy()[:, :k], rtol=1e-4 if x.dtype == torch.float32 else 1e-9)
        assert np.array_equal(out.successes.cpu().numpy()[:, :k], ref.successes.cpu().numpy()[:, :k]), name
        # finished problems are left alone by further steps
        x_done = x.clone()
        assert opt.Step() == 0
        assert torch.equal(x, x_done), name


def test_bounded_run_and_in_place_updates(ta, oracle):
    """`optimizer(x, f, max_iters)`: a bounded number of passes; x moves at every accepted step."""
    A, b, x0, xs = oracle.synth_dense_row(8, 12, 500, np.float64, seed=9)
    model = ta.DenseRow.from_arrays(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
    x = torch.from_numpy(x0.copy()).cuda()
    opt = ta.Optimizer(x, model, ta.Options.benchmark())
    assert torch.equal(x, torch.from_numpy(x0).cuda())                           # construction does not touch x
    out = opt(max_iters=2)
    torch.cuda.synchronize()
    assert (out.num_iters.cpu().numpy() == 2).all() and (out.stop_reason.cpu().numpy() == 0).all()
    assert float((x - torch.from_numpy(x0).cuda()).abs().max()) > 1e-3           # moved
    err2 = float((x - torch.from_numpy(xs).cuda()).abs().max())
    out = opt()                                                                   # run to the end
    torch.cuda.synchronize()
    assert (out.stop_reason.cpu().numpy() > 0).all()
    assert float((x - torch.from_numpy(xs).cuda()).abs().max()) < min(err2, 1e-2)
