"""The stepping form (`optimizer.Step(x, acc, out)`, include/tinyopt/optimizers/optimizer.h:199,331-539) and the host-side stop
controls (options.h:96-106; optimizer.h:302-305,529-534) for n >= 64: they run on the launch-per-stage pipeline
(csrc/large_n.hip) whatever the size, with the state parked in the caller's block between calls.  Stepping to the end must
reproduce that pipeline's own `Optimize` BIT FOR BIT (same kernels, same order), one iteration per call."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _problem(ta, oracle, P, n, m, dtype, seed):
    A, b, x0, xs = oracle.synth_dense_row(P, n, m, dtype, seed=seed)
    model = ta.DenseRowNatural(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
    return model, x0, xs


CASES = [(6, 64, 256, np.float32), (5, 96, 300, np.float64), (4, 128, 512, np.float32), (3, 200, 600, np.float32),
         (3, 130, 400, np.float64), (2, 70, 211, np.float32)]   # (70 x 211: rows not 16-byte aligned -> the library Gram)


@pytest.mark.parametrize("P,n,m,dtype", CASES)
def test_stepping_reproduces_the_pipeline_bit_for_bit(ta, oracle, P, n, m, dtype):
    model, x0, _ = _problem(ta, oracle, P, n, m, dtype, seed=n)
    o = ta.Options.benchmark() if n != 96 else ta.Options()   # (the default options keep the last Hessian: final_hessian)
    ctx = ta.api.default_context()
    x_ref = torch.from_numpy(x0.copy()).cuda()
    with ctx.tuning(large_pipeline=1):   # 64 <= n <= 128 would otherwise take the one-kernel form (other summation order)
        ref = ta.Optimize(x_ref, model, o, history=True)
    x = torch.from_numpy(x0.copy()).cuda()
    opt = ta.Optimizer(x, model, o, history=True)
    assert torch.equal(x, torch.from_numpy(x0).cuda())   # construction does not touch x
    steps = 0
    while True:
        active = opt.Step()
        steps += 1
        running = opt.out.stop_reason.cpu().numpy() == 0
        if running.any():
            assert (opt.out.num_iters.cpu().numpy()[running] == steps).all()
        assert active == int(running.sum())
        if active == 0:
            break
        assert steps < o.max_iters + 3
    torch.cuda.synchronize()
    out = opt.out
    assert steps == int(ref.num_iters.max())
    assert torch.equal(x, x_ref)
    for f in ("stop_reason", "num_iters", "final_cost", "num_failures", "errs", "deltas2", "successes", "final_hessian"):
        if getattr(ref, f) is not None:
            assert torch.equal(getattr(out, f), getattr(ref, f)), f
    # (the whole-solve form serves some Builds from the memo of the last accepted linearisation — counters[4]; the stepping form,
    #  whose state block has one H slot, streams every one of them: round 5)
    assert int(out.counters[0]) == int(ref.counters[0] + ref.counters[4]) and int(out.counters[4]) == 0
    assert torch.equal(out.counters[1:4], ref.counters[1:4])
    x_done = x.clone()
    assert opt.Step() == 0 and torch.equal(x, x_done)


def test_rejected_steps_and_eval_only_iterations(ta, oracle):
    """A start far enough out that steps get rejected: roll-backs and eval-only iterations keep solving with the H of the
    last build, which lives in the state block between calls."""
    P, n, m = 4, 80, 240
    A, b, x0, _ = oracle.synth_dense_row(P, n, m, np.float64, seed=11)
    b = b + 40.0 * np.sin(np.arange(m))[None, :]   # large residuals: the first steps overshoot
    model = ta.DenseRowNatural(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
    o = ta.Options(); o.max_iters = 30; o.lm.damping_init = 1e-8; o.min_rerr_dec = 0.0
    ctx = ta.api.default_context()
    x_ref = torch.from_numpy(x0.copy()).cuda()
    with ctx.tuning(large_pipeline=1):
        ref = ta.Optimize(x_ref, model, o, history=True)
    x = torch.from_numpy(x0.copy()).cuda()
    opt = ta.Optimizer(x, model, o, history=True)
    out = opt()
    torch.cuda.synchronize()
    assert torch.equal(x, x_ref)
    for f in ("stop_reason", "num_iters", "final_cost", "num_failures", "successes"):
        assert torch.equal(getattr(out, f), getattr(ref, f)), f
    assert int((ref.successes == 0).sum()) > 0 or int(ref.num_failures.sum()) > 0, "no rejected step in this fixture"


@pytest.mark.parametrize("n,m,dtype", [(70, 200, np.float64), (160, 480, np.float32)])
def test_solver_retries_stay_inside_one_step(ta, oracle, n, m, dtype):
    """A zero column: J^T J is singular and the diagonal damping cannot repair it, so every solve fails and is retried with a
    larger damping INSIDE its iteration (optimizer.h:370-390) until max_consec_failures ends the problem — one Step call."""
    P = 3
    A, b, x0, _ = oracle.synth_dense_row(P, n, m, dtype, seed=7)
    A[1, :, 7] = 0.0
    model = ta.DenseRowNatural(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
    o = ta.Options()
    ctx = ta.api.default_context()
    x_ref = torch.from_numpy(x0.copy()).cuda()
    with ctx.tuning(large_pipeline=1):
        ref = ta.Optimize(x_ref, model, o, history=True)
    assert int(ref.num_failures[1]) >= 2 and int(ref.num_iters[1]) <= 1
    x = torch.from_numpy(x0.copy()).cuda()
    opt = ta.Optimizer(x, model, o, history=True)
    active = opt.Step()
    assert int(opt.out.stop_reason[1]) == int(ref.stop_reason[1]) != 0        # ended within the first call, retries included
    assert active == P - 1
    out = opt()
    torch.cuda.synchronize()
    assert torch.equal(x, x_ref)
    for f in ("stop_reason", "num_iters", "final_cost", "num_failures", "num_consec_failures", "successes"):
        if getattr(ref, f, None) is not None:
            assert torch.equal(getattr(out, f), getattr(ref, f)), f


def test_stop_callback_and_timeout_at_n_96_and_256(ta, oracle):
    for n, m, dtype in ((96, 300, np.float64), (256, 768, np.float32)):
        P = 6
        model, x0, _ = _problem(ta, oracle, P, n, m, dtype, seed=5 + n)
        o = ta.Options()
        ctx = ta.api.default_context()
        xr = torch.from_numpy(x0.copy()).cuda()
        with ctx.tuning(large_pipeline=1):
            ref = ta.Optimize(xr, model, o, history=True)
        e1 = ref.errs.cpu().numpy()[:, 1]
        thr = float(np.median(e1))
        calls = []

        def cb(err, dx2, g2):
            calls.append((err, dx2, g2))
            return len(calls) > P and err > thr      # from the second pass on

        o2 = ta.Options(); o2.stop_callback = cb
        x = torch.from_numpy(x0.copy()).cuda()
        out = ta.Optimize(x, model, o2, history=True)
        torch.cuda.synchronize()
        stop, iters = out.stop_reason.cpu().numpy(), out.num_iters.cpu().numpy()
        user = e1 > thr
        assert user.sum() == P // 2
        assert (stop[user] == int(ta.StopReason.kUserStopped)).all() and (iters[user] == 2).all()
        assert np.array_equal(stop[~user], ref.stop_reason.cpu().numpy()[~user])
        assert np.array_equal(iters[~user], ref.num_iters.cpu().numpy()[~user])
        keep = torch.from_numpy(~user).cuda()
        assert torch.equal(x[keep], xr[keep])                       # untouched by the neighbours' stop
        assert np.array_equal(sorted(c[0] for c in calls[:P]), sorted(out.errs.cpu().numpy()[:, 0].astype(np.float64)))
        assert np.array_equal(sorted(c[1] for c in calls[:P]), sorted(out.deltas2.cpu().numpy()[:, 0].astype(np.float64)))
        assert all(c[2] > 0 for c in calls)
        um = torch.from_numpy(user).cuda()
        assert (out.final_cost[um] > 0).all() and out.final_hessian[um].abs().sum() > 0

        seen = []
        o4 = ta.Options()
        o4.stop_callback2 = lambda err, dx, g: (seen.append((dx.shape, g.shape)) or True)
        x = torch.from_numpy(x0.copy()).cuda()
        out = ta.Optimize(x, model, o4)
        assert (out.stop_reason.cpu().numpy() == int(ta.StopReason.kUserStopped)).all()
        assert seen == [((n,), (n,))] * P

        o3 = ta.Options(); o3.max_duration_ms = 1e-6
        x = torch.from_numpy(x0.copy()).cuda()
        out = ta.Optimize(x, model, o3)
        torch.cuda.synchronize()
        assert (out.stop_reason.cpu().numpy() == int(ta.StopReason.kTimedOut)).all()
        assert (out.num_iters.cpu().numpy() == 1).all()
        assert float((x - torch.from_numpy(x0).cuda()).abs().max()) > 1e-4


def test_general_lu_solver_steps_too(ta, oracle):
    """use_ldlt = false (gn.h:157-162) goes through the library's LU in the pipeline: the stepping form takes the same route."""
    model, x0, _ = _problem(ta, oracle, 3, 72, 220, np.float64, seed=2)
    o = ta.Options(); o.hessian.use_ldlt = False
    x_ref = torch.from_numpy(x0.copy()).cuda()
    ref = ta.Optimize(x_ref, model, o)
    assert int(ref.num_iters.min()) >= 2
    x = torch.from_numpy(x0.copy()).cuda()
    out = ta.Optimizer(x, model, o)()
    torch.cuda.synchronize()
    assert torch.equal(x, x_ref) and torch.equal(out.stop_reason, ref.stop_reason) and torch.equal(out.num_iters, ref.num_iters)


def test_refusals(ta, oracle):
    model, x0, _ = _problem(ta, oracle, 2, 32, 128, np.float64, seed=1)
    with pytest.raises(Exception, match="stepping form starts at n = 64"):
        ta.Optimizer(torch.from_numpy(x0.copy()).cuda(), model, ta.Options())
    # (round 5: a loss IS available in the stepping form at n >= 64 — tests/test_gpu_large_n.py::test_stepping_form_with_a_loss_beyond_one_wavefront)
