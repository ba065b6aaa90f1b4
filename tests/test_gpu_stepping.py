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


def test_stop_callback2_reference_test(ta):
    """tests/basic.cpp:126-143 "User stop callback": x - 2 from x = 1, min_error / min_grad_norm2 disabled,
    stop_callback2 = g.norm() < 2 -> kUserStopped; the callback sees (err, dx, g) of the iteration."""
    seen = []

    def cb(err, dx, g):
        seen.append((err, dx.copy(), g.copy()))
        return float(np.linalg.norm(g)) < 2.0

    o = ta.Options()
    o.min_error = 0.0
    o.min_grad_norm2 = 0.0
    o.stop_callback2 = cb
    x = torch.ones(1, 1, dtype=torch.float64, device="cuda")
    out = ta.Optimize(x, ta.TestFn("x_minus_2", 1), o)
    torch.cuda.synchronize()
    assert int(out.stop_reason[0]) == int(ta.StopReason.kUserStopped)
    assert bool(out.Succeeded()[0]) and not bool(out.Converged()[0])
    assert len(seen) == 1 and int(out.num_iters[0]) == 1
    err, dx, g = seen[0]
    assert err == 1.0 and abs(float(g[0]) + 1.0) < 1e-6 and abs(float(dx[0]) - 1.0 / 1.0001) < 1e-6   # grad = res = -1, H = 1
    assert abs(float(x[0, 0]) - (1.0 + 1.0 / 1.0001)) < 1e-9         # the step of that iteration is still applied
    # finalised like a problem that stops by itself: after ONE iteration prev_lambda is still 0, so Hessian() returns the
    # damped diagonal as is (lm.h:157-171) — the reference's own quirk, reproduced
    assert abs(float(out.final_hessian[0, 0, 0]) - 1.0001) < 1e-7


def test_stop_callback_per_problem_and_timeout(ta, oracle):
    """stop_callback(err, |dx|^2, |g|^2) per problem of a batch: the named problems stop with kUserStopped after that
    iteration, every other problem runs exactly as without the callback; max_duration_ms -> kTimedOut for everything
    still running (tests/basic.cpp:88-106)."""
    A, b, x0, xs = oracle.synth_dense_row(12, 12, 500, np.float64, seed=21)
    model = ta.DenseRow.from_arrays(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
    o = ta.Options()
    xr = torch.from_numpy(x0.copy()).cuda()
    ref = ta.Optimize(xr, model, o, history=True)
    e1 = ref.errs.cpu().numpy()[:, 1]                 # cost seen at iteration 1
    se = np.sort(e1)
    thr = float(0.5 * (se[6] + se[7]))                # between two costs: the fused kernel (reference run) and the stepping form
                                                      # (callback run) sum a_i.x in different orders for fp64 n <= 15 (last bits)
    calls = []

    def cb(err, dx2, g2):
        calls.append((err, dx2, g2))
        return len(calls) > 12 and err > thr          # from the second pass on: stop the problems whose cost is above thr

    o2 = ta.Options()
    o2.stop_callback = cb
    x = torch.from_numpy(x0.copy()).cuda()
    out = ta.Optimize(x, model, o2, history=True)
    torch.cuda.synchronize()
    stop, iters = out.stop_reason.cpu().numpy(), out.num_iters.cpu().numpy()
    user = e1 > thr
    assert user.sum() == 5
    assert (stop[user] == int(ta.StopReason.kUserStopped)).all() and (iters[user] == 2).all()
    assert np.array_equal(stop[~user], ref.stop_reason.cpu().numpy()[~user])
    assert np.array_equal(iters[~user], ref.num_iters.cpu().numpy()[~user])
    keep = torch.from_numpy(~user).cuda()
    assert float((x[keep] - xr[keep]).abs().max()) < 1e-9      # stepping form vs fused kernel: equal up to the fold order of H
    # the callback saw the same (err, |dx|^2) the history records
    assert np.allclose(sorted(c[0] for c in calls[:12]), sorted(out.errs.cpu().numpy()[:, 0]), rtol=1e-12)
    assert np.allclose(sorted(c[1] for c in calls[:12]), sorted(out.deltas2.cpu().numpy()[:, 0]), rtol=1e-12)
    assert all(c[2] > 0 for c in calls)               # |g|^2 is formed for the callbacks even with min_grad_norm2 == 0 (optimizer.h:413-415)
    # stopped problems still carry a finalised Output row
    assert (out.final_cost.cpu().numpy()[user] > 0).all() and out.final_hessian[torch.from_numpy(user).cuda()].abs().sum() > 0

    o3 = ta.Options()
    o3.max_duration_ms = 1e-6
    x = torch.from_numpy(x0.copy()).cuda()
    out = ta.Optimize(x, model, o3)
    torch.cuda.synchronize()
    assert (out.stop_reason.cpu().numpy() == int(ta.StopReason.kTimedOut)).all()
    assert (out.num_iters.cpu().numpy() == 1).all()
    assert bool(out.Succeeded().all()) and not bool(out.Converged().any())
    assert float((x - torch.from_numpy(x0).cuda()).abs().max()) > 1e-3       # the first step was applied


def test_callbacks_are_consulted_on_the_last_allowed_iteration(ta):
    """optimizer.h:529-534 evaluates the callbacks inside Step on every iteration; kMaxIters is only labelled after the loop
    when nothing else stopped it (:320-321).  With max_iters = 1 the loop makes max_iters + 1 = 2 passes (:248-250): a callback
    that trips on the second — the last allowed — pass gives kUserStopped; one that stays false gives kMaxIters, and has still
    seen that last pass.  stop_callback2 receives err as a float (options.h:148-149)."""
    for trip in (True, False):
        seen = []

        def cb(err, dx, g):
            seen.append(err)
            return trip and len(seen) == 2

        o = ta.Options()
        o.min_error = 0.0
        o.min_grad_norm2 = 0.0
        o.max_iters = 1
        o.stop_callback2 = cb
        x = torch.ones(1, 1, dtype=torch.float64, device="cuda")
        out = ta.Optimize(x, ta.TestFn("x_minus_2", 1), o)
        torch.cuda.synchronize()
        assert len(seen) == 2 and int(out.num_iters[0]) == 2
        assert all(e == float(np.float32(e)) for e in seen)
        want = ta.StopReason.kUserStopped if trip else ta.StopReason.kMaxIters
        assert int(out.stop_reason[0]) == int(want)


def test_per_iteration_log_line(ta, oracle):
    """The reference's per-iteration log line (optimizer.h:463-516; Options::log, options.h:113-125; VERDICT r05 "missing" #4) through the
    stepping form, off by default.  sqrt(2) from x0 = 1 reproduces README.md:91-96 — x 1 -> 1.49995 -> 1.41667 -> 1.41422, |dx| 5.00e-01,
    8.33e-02, 2.45e-03 — one line per iteration of every logged problem, each with the iteration's cost as the history records it,
    the damping the solver holds after the step (1 / lambda, lm.h:150-154) and, on request, the inliers."""
    lines = []
    model = ta.Sqrt2(3, torch.float64)
    x = torch.tensor([[1.0], [-0.3], [3.2]], dtype=torch.float64, device="cuda")
    o = ta.Options()
    o.max_iters, o.max_consec_failures = 20, 0            # tests/sqrt2.cpp:22-28
    o.log.enable, o.log.print_x, o.log.sink = True, True, lines.append
    out = ta.Optimize(x, model, o, history=True)
    torch.cuda.synchronize()
    assert len(lines) == int(out.num_iters[0])            # problem 0 only by default
    assert lines[0].startswith("ℹ️#0 x:[1] ") and "|δx|:5.00e-01" in lines[0]
    assert "#1 " in lines[1] and "x:[1.49995" in lines[1] and "|δx|:8.33e-02" in lines[1]
    assert "#2 " in lines[2] and "x:[1.41667" in lines[2] and "|δx|:2.4" in lines[2]
    errs = out.errs.cpu().numpy()
    for it, line in enumerate(lines):
        assert f"ε²:{errs[0, it]:.4e} n:1 " in line and "○:" in line and "τ:" in line
    assert "○:1.00e+04" in lines[0]                   # 1 / damping_init: the first iteration changes nothing (iter == 0)
    # every problem, without emoji / x / time, with the inliers of a loss
    from test_gpu_robust_dense import _with_outliers
    P, n, m = 3, 12, 203
    A, b, x0, xs, _ = _with_outliers(oracle, P, n, m, np.float64)
    lines2 = []
    o2 = ta.Options()
    o2.log.enable, o2.log.print_emoji, o2.log.print_t, o2.log.print_inliers, o2.log.problems, o2.log.sink = True, False, False, True, "all", lines2.append
    md = ta.DenseRow.from_arrays(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda()).with_loss("huber", 0.05)
    xg = torch.from_numpy(x0.copy()).cuda()
    out2 = ta.Optimize(xg, md, o2, history=True)
    torch.cuda.synchronize()
    assert len(lines2) == int(out2.num_iters.sum()) and all(l.startswith("#") and f"n:{m} " in l and "in:" in l for l in lines2)
    # off (the default): the same result, nothing printed
    xq = torch.from_numpy(x0.copy()).cuda()
    outq = ta.Optimize(xq, md, ta.Options(), history=True)
    torch.cuda.synchronize()
    assert torch.equal(outq.num_iters, out2.num_iters) and torch.equal(outq.stop_reason, out2.stop_reason) and float((xq - xg).abs().max()) < 1e-12
