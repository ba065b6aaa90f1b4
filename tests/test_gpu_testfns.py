"""GPU parity of the LM state machine's hard branches: the analytic functions of the reference's optimizer tests
(tests/optimize_easy.cpp:35-221 Rosenbrock / plateau / Powell with exact Hessians; tests/optimize_hard.cpp:34-102
Beale / Himmelblau) as device Accumulate callbacks, run with the reference tests' own options from the reference's
start AND a batch of perturbed starts, against the oracle's trajectories (bad steps, indefinite Hessians ->
failed solves -> re-damping, rollbacks, every StopReason that occurs)."""
import numpy as np
import pytest
import torch

from parity import first_divergence

pytestmark = pytest.mark.gpu

CASES = {  # name: (reference start, known answer, tolerance of the reference test)
    "rosenbrock": ([-1.2, 1.0], [1.0, 1.0], 1e-5),
    "plateau": ([3.0, 3.0], [np.pi, np.pi], 1e-4),
    "powell": ([3.0, -1.0, 0.0, 1.0], [0.0, 0.0, 0.0, 0.0], 1e-3),
    "beale": ([1.0, 1.0], [3.0, 0.5], 1e-4),
    "himmelblau": ([3.5, 2.5], [3.0, 2.0], 1e-4),
}


def _options(ta, oracle, name):
    o = ta.Options()
    for path, val in oracle.testfn_options(name).items():
        obj = o
        parts = path.split(".")
        for q in parts[:-1]:
            obj = getattr(obj, q)
        setattr(obj, parts[-1], val)
    return o


@pytest.mark.parametrize("name", list(CASES))
def test_accumulate_callback(ta, oracle, name):
    """(g, H, cost) of the manual callbacks vs the oracle's, incl. the cost-only form (grad == nullptr)."""
    x0, _, _ = CASES[name]
    rng = np.random.default_rng(3)
    x = np.array(x0)[None, :] + rng.uniform(-0.5, 0.5, (32, len(x0)))
    g_ref, H_ref, c_ref = oracle.testfn_accumulate(name, x)
    model = ta.TestFn(name, 32)
    g, H, c, nres = ta.accumulate(model, torch.from_numpy(x).cuda())
    assert np.allclose(g.cpu().numpy(), g_ref, rtol=1e-12, atol=1e-12 * np.abs(g_ref).max())
    assert np.allclose(H.cpu().numpy(), H_ref, rtol=1e-12, atol=1e-12 * np.abs(H_ref).max())
    assert np.allclose(c.cpu().numpy(), c_ref, rtol=1e-12, atol=1e-300)
    assert (nres.cpu().numpy() == model.m).all()
    c0 = ta.accumulate(model, torch.from_numpy(x).cuda(), want_grad=False)[2]
    assert np.allclose(c0.cpu().numpy(), c_ref, rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize("name", list(CASES))
def test_reference_start_known_answer(ta, oracle, name):
    """The reference test itself: its start, its options, its REQUIREs."""
    x0, answer, tol = CASES[name]
    o = _options(ta, oracle, name)
    x = torch.tensor([x0], dtype=torch.float64, device="cuda")
    out = ta.Optimize(x, ta.TestFn(name, 1), o, history=True)
    torch.cuda.synchronize()
    xg = x.cpu().numpy()[0]
    if name != "himmelblau":  # tests/optimize_hard.cpp:72-102 only checks x
        assert int(out.stop_reason[0]) >= 0                                   # REQUIRE(out.Succeeded())
    if name == "rosenbrock":
        assert 1 <= int(out.stop_reason[0]) < 5                               # REQUIRE(out.Converged())
        k = int(out.num_iters[0])
        assert (out.successes.cpu().numpy()[0, :k] == 0).any()                # the bad-step branch ran on the device
    if name == "powell":
        assert np.abs(xg).max() < tol
    else:
        assert np.abs(xg - np.array(answer)).max() < tol


@pytest.mark.parametrize("name", list(CASES))
def test_batch_of_starts_matches_oracle(ta, oracle, name):
    """256 perturbed starts with the reference test's options.  The device trajectory (per-iteration cost, accept /
    reject flag, failure counters, StopReason) must equal the oracle's up to the first decision taken AT THE ROUND-OFF
    FLOOR: after a rejected step tinyopt rolls x back and re-evaluates there, so `err < final_cost` compares two
    numbers that are equal in exact arithmetic (optimizer.h:428-429) and the outcome is decided by the last bit of the
    cost evaluation — the only place where the two implementations may legitimately part.  Everything before such a
    tie must agree; problems without a tie must agree to the end."""
    x0, _, _ = CASES[name]
    rng = np.random.default_rng(11)
    P = 256
    starts = np.array(x0)[None, :] + rng.uniform(-0.3, 0.3, (P, len(x0)))
    starts[0] = x0
    o = _options(ta, oracle, name)
    hs = o.max_iters + 2
    ref = oracle.testfn_lm(name, starts, o.to_pod(), hist_stride=hs)
    x = torch.from_numpy(starts.copy()).cuda()
    out = ta.Optimize(x, ta.TestFn(name, P), o, history=True)
    torch.cuda.synchronize()
    stop, iters = out.stop_reason.cpu().numpy(), out.num_iters.cpu().numpy()
    fails = out.num_failures.cpu().numpy()
    errs, succ = out.errs.cpu().numpy(), out.successes.cpu().numpy()
    xs = x.cpu().numpy()
    full, ties = 0, 0
    for p in range(P):
        k = first_divergence(errs[p], succ[p], iters[p], ref["errs"][p], ref["succ"][p], ref["iters"][p])
        if k is None:
            full += 1
            assert stop[p] == ref["stop"][p] and fails[p] == ref["fails"][p], (p, stop[p], ref["stop"][p])
            assert np.abs(xs[p] - ref["x"][p]).max() < 1e-6 * max(1.0, np.abs(ref["x"][p]).max())
            continue
        # the decision at iteration k must be a tie: the evaluated cost equals the last ACCEPTED cost to round-off
        assert k >= 1, (p, "diverged at the first iteration")
        acc = [i for i in range(k) if succ[p, i] or i == 0][-1]
        e_gpu, e_ref, last_ok = errs[p, k], ref["errs"][p, k], errs[p, acc]
        assert np.isclose(e_gpu, e_ref, rtol=1e-7, atol=1e-13), (p, k, e_gpu, e_ref)      # same point evaluated
        assert abs(e_gpu - last_ok) <= 1e-9 * max(abs(last_ok), 1e-300) + 1e-15, (p, k, e_gpu, last_ok)
        ties += 1
    assert full + ties == P                   # every problem either identical to the end or parted at a PROVEN tie
    k0 = first_divergence(errs[0], succ[0], iters[0], ref["errs"][0], ref["succ"][0], ref["iters"][0])
    assert k0 is None                         # the reference's own start runs identically to the end
    if name in ("rosenbrock", "plateau", "beale"):
        assert ref["fails"].sum() > 0 and fails.sum() > 0     # rejected steps did occur on both sides


def _success_checks(out, expected_stop, min_iters=2, max_iters=5):
    """SuccessChecks of tests/basic.cpp:22-37."""
    assert int(out.stop_reason[0]) >= 0                                  # Succeeded()
    k = int(out.num_iters[0])
    assert min_iters <= k <= max_iters
    assert float(out.final_cost[0]) < 1e-5
    assert 1 <= int(out.stop_reason[0]) < 5                              # Converged()
    assert out.final_hessian is not None and float(out.final_hessian[0, 0, 0]) > 0
    assert int(out.stop_reason[0]) == expected_stop


def test_basic_x_minus_2(ta, oracle):
    """tests/basic.cpp:41-54 (LM -> kMinDeltaNorm), :72-87 (GaussNewton -> kMinError), :107-124 (min_error = 1e-2 with
    GaussNewton -> kMinError): the scalar callback grad = res, H = 1, cost = |res|, x0 = 1."""
    for solver, min_error, expected in ((0, None, ta.StopReason.kMinDeltaNorm),      # 0 = LevenbergMarquardt, 1 = GaussNewton
                                        (1, None, ta.StopReason.kMinError), (1, 1e-2, ta.StopReason.kMinError)):
        o = ta.Options()
        o.solver_type = solver
        if min_error is not None:
            o.min_error = min_error
        x = torch.ones(1, 1, dtype=torch.float64, device="cuda")
        out = ta.Optimize(x, ta.TestFn("x_minus_2", 1), o, history=True)
        torch.cuda.synchronize()
        _success_checks(out, int(expected))
        ref = oracle.testfn_lm("x_minus_2", np.ones((1, 1)), o.to_pod(), hist_stride=o.max_iters + 2)
        assert int(out.stop_reason[0]) == ref["stop"][0] and int(out.num_iters[0]) == ref["iters"][0]
        k = int(out.num_iters[0])
        assert np.allclose(out.errs.cpu().numpy()[0, :k], ref["errs"][0, :k], rtol=1e-12, atol=1e-300)
        assert abs(float(x[0, 0]) - ref["x"][0, 0]) < 1e-12
