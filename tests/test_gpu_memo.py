"""The memo of the last accepted linearisation (lm_device.hpp): a Build whose x equals, bit for bit, a point whose Gram is
still at hand is served from that Gram instead of a second pass over the rows (optimizer.h:283-287 + :266: the iteration after
a rejected step accumulates again at the rolled-back point; optimizer.h:358-393: a failed solve re-enters Build at the same x).

What must hold: (i) every result — x, StopReason, iteration and failure counts, the whole cost / |dx|^2 / accept history, the
exported Hessian — is BIT-IDENTICAL with the memo on and off (toa_tuning::memo_off), also through option sets that keep a problem
bouncing between rejected steps; (ii) the work is really saved: passes streamed + Builds served from the memo == passes
streamed without it; (iii) the oracle (which always re-accumulates) still agrees."""
import os

import numpy as np
import pytest
import torch

from parity import check_trajectories, gpu_dict

pytestmark = pytest.mark.gpu


def _solve(ta, model, x0, opts, memo):
    with ta.api.default_context().tuning(memo_off=0 if memo else 1):     # toa_tuning (the product reads no environment variable)
        x = torch.from_numpy(x0).cuda()
        out = ta.Optimize(x, model, opts, history=True)
        torch.cuda.synchronize()
    return x.cpu().numpy(), out


def _same_bits(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.tobytes() == b.tobytes()


def _assert_identical(xa, oa, xb, ob):
    assert _same_bits(xa, xb), "x differs with the memo on"
    for f in ("stop_reason", "num_iters", "num_failures", "num_consec_failures", "final_cost", "final_num_residuals",
              "final_rerr_dec", "errs", "deltas2", "successes", "final_hessian"):
        ta_, tb_ = getattr(oa, f), getattr(ob, f)
        if ta_ is None:
            assert tb_ is None
            continue
        assert _same_bits(ta_.cpu().numpy(), tb_.cpu().numpy()), f"{f} differs with the memo on"


def _opts(ta, **kw):
    o = ta.Options.benchmark()
    o.hessian.save_last = True
    for k, v in kw.items():
        setattr(o, k, v)
    return o


CASES = [
    # dtype, n, m, P, option overrides
    (np.float32, 50, 2000, 96, {}),                                   # the C4 shape: every problem ends on rejected steps
    (np.float32, 50, 2000, 48, dict(max_consec_failures=8, max_iters=30)),   # long eval-only chains after the roll-back
    (np.float32, 50, 2000, 48, dict(max_consec_failures=0, max_iters=25)),   # no failure limit at all
    (np.float64, 12, 500, 64, dict(min_rerr_dec=0.0, min_step_norm2=0.0, max_iters=20)),   # C3 shape driven to the fp64 floor
    (np.float64, 50, 300, 16, dict(min_rerr_dec=0.0, min_step_norm2=0.0, max_iters=20, max_consec_failures=6)),
    (np.float32, 31, 400, 32, {}),                                    # NBM = 2, no thin tail
    (np.float32, 18, 300, 32, {}),                                    # NBM = 1 + thin 3
    (np.float32, 6, 200, 32, dict(max_consec_failures=5)),            # one block
    (np.float32, 63, 500, 16, {}),                                    # NBM = 4
]


@pytest.mark.parametrize("dtype,n,m,P,over", CASES)
def test_results_do_not_depend_on_the_memo(ta, oracle, dtype, n, m, P, over):
    A, b, x0, _ = oracle.synth_dense_row(P, n, m, dtype, seed=77 + n)
    model = ta.DenseRow.from_arrays(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
    opts = _opts(ta, **over)
    x_off, o_off = _solve(ta, model, x0, opts, memo=False)
    x_on, o_on = _solve(ta, model, x0, opts, memo=True)
    _assert_identical(x_off, o_off, x_on, o_on)
    c_off = o_off.counters.cpu().numpy()
    c_on = o_on.counters.cpu().numpy()
    assert c_off[4] == 0, "toa_tuning::memo_off must stream every Build"
    assert c_on[0] + c_on[4] == c_off[0], "a Build is either streamed or served from the memo"
    assert c_on[1] == c_off[1] and c_on[2] == c_off[2] and c_on[3] == c_off[3] == P
    # the oracle always re-accumulates: the trajectories must still be its own
    ref = oracle.dense_row_lm(A, b, x0, opts.to_pod(), history=True)
    check_trajectories(gpu_dict(o_on, torch.from_numpy(x_on)), dict(errs=ref["errs"], succ=ref["succ"], iters=ref["iters"], stop=ref["stop"],
                                                  x=ref["x"], cost=ref["cost"], fails=ref["fails"], deltas2=ref["deltas2"]),
                       dtype, opts.to_pod())


def test_the_memo_saves_the_pass_after_a_rejected_step(ta, oracle):
    """At the C4 shape in fp32 every problem ends on the noise floor: rejected step -> roll-back -> accumulate again at the
    point it came from.  Nearly all of those roll-backs restore x exactly, and their Builds must come from the memo."""
    P, n, m = 256, 50, 2000
    A, b, x0, _ = oracle.synth_dense_row(P, n, m, np.float32, seed=5)
    model = ta.DenseRow.from_arrays(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
    opts = ta.Options.benchmark()
    _, o_on = _solve(ta, model, x0, opts, memo=True)
    c = o_on.counters.cpu().numpy()
    fails = o_on.num_failures.cpu().numpy()
    assert (fails > 0).mean() > 0.9, "the workload is expected to end on rejected steps"
    assert c[4] >= 0.8 * (fails > 0).sum(), f"only {c[4]} Builds served from the memo for {(fails > 0).sum()} problems with rejected steps"


def test_memo_with_device_ad_rows(ta, oracle):
    """JetRowModel (device forward-mode AD in MFMA operand order) parks the same Gram registers."""
    P, n, m = 24, 12, 200
    A, b, x0, _ = oracle.synth_dense_row(P, n, m, np.float32, seed=3)
    model = ta.DenseRowAD(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
    opts = _opts(ta, max_consec_failures=6)
    x_off, o_off = _solve(ta, model, x0, opts, memo=False)
    x_on, o_on = _solve(ta, model, x0, opts, memo=True)
    _assert_identical(x_off, o_off, x_on, o_on)
    c_off, c_on = o_off.counters.cpu().numpy(), o_on.counters.cpu().numpy()
    assert c_on[0] + c_on[4] == c_off[0]


# ---- round 5: the same memo beyond one wavefront — a workgroup per problem (64 <= n <= 128, large_fused.hip:
# two H slots per workgroup, a memo hit makes the parked slot the current one again) and the launch-per-stage pipeline
# (128 < n <= 1024 and the few-huge-problems form, large_n.hip: a memo slot per problem, the rows / Gram kernels skip the problem)
LARGE_CASES = [
    # dtype, n, m, P, option overrides, forced pipeline
    (np.float32, 128, 700, 12, {}, 0),                                        # tile-split pass (fp32, 112 < n <= 128)
    (np.float32, 100, 500, 12, dict(max_consec_failures=8, max_iters=25), 0),   # row-split pass + fold
    (np.float32, 64, 300, 16, dict(max_consec_failures=0, max_iters=20), 0),
    (np.float64, 72, 300, 8, dict(min_rerr_dec=0.0, min_step_norm2=0.0, max_iters=20, max_consec_failures=6), 0),
    (np.float64, 128, 520, 4, dict(min_rerr_dec=0.0, min_step_norm2=0.0, max_iters=16, max_consec_failures=5), 0),   # two half-tile passes
    (np.float32, 160, 700, 20, {}, 0),                                        # n > 128: own Gram + own Cholesky, two lanes
    (np.float32, 256, 900, 6, dict(max_consec_failures=8, max_iters=25), 0),  # the operand-sharing Gram, one lane
    (np.float32, 132, 600, 5, dict(max_consec_failures=0, max_iters=18), 0),
    (np.float32, 96, 500, 9, {}, 1),                                          # the pipeline forced below 128 (toa_tuning::large_pipeline)
    (np.float64, 160, 500, 4, dict(min_rerr_dec=0.0, min_step_norm2=0.0, max_iters=16, max_consec_failures=5), 0),   # library Gram (fp64)
    (np.float32, 130, 450, 5, {}, 0),                                         # rows not 16-byte aligned: general rows kernel + library GEMM / GEMV
]


@pytest.mark.parametrize("dtype,n,m,P,over,pipeline", LARGE_CASES)
def test_results_do_not_depend_on_the_memo_beyond_one_wavefront(ta, oracle, dtype, n, m, P, over, pipeline):
    A, b, x0, _ = oracle.synth_dense_row(P, n, m, dtype, seed=177 + n)
    model = ta.DenseRowNatural(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
    opts = _opts(ta, **over)
    ctx = ta.api.default_context()
    with ctx.tuning(large_pipeline=pipeline):
        x_off, o_off = _solve_t(ta, model, x0, opts, dict(memo_off=1, large_pipeline=pipeline))
        x_on, o_on = _solve_t(ta, model, x0, opts, dict(memo_off=0, large_pipeline=pipeline))
    _assert_identical(x_off, o_off, x_on, o_on)
    c_off = o_off.counters.cpu().numpy()
    c_on = o_on.counters.cpu().numpy()
    assert c_off[4] == 0, "toa_tuning::memo_off must stream every Build"
    assert c_on[0] + c_on[4] == c_off[0], "a Build is either streamed or served from the memo"
    assert c_on[1] == c_off[1] and c_on[3] == c_off[3] == P
    if dtype == np.float32:     # (in fp64 (x + dx) - dx rarely restores x bit for bit: nothing to read back, nothing to assert)
        assert c_on[4] > 0, "no Build was served from the memo: the case does not exercise it"
    ref = oracle.dense_row_lm(A, b, x0, opts.to_pod(), history=True)
    check_trajectories(gpu_dict(o_on, torch.from_numpy(x_on)), dict(errs=ref["errs"], succ=ref["succ"], iters=ref["iters"], stop=ref["stop"],
                                                  x=ref["x"], cost=ref["cost"], fails=ref["fails"], deltas2=ref["deltas2"]),
                       dtype, opts.to_pod())


def _solve_t(ta, model, x0, opts, tune):
    with ta.api.default_context().tuning(**tune):
        x = torch.from_numpy(x0).cuda()
        out = ta.Optimize(x, model, opts, history=True)
        torch.cuda.synchronize()
    return x.cpu().numpy(), out
