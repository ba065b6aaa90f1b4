"""Tie-aware trajectory parity between two runs of the LM state machine (the HIP path vs the CPU oracle).

SURVEY §8(c) asks for identical StopReason / iteration counts.  Two correct implementations of
optimizer.h:331-539 can still part in exactly one way: a decision `err < final_cost` (optimizer.h:428-429) or a stop
test (`rel_derr < min_rerr_dec`, `dx_norm2 < min_step_norm2`, optimizer.h:519-534) taken when the two numbers
compared are equal to within the round-off of the cost evaluation — the Gram on the matrix cores and the oracle's
sequential sum round differently, so the sign of a difference that is zero in exact arithmetic is not reproducible.

`check_trajectories` therefore requires, per problem:
  * iteration by iteration up to the first difference: the same cost (to `err_rtol`) and the same accept / reject flag;
  * no difference at all  =>  identical StopReason, iteration count, failure count, and x / final cost to tolerance;
  * a difference at iteration k  =>  it must be PROVEN a tie: both sides evaluated the same point (costs agree to
    `err_rtol`) and that cost equals the last ACCEPTED cost to `floor_rtol` (the round-off floor of the evaluation);
    after it every cost either run evaluates must stay on that floor, and both must land on the same x / final cost.
Nothing is accepted on a fraction-of-problems basis.
"""
from __future__ import annotations

import numpy as np

# err_rtol: agreement of one cost evaluated by both sides at the same point; floor_rtol: how close to the last accepted
# cost a decision must be to count as "decided by round-off".  fp32: the oracle sums m ~ 10^3 squared residuals
# sequentially in float (relative error up to m * 6e-8), the matrix cores in blocked order.
TOL = {
    "f64": dict(err_rtol=1e-9, floor_rtol=1e-11, x_tol=1e-8, cost_rtol=1e-9),
    "f32": dict(err_rtol=5e-4, floor_rtol=5e-4, x_tol=2e-3, cost_rtol=1e-3),
}


def tag_of(dtype) -> str:
    return "f32" if np.dtype(dtype) == np.float32 else "f64"


def first_divergence(errs_a, succ_a, ka, errs_b, succ_b, kb, rtol=1e-7, atol=1e-13):
    """Index of the first iteration at which two trajectories differ (cost value or accept/reject flag); the common
    length when one is a strict prefix of the other; None when they are the same to the end."""
    for i in range(min(ka, kb)):
        if bool(succ_a[i]) != bool(succ_b[i]) or not np.isclose(errs_a[i], errs_b[i], rtol=rtol, atol=atol):
            return i
    return None if ka == kb else min(ka, kb)


def _last_accepted(errs, succ, k):
    """Cost of the last accepted iteration before k (iteration 0 is always accepted, optimizer.h:441)."""
    acc = [i for i in range(k) if succ[i] or i == 0]
    return errs[acc[-1]] if acc else None


def _threshold_straddle(pod, stop_short, errs, d2, succ, j, rtol):
    """One run stopped after iteration j on a threshold test (optimizer.h:519-534) and the other did not: true when
    the tested quantity of that iteration sits on its threshold to `rtol` (relative), i.e. the test was decided by
    round-off.  errs / d2 / succ: the history of either run (they agree up to j)."""
    err = errs[j]
    if stop_short == 1 and pod.min_error > 0:                       # kMinError: err < min_error
        return abs(err - pod.min_error) <= rtol * pod.min_error
    if stop_short == 2 and pod.min_rerr_dec > 0 and j >= 1:         # kMinRelError: 0 < rel_derr < min_rerr_dec
        last_ok = _last_accepted(errs, succ, j)
        rel = (last_ok - err) / last_ok if last_ok else 0.0
        # rel is a difference of two costs: its own round-off is err_noise / last_ok
        return abs(rel - pod.min_rerr_dec) <= rtol or abs(rel) <= rtol
    if stop_short == 3 and pod.min_step_norm2 > 0 and d2 is not None:   # kMinDeltaNorm: |dx|^2 < min_step_norm2
        return 0.25 * pod.min_step_norm2 <= d2[j] <= 4.0 * pod.min_step_norm2
    return False


def check_trajectories(gpu, ref, dtype, pod, *, tol=None, x_scale=1.0, label=""):
    """gpu / ref: dicts of numpy arrays — errs [P, hs], succ [P, hs], iters [P], stop [P], x [P, n], cost [P],
    optionally fails [P], deltas2 [P, hs]; pod: the ToaOptions of the run.
    Returns {"full": problems identical to the end, "ties": problems that part at a proven tie, "tie_iters": the iterations at which they part}.  Raises AssertionError on the first unexplained difference."""
    t = dict(TOL[tag_of(dtype)])
    if tol:
        t.update(tol)
    P = len(gpu["iters"])
    full, ties, tie_iters = 0, 0, []
    for p in range(P):
        ge, gs, gk = gpu["errs"][p], gpu["succ"][p], int(gpu["iters"][p])
        re_, rs, rk = ref["errs"][p], ref["succ"][p], int(ref["iters"][p])
        hs = min(len(ge), len(re_))
        # failed problems record no history (tests/basic.cpp:147-218: empty errs): compare the verdict only
        if gpu["stop"][p] < 0 or ref["stop"][p] < 0:
            assert gpu["stop"][p] == ref["stop"][p] and gk == rk, (label, p, gpu["stop"][p], ref["stop"][p], gk, rk)
            full += 1
            continue
        k = first_divergence(ge, gs, min(gk, hs), re_, rs, min(rk, hs), rtol=t["err_rtol"], atol=1e-300)
        xg, xr = np.asarray(gpu["x"][p], np.float64), np.asarray(ref["x"][p], np.float64)
        xs = max(1.0, float(np.abs(xr).max())) * x_scale
        if k is None:
            full += 1
            assert gpu["stop"][p] == ref["stop"][p], (label, p, "stop", gpu["stop"][p], ref["stop"][p])
            if "fails" in gpu and "fails" in ref and gpu["fails"] is not None and ref["fails"] is not None:
                assert gpu["fails"][p] == ref["fails"][p], (label, p, "fails", gpu["fails"][p], ref["fails"][p])
            assert np.abs(xg - xr).max() < t["x_tol"] * xs, (label, p, "x", np.abs(xg - xr).max())
            assert np.isclose(gpu["cost"][p], ref["cost"][p], rtol=t["cost_rtol"], atol=1e-300), (label, p, "cost")
            continue
        # ---- the two runs part at iteration k: prove it is a tie
        assert k >= 1, (label, p, "diverged at the first iteration", ge[0], re_[0])
        if k < min(gk, rk):
            j = k                                     # different accept/reject (or cost) at iteration k
            assert np.isclose(ge[j], re_[j], rtol=t["err_rtol"], atol=1e-300), (label, p, j, "different point", ge[j], re_[j])
        else:
            j = k - 1                                 # same flags, but only one side's stop test fired after iteration k-1
        last_ok = _last_accepted(ge, gs, j) if j >= 1 else ge[0]
        gap = abs(ge[j] - last_ok)
        at_floor = gap <= t["floor_rtol"] * max(abs(last_ok), 1e-300)
        if not at_floor and k >= min(gk, rk):         # a stop threshold straddled by round-off (the shorter run stopped)
            short = gpu if gk < rk else ref
            d2 = gpu["deltas2"][p] if gpu.get("deltas2") is not None else None
            at_floor = _threshold_straddle(pod, int(short["stop"][p]), ge, d2, gs, j, t["floor_rtol"])
        assert at_floor, (label, p, f"runs part at iteration {j} away from the round-off floor", ge[j], re_[j], last_ok, gap,
                          int(gpu["stop"][p]), int(ref["stop"][p]))
        ties += 1
        tie_iters.append(j)
        # from here on BOTH runs only trade last bits: every later cost stays on the floor (a run that wandered off it
        # after the tie would be a real difference), and both still land on the same answer.  The iteration counts are
        # not comparable after a tie: an accepted last-bit improvement resets the consecutive-failure counter
        # (optimizer.h:441-446), so the length of the tail is itself decided by round-off.
        band = 10 * t["floor_rtol"] * max(abs(last_ok), 1e-300)
        for name, e, kk in (("gpu", ge, min(gk, hs)), ("ref", re_, min(rk, hs))):
            tail = np.abs(np.asarray(e[j:kk], np.float64) - last_ok)
            assert (tail <= band).all(), (label, p, name, "left the floor after the tie", j, e[j:kk], last_ok)
        assert gpu["stop"][p] >= 0 and ref["stop"][p] >= 0, (label, p, gpu["stop"][p], ref["stop"][p])
        assert np.abs(xg - xr).max() < t["x_tol"] * xs, (label, p, "x after tie", np.abs(xg - xr).max())
        assert np.isclose(gpu["cost"][p], ref["cost"][p], rtol=max(t["cost_rtol"], 10 * t["floor_rtol"]), atol=1e-300), \
            (label, p, "cost after tie", gpu["cost"][p], ref["cost"][p])
    assert full + ties == P
    return {"full": full, "ties": ties, "tie_iters": tie_iters}


def gpu_dict(out, x):
    """tinyopt_amd Output (history=True) + x tensor -> the numpy dict check_trajectories takes."""
    d = dict(errs=out.errs.cpu().numpy(), succ=out.successes.cpu().numpy(), iters=out.num_iters.cpu().numpy(),
             stop=out.stop_reason.cpu().numpy(), x=x.cpu().numpy(), cost=out.final_cost.cpu().numpy(),
             fails=out.num_failures.cpu().numpy(), deltas2=out.deltas2.cpu().numpy())
    return d


# ---- the second reading of the LM state machine (tests/golden/make_reference_traces.py) -------------------------------------
def load_reference_traces(name="reference_traces.json"):
    """[(case dict, ToaOptions)] of tests/golden/reference_traces.json — per-iteration traces of an independent Python
    restatement of optimizer.h / lm.h / gn.h (written from the reference, not from oracle/).  name = "reference_traces_robust.json":
    part 2 (round 5), the M-estimators inside the loop (make_reference_traces_robust.py)."""
    import json
    import os
    from tinyopt_amd._capi import ToaOptions
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name)) as f:
        cases = json.load(f)["cases"]
    out = []
    for c in cases:
        o = c["options"]
        p = ToaOptions()
        p.solver_type = 0 if o["solver"] == "lm" else 1
        p.max_iters = o["max_iters"]
        p.min_error, p.min_rerr_dec = o["min_error"], o["min_rerr_dec"]
        p.min_step_norm2, p.min_grad_norm2 = o["min_step_norm2"], o["min_grad_norm2"]
        p.max_total_failures, p.max_consec_failures = o["max_total_failures"], o["max_consec_failures"]
        p.damping_init, p.damping_min, p.damping_max = o["damping_init"], o["damping_range"][0], o["damping_range"][1]
        p.good_factor, p.bad_factor = o["good_factor"], o["bad_factor"]
        p.grad_clipping, p.check_min_H_diag = o["grad_clipping"], o["check_min_H_diag"]
        p.check_final_cost, p.use_step_quality_approx = int(o["check_final_cost"]), int(o["use_step_quality_approx"])
        p.use_ldlt, p.H_is_full, p.save_last = int(o["use_ldlt"]), 1, 1
        p.use_squared_norm, p.downscale_by_2, p.normalize = int(o["use_squared_norm"]), int(o["downscale_by_2"]), int(o["normalize"])
        out.append((c, p))
    return out


def check_against_trace(c, got, label=""):
    """`got` (dict: errs, deltas2, succ, stop, iters, fails, x, cost — one problem) against one fixture case: identical
    StopReason / iteration / failure counts and accept-reject flags, costs and steps to 1e-8 relative (the traces amplify
    the last-bit differences of two LDL^T operation orders over up to 66 iterations; measured <= 1e-10).
    Returns "full", or "tie" when the two part at a PROVEN tie: after a rejected step the loop rolls x back and accumulates
    again at that point (optimizer.h:283-287, :266), so `err < final_cost` (:428-429) compares two numbers that are equal in
    exact arithmetic — 0 exactly when (x + dx) - dx restored x bit for bit, a last-bit coin toss otherwise.  Then: same
    point evaluated on both sides, its cost equal to the last accepted cost to 1e-12, everything before identical."""
    f32 = c.get("dtype", "float64") == "float32"   # the double-precision reading against an fp32 run: FloatEpsilon-class tolerances
    rt, rt2, xt = (2e-3, 2e-2, 5e-3) if f32 else (1e-8, 1e-6, 1e-8)
    k = len(c["errs"])
    e, d2 = np.asarray(c["errs"]), np.asarray(c["deltas2"])
    kg = int(got["iters"])
    ge, gs = np.asarray(got["errs"]), np.asarray(got["succ"], dtype=np.int64)
    fs = np.asarray(c["successes"], dtype=np.int64)
    scale = np.abs(e).max() if k else 1.0
    kk = min(k, kg, len(ge))
    div = None
    for i in range(kk):
        if gs[i] != fs[i]:
            div = i
            break
    upto = kk if div is None else div + 1
    assert np.allclose(ge[:upto], e[:upto], rtol=rt, atol=(1e-6 if f32 else 1e-12) * scale), (label, "cost history", div)
    if div is not None:
        assert not f32, (label, "an fp32 fixture is only emitted when every decision is 1e-3 from flipping", div)
        assert div >= 2, (label, "parted before any step was rejected", div)
        acc_f = [i for i in range(div) if fs[i] or i == 0][-1]
        assert abs(e[div] - e[acc_f]) <= 1e-12 * abs(e[acc_f]) and abs(ge[div] - ge[acc_f]) <= 1e-12 * abs(ge[acc_f]), \
            (label, "accept / reject flags differ away from a tie", div, e[div], ge[div], e[acc_f])
        return "tie"
    assert int(got["stop"]) == c["stop_reason"], (label, "stop", int(got["stop"]), c["stop_reason"])
    assert kg == c["num_iters"], (label, "iters", kg, c["num_iters"])
    assert int(got["fails"]) == c["num_failures"], (label, "fails", int(got["fails"]), c["num_failures"])
    if k:
        assert np.allclose(np.asarray(got["deltas2"])[:k], d2, rtol=rt2, atol=(1e-6 if f32 else 1e-12) * max(d2.max(), 1e-300)), (label, "|dx|^2 history")
    xs = np.asarray(c["x"])
    assert np.abs(np.asarray(got["x"]) - xs).max() <= xt * max(1.0, np.abs(xs).max()), (label, "x")
    if c["final_cost"] < 1e300:
        assert abs(float(got["cost"]) - c["final_cost"]) <= rt * abs(c["final_cost"]) + (1e-6 if f32 else 1e-12) * scale, (label, "final cost")
    return "full"
