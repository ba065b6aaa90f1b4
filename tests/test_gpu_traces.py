"""The second reading on the device: every case of tests/golden/reference_traces.json (per-iteration traces of an independent
Python restatement of optimizer.h:242-539 / lm.h / gn.h — tests/golden/make_reference_traces.py) replayed on the device
TestFn models through the C-ABI.  StopReason, iteration and failure counts and the whole accept / reject sequence must be
the fixture's; costs, steps and x agree to 1e-8 (rejected steps, roll-backs, eval-only iterations, failed solves
re-entering Build, max_consec / max_total failure limits, check_final_cost, GaussNewton)."""
import numpy as np
import pytest
import torch

from parity import check_against_trace, load_reference_traces

pytestmark = pytest.mark.gpu


def _options(ta, c):
    o = ta.Options()
    f = c["options"]
    o.solver_type = 0 if f["solver"] == "lm" else 1
    o.max_iters = f["max_iters"]
    for k in ("min_error", "min_rerr_dec", "min_step_norm2", "min_grad_norm2", "max_total_failures", "max_consec_failures",
              "check_final_cost", "use_step_quality_approx", "grad_clipping"):
        setattr(o, k, f[k])
    o.hessian.check_min_H_diag, o.hessian.use_ldlt = f["check_min_H_diag"], bool(f["use_ldlt"])
    o.cost.use_squared_norm, o.cost.downscale_by_2, o.cost.normalize = bool(f["use_squared_norm"]), bool(f["downscale_by_2"]), bool(f["normalize"])
    o.lm.damping_init = f["damping_init"]
    o.lm.damping_range = tuple(f["damping_range"])
    o.lm.good_factor, o.lm.bad_factor = f["good_factor"], f["bad_factor"]
    return o


def test_device_follows_the_second_reading(ta):
    cases = load_reference_traces()
    assert len(cases) >= 55
    ties = 0
    for c, pod in cases:
        o = _options(ta, c)
        got_pod = o.to_pod()
        for name, _ in pod._fields_:       # the Options mirror produces exactly the fixture's POD
            if name not in ("save_last", "H_is_full"):
                assert getattr(got_pod, name) == getattr(pod, name), (name, getattr(got_pod, name), getattr(pod, name))
        tdt = torch.float32 if c.get("dtype", "float64") == "float32" else torch.float64
        x = torch.tensor([c["x0"]], dtype=tdt, device="cuda")
        out = ta.Optimize(x, ta.TestFn(c["function"], 1, dtype=tdt), o, history=True)
        torch.cuda.synchronize()
        got = dict(errs=out.errs.cpu().numpy()[0], deltas2=out.deltas2.cpu().numpy()[0], succ=out.successes.cpu().numpy()[0],
                   stop=int(out.stop_reason[0]), iters=int(out.num_iters[0]), fails=int(out.num_failures[0]),
                   x=x.cpu().numpy()[0], cost=float(out.final_cost[0]))
        if check_against_trace(c, got, label=f"{c['function']} {c['x0']} ({c['comment']})") == "tie":
            ties += 1
            continue
        if c["options"]["solver"] == "lm" and out.final_hessian is not None and c["stop_reason"] >= 0:
            H = np.asarray(c["final_hessian"])
            f32 = tdt == torch.float32
            assert np.allclose(out.final_hessian.cpu().numpy()[0], H, rtol=5e-3 if f32 else 1e-7, atol=(1e-5 if f32 else 1e-9) * np.abs(H).max())
    assert ties <= len(cases) // 8, f"{ties} of {len(cases)} cases parted at a round-off tie"


def test_device_follows_the_second_reading_stepping_form(ta):
    """The same traces through the stepping form (`optimizer.Step`, optimizer.h:331-539): one loop pass per call."""
    cases = load_reference_traces()
    for c, _ in cases[:12] + [cc for cc in cases if cc[0].get("dtype") == "float32" or cc[0]["options"]["grad_clipping"] != 0 or not cc[0]["options"]["use_ldlt"]]:
        o = _options(ta, c)
        tdt = torch.float32 if c.get("dtype", "float64") == "float32" else torch.float64
        x = torch.tensor([c["x0"]], dtype=tdt, device="cuda")
        opt = ta.Optimizer(x, ta.TestFn(c["function"], 1, dtype=tdt), o, history=True)
        out = opt()
        torch.cuda.synchronize()
        got = dict(errs=out.errs.cpu().numpy()[0], deltas2=out.deltas2.cpu().numpy()[0], succ=out.successes.cpu().numpy()[0],
                   stop=int(out.stop_reason[0]), iters=int(out.num_iters[0]), fails=int(out.num_failures[0]),
                   x=x.cpu().numpy()[0], cost=float(out.final_cost[0]))
        check_against_trace(c, got, label=f"stepping {c['function']} {c['x0']}")


def test_device_follows_the_second_reading_of_the_robust_losses(ta):
    """Part 2 of the second reading (tests/golden/reference_traces_robust.json, round 5): the M-estimators inside the loop, on the
    device — DenseRow + toa_set_loss (the launch-per-iteration form's robust passes) in fp64: same decisions, costs, steps, x and
    inlier ratio as the independent Python restatement of robust_norms.h; also through the stepping form."""
    cases = load_reference_traces("reference_traces_robust.json")
    assert len(cases) >= 60
    for k, (c, _) in enumerate(cases):
        o = _options(ta, c)
        o.hessian.save_last = False
        A = torch.tensor([c["A"]], dtype=torch.float64, device="cuda")
        b = torch.tensor([c["b"]], dtype=torch.float64, device="cuda")
        model = ta.DenseRow.from_arrays(A, b).with_loss(c["loss"], c["th2"] ** 0.5)
        for form in (("launch",) if k % 4 else ("launch", "step")):
            x = torch.tensor([c["x0"]], dtype=torch.float64, device="cuda")
            out = ta.Optimize(x, model, o, history=True) if form == "launch" else ta.Optimizer(x, model, o, history=True)()
            torch.cuda.synchronize()
            got = dict(errs=out.errs.cpu().numpy()[0], deltas2=out.deltas2.cpu().numpy()[0], succ=out.successes.cpu().numpy()[0],
                       stop=int(out.stop_reason[0]), iters=int(out.num_iters[0]), fails=int(out.num_failures[0]),
                       x=x.cpu().numpy()[0], cost=float(out.final_cost[0]))
            assert check_against_trace(c, got, label=f"{form}: {c['comment']}") == "full"
            assert abs(float(out.final_inlier_ratio[0]) - c["final_inlier_ratio"]) < 1e-6, c["comment"]


def test_device_follows_the_second_reading_of_a_bundle_adjustment(ta):
    """Part 3 (tests/golden/reference_traces_ba.json, round 5): both device bundle-adjustment forms — points eliminated, Schur
    complement on the matrix cores (toa_ba_run) and the visibility-list pipeline (toa_ba_lists_run) — against the independent
    Python restatement that solves the dense (6C + 3N)^2 system with Jacobians from dual numbers: same decisions, costs, steps,
    inlier ratio, with and without an M-estimator on each observation."""
    cases = load_reference_traces("reference_traces_ba.json")
    assert len(cases) >= 8
    for c, _ in cases:
        o = _options(ta, c)
        o.hessian.save_last = False
        C_, N_ = c["ncam"], c["npts"]
        data = torch.tensor([c["data"]], dtype=torch.float64, device="cuda")
        k = c["num_iters"]
        for form in ("dense mask", "lists"):
            model = ta.BundleAdjustment(data, C_, N_) if form == "dense mask" else ta.BundleAdjustmentLists.from_dense(data, C_, N_)
            if c["loss"]:
                model = model.with_loss(c["loss"], c["th2"] ** 0.5)
            x = torch.tensor([c["x0"]], dtype=torch.float64, device="cuda")
            out = ta.Optimize(x, model, o, history=True)
            torch.cuda.synchronize()
            lab = f"{form}: {c['comment']}"
            assert int(out.num_iters[0]) == k and int(out.stop_reason[0]) == c["stop_reason"] and int(out.num_failures[0]) == c["num_failures"], lab
            assert np.array_equal(out.successes.cpu().numpy()[0][:k].astype(int), np.asarray(c["successes"])), lab
            assert np.allclose(out.errs.cpu().numpy()[0][:k], c["errs"], rtol=1e-7), lab
            assert np.allclose(out.deltas2.cpu().numpy()[0][:k], c["deltas2"], rtol=1e-4, atol=1e-13), lab
            assert abs(float(out.final_cost[0]) - c["final_cost"]) <= 1e-7 * abs(c["final_cost"]), lab
            assert int(out.final_num_residuals[0]) == c["final_num_residuals"], lab
            assert abs(float(out.final_inlier_ratio[0]) - c["final_inlier_ratio"]) < 1e-6, lab
            assert float((x.cpu() - torch.tensor([c["x"]], dtype=torch.float64)).abs().max()) < 1e-4, lab
