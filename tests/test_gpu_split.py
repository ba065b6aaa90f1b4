"""Row-split ("wide") execution: same contract and results as the one-wave-per-problem path, for single huge
problems (BASELINE configs C2: n=6, m=1000 fp64, P=1; C5: SE3, 50 000 residuals fp64, P=1)."""
import numpy as np
import pytest
import torch

from parity import check_trajectories, gpu_dict

pytestmark = pytest.mark.gpu


def _run_both(ta, model, x0, opts, splits):
    xa = x0.clone()
    oa = ta.Optimize(xa, model, opts, history=True)
    xb = x0.clone()
    ob = ta.Optimize(xb, model, opts, history=True, splits=splits)
    torch.cuda.synchronize()
    return (xa, oa), (xb, ob)


@pytest.mark.parametrize("dtype,n,m,P,splits", [
    (np.float64, 6, 1000, 1, 0),      # C2, automatic chunk count
    (np.float64, 6, 1000, 1, 7),      # ragged chunks
    (np.float64, 12, 500, 5, 3),
    (np.float64, 12, 64, 2, 1),       # a single chunk: must equal the fused path up to summation order
    (np.float32, 50, 2000, 2, 16),    # thin-tail layout, fp32
    (np.float64, 18, 400, 3, 4),      # thin-tail layout, fp64
])
def test_split_dense_row_matches_oracle_and_fused(ta, oracle, dtype, n, m, P, splits):
    A, b, x0, xs = oracle.synth_dense_row(P, n, m, dtype, seed=31)
    for opts in (ta.Options.benchmark(), ta.Options()):
        opts.hessian.save_last = True
        ref = oracle.dense_row_lm(A, b, x0, opts.to_pod(), history=True)
        model = ta.DenseRow.from_arrays(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
        x = torch.from_numpy(x0.copy()).cuda()
        out = ta.Optimize(x, model, opts, history=True, splits=splits)
        torch.cuda.synchronize()
        xg = x.cpu().numpy()
        stop = out.stop_reason.cpu().numpy()
        iters = out.num_iters.cpu().numpy()
        assert (stop >= 0).all()
        refd = dict(errs=ref["errs"], succ=ref["succ"], iters=ref["iters"], stop=ref["stop"], x=ref["x"], cost=ref["cost"],
                    fails=ref["fails"], deltas2=ref["deltas2"])
        st = check_trajectories(gpu_dict(out, x), refd, dtype, opts.to_pod(), label=f"split {n}x{m}/{splits}")
        assert st["full"] + st["ties"] == P and all(j >= 2 for j in st["tie_iters"]), st
        if dtype == np.float64:
            assert np.abs(xg - ref["x"]).max() < 1e-8
            assert np.allclose(out.final_cost.cpu().numpy(), ref["cost"], rtol=1e-9)
            assert np.allclose(out.final_hessian.cpu().numpy(), ref["H"], rtol=1e-9, atol=1e-9 * np.abs(ref["H"]).max())
        else:
            assert np.abs(xg - ref["x"]).max() < 2e-3
            assert np.allclose(out.final_cost.cpu().numpy(), ref["cost"], rtol=1e-3)
        assert np.abs(xg - xs).max() < 5e-3
        cnt = out.counters.cpu().numpy()
        assert cnt[3] == P and cnt[0] + cnt[1] >= iters.sum()


def test_split_se3_config5(ta, oracle):
    """C5: one SE3 pose, 25 000 points (50 000 residuals), fp64 — row-split is chosen automatically (P*4 <= #CUs,
    m >= 1024) and must reproduce the oracle's StopReason / iterations / pose."""
    npts = 25000
    data, p0, pstar = oracle.synth_se3_reproj(1, npts, np.float64, seed=4)
    o = ta.Options()
    ref = oracle.se3_reproj_lm(data, p0, npts, o.to_pod())
    model = ta.SE3Reproj(torch.from_numpy(data).cuda(), npts)
    for splits in (None, 0, 5):
        x = torch.from_numpy(p0.copy()).cuda()
        out = ta.Optimize(x, model, o, splits=splits)
        torch.cuda.synchronize()
        xg = x.cpu().numpy()
        assert np.array_equal(out.stop_reason.cpu().numpy(), ref["stop"])
        assert np.array_equal(out.num_iters.cpu().numpy(), ref["iters"])
        assert np.abs(xg - ref["x"]).max() < 1e-9
        assert np.allclose(out.final_cost.cpu().numpy(), ref["cost"], rtol=1e-10)
        assert np.abs(xg - pstar).max() < 5e-4            # pixel noise 0.5 px over 25 000 points
        R = xg[:, :9].reshape(3, 3)
        assert np.abs(R @ R.T - np.eye(3)).max() < 1e-12


def test_split_failure_paths(ta, oracle):
    """NaN in one chunk of one problem -> kSystemHasNaNOrInf for that problem only (tests/basic.cpp:147-218)."""
    A, b, x0, _ = oracle.synth_dense_row(3, 6, 256, np.float64, seed=2)
    b[1, 200] = np.nan
    o = ta.Options()
    ref = oracle.dense_row_lm(A, b, x0, o.to_pod())
    model = ta.DenseRow.from_arrays(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
    x = torch.from_numpy(x0.copy()).cuda()
    out = ta.Optimize(x, model, o, splits=4)
    torch.cuda.synchronize()
    assert np.array_equal(out.stop_reason.cpu().numpy(), ref["stop"])
    assert np.array_equal(out.num_iters.cpu().numpy(), ref["iters"])
    assert int(out.stop_reason[1]) == -2 and np.array_equal(x.cpu().numpy()[1], x0[1])


def test_split_rejected_for_other_models(ta):
    with pytest.raises(ta.ToaError):
        ta.Optimize(torch.ones(2, 1, dtype=torch.float64, device="cuda"), ta.Sqrt2(2, torch.float64), splits=2)


@pytest.mark.parametrize("dtype,n,m,P", [(np.float64, 6, 1000, 100), (np.float64, 15, 600, 37), (np.float32, 12, 2000, 200)])
def test_team_form_for_batches_of_small_problems(ta, oracle, dtype, n, m, P):
    """Automatic selection for batches of SMALL problems (n <= 15, 512..4096 rows, up to 1-2 per CU): one workgroup per
    problem, the chunk partials folded through LDS.  Same results as the oracle; the problems of a batch are independent
    (a sub-batch reproduces its rows bit for bit: the fold order depends only on the shape)."""
    A, b, x0, xs = oracle.synth_dense_row(P, n, m, dtype, seed=77)
    model = ta.DenseRow.from_arrays(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
    for opts in (ta.Options.benchmark(), ta.Options()):
        ref = oracle.dense_row_lm(A, b, x0, opts.to_pod())
        x = torch.from_numpy(x0.copy()).cuda()
        out = ta.Optimize(x, model, opts)
        torch.cuda.synchronize()
        assert (out.stop_reason.cpu().numpy() >= 0).all()
        assert np.abs(x.cpu().numpy() - ref["x"]).max() < (1e-8 if dtype == np.float64 else 2e-3)
        if dtype == np.float64:
            assert np.array_equal(out.stop_reason.cpu().numpy(), ref["stop"])
            assert np.array_equal(out.num_iters.cpu().numpy(), ref["iters"])
            assert np.allclose(out.final_cost.cpu().numpy(), ref["cost"], rtol=1e-9)
        cnt = out.counters.cpu().numpy()
        assert cnt[3] == P and cnt[2] >= P
        K = 11
        sub_model = ta.DenseRow.from_arrays(torch.from_numpy(A[5:5 + K]).cuda(), torch.from_numpy(b[5:5 + K]).cuda())
        xsub = torch.from_numpy(x0[5:5 + K].copy()).cuda()
        ta.Optimize(xsub, sub_model, opts)
        torch.cuda.synchronize()
        assert torch.equal(xsub, x[5:5 + K])
