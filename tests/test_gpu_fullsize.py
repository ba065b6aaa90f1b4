"""BASELINE-size runs (C3: 10 000 x n=12 x m=500 fp64; C4 shard: 12 500 x n=50 x m=2000 fp32 = 5.1 GB) checked through
size-independent properties, plus the oracle on a sample of problem ids cut out of the full batch:
  * every problem succeeds and recovers its planted solution;
  * accepted costs never increase along a problem's history (optimizer.h:428-446);
  * idempotence: solving again from the solution stops at once and does not move x;
  * batch independence: a problem's result does not depend on the batch it travels in (work queue, occupancy) —
    the same problem ids solved alone are BIT-identical to their rows in the full batch;
  * sampled parity: those ids against the oracle (the same comparison the small cases make)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CONFIGS = [
    ("c3", 10000, 12, 500, np.float64, torch.float64, 1e-3, 1e-8),
    ("c4", 12500, 50, 2000, np.float32, torch.float32, 2e-2, 2e-3),
]


@pytest.mark.parametrize("tag,P,n,m,dtype,tdt,tol_star,tol_oracle", CONFIGS)
def test_full_size_properties(ta, oracle, tag, P, n, m, dtype, tdt, tol_star, tol_oracle):
    opts = ta.Options.benchmark()            # benchmarks/options.h:10-27, the options of the timed runs
    model, x0, xstar = ta.DenseRow.synthetic(P, n, m, tdt)
    x = x0.clone()
    out = ta.Optimize(x, model, opts, history=True)
    torch.cuda.synchronize()
    stop = out.stop_reason.cpu().numpy()
    iters = out.num_iters.cpu().numpy()
    assert (stop >= 0).all()                                                   # Succeeded() for all
    assert float((x - xstar).abs().max()) < tol_star                           # planted solution recovered
    # accepted costs are non-increasing
    errs, succ = out.errs.cpu().numpy(), out.successes.cpu().numpy().astype(bool)
    hs = errs.shape[1]
    valid = np.arange(hs)[None, :] < iters[:, None]
    acc = np.where(valid & (succ | (np.arange(hs)[None, :] == 0)), errs, np.inf)
    run_min = np.minimum.accumulate(acc, axis=1)
    for k in range(1, hs):
        col = acc[:, k]
        ok = ~np.isfinite(col) | (col <= run_min[:, k - 1])
        assert ok.all(), (k, np.flatnonzero(~ok)[:5])
    # counters: one linear solve per iteration, at least one data pass per iteration
    cnt = out.counters.cpu().numpy()
    assert cnt[3] == P and cnt[2] == iters.sum() and cnt[0] + cnt[1] >= iters.sum()

    # idempotence: restart from the solution
    x2 = x.clone()
    out2 = ta.Optimize(x2, model, opts)
    torch.cuda.synchronize()
    assert (out2.stop_reason.cpu().numpy() >= 0).all()
    it2 = out2.num_iters.cpu().numpy()     # at the fp32 round-off floor a few problems keep trading last bits
    assert np.median(it2) <= 4 and it2.max() <= opts.max_iters + 1
    assert float((x2 - x).abs().max()) < (1e-9 if dtype == np.float64 else 2e-4)
    # same point, cost re-evaluated (Gram pass vs cost-only pass round differently in fp32)
    rt = 1e-9 if dtype == np.float64 else 1e-3
    assert bool((out2.final_cost <= out.final_cost * (1 + rt) + 1e-12).all())

    # batch independence + sampled oracle parity
    for first in (0, P // 2 + 37, P - 72):
        S = 72   # > #CUs / 4 problems: stays on the fused path (a smaller batch of m >= 512 problems is row-split,
                 # which folds the rows in a different order and is compared with a tolerance in test_gpu_split.py)
        sub_model, sub_x0, _ = ta.DenseRow.synthetic(S, n, m, tdt, problem0=first)
        assert torch.equal(sub_x0, x0[first:first + S])
        xs = sub_x0.clone()
        sub = ta.Optimize(xs, sub_model, opts)
        torch.cuda.synchronize()
        assert torch.equal(xs, x[first:first + S])                              # bit-identical rows
        assert torch.equal(sub.stop_reason, out.stop_reason[first:first + S])
        assert torch.equal(sub.num_iters, out.num_iters[first:first + S])
        assert torch.equal(sub.final_cost, out.final_cost[first:first + S])
        K = 8    # oracle on the first K of them
        A, b, x0h, _ = oracle.synth_dense_row(K, n, m, dtype, problem0=first)
        ref = oracle.dense_row_lm(A, b, x0h, opts.to_pod())
        assert (ref["stop"] >= 0).all()
        assert np.abs(xs[:K].cpu().numpy() - ref["x"]).max() < tol_oracle
        if dtype == np.float64:
            assert np.array_equal(sub.stop_reason[:K].cpu().numpy(), ref["stop"])
            assert np.array_equal(sub.num_iters[:K].cpu().numpy(), ref["iters"])
            assert np.allclose(sub.final_cost[:K].cpu().numpy(), ref["cost"], rtol=1e-9)
