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

from parity import check_trajectories

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
    assert cnt[3] == P and cnt[2] == iters.sum() and cnt[0] + cnt[1] + cnt[4] >= iters.sum()   # [4]: Builds served from the memo

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

    # ---- sampled oracle parity on the headline dtype too: 3 x 176 = 528 problem ids cut out of the full
    # batch, the whole trajectory (cost, accept / reject, StopReason, iteration count) against the oracle's with the
    # tie-aware comparator, and the iteration-count DISTRIBUTION bounded: `value` of bench.py counts LM iterations, so
    # the device may not buy throughput with extra floor-thrashing iterations the reference algorithm would not make.
    K = 176
    it_gpu, it_ref, ties, full = [], [], 0, 0
    errs_all, succ_all = out.errs.cpu().numpy(), out.successes.cpu().numpy()
    fails_all, d2_all = out.num_failures.cpu().numpy(), out.deltas2.cpu().numpy()
    cost_all, x_all = out.final_cost.cpu().numpy(), x.cpu().numpy()
    for first in (0, P // 2 + 37, P - K):
        A, b, x0h, _ = oracle.synth_dense_row(K, n, m, dtype, problem0=first)
        ref = oracle.dense_row_lm(A, b, x0h, opts.to_pod(), history=True, nthreads=oracle.load().oracle_num_threads_max())
        assert (ref["stop"] >= 0).all()
        sl = slice(first, first + K)
        g = dict(errs=errs_all[sl], succ=succ_all[sl], iters=iters[sl], stop=stop[sl], x=x_all[sl], cost=cost_all[sl],
                 fails=fails_all[sl], deltas2=d2_all[sl])
        refd = dict(errs=ref["errs"], succ=ref["succ"], iters=ref["iters"], stop=ref["stop"], x=ref["x"], cost=ref["cost"],
                    fails=ref["fails"], deltas2=ref["deltas2"])
        st = check_trajectories(g, refd, dtype, opts.to_pod(), label=f"{tag} ids {first}..")
        assert all(j >= 2 for j in st["tie_iters"]), st       # the pre-floor prefix is identical for every problem
        ties += st["ties"]
        full += st["full"]
        it_gpu.append(iters[sl])
        it_ref.append(ref["iters"])
        assert np.abs(x_all[sl] - ref["x"]).max() < tol_oracle
    it_gpu, it_ref = np.concatenate(it_gpu), np.concatenate(it_ref)
    assert full + ties == 3 * K
    if dtype == np.float64:
        # C3 stops on |dx|^2 < min_step_norm2 before any cost comparison can tie: identical counts (a |dx|^2 that sits
        # on the threshold to round-off is the only way to part, and check_trajectories has proven any such case)
        assert np.array_equal(it_gpu, it_ref) or ties > 0
        assert abs(it_gpu.mean() - it_ref.mean()) <= 0.01
    else:
        # fp32: the device's blocked sums resolve slightly smaller cost decreases than the oracle's sequential float
        # sum, so it takes a few more last-bit steps; bound the mean (measured round 1: 7.44 vs 7.25 it/problem)
        # (measured: +0.19; the bound is on the DIRECTION too — the metric counts iterations, so the device must not buy
        # throughput with iterations the reference algorithm would not make — bench.py prints `value_at_oracle_iters`)
        # (tools/iter_inflation.py: the oracle itself takes 7.462 when ITS a_i.x is a 16-lane tree like the device's — the gap is
        #  the accuracy of that dot product, not floor-thrashing; profiles/r04_iter_inflation.txt)
        # (7.485 vs 7.295 on these 528 ids, the same since round 2 — the inputs and the kernels' sums are deterministic; the
        # bound is that measurement plus a margin for a kernel change that re-orders a sum, not a licence for more)
        assert -0.05 <= it_gpu.mean() - it_ref.mean() <= 0.22, (it_gpu.mean(), it_ref.mean())
        assert np.abs(it_gpu.astype(int) - it_ref.astype(int)).max() <= opts.max_iters + 1
    print(f"[{tag}] sampled parity: {full} identical to the end, {ties} proven ties; iterations/problem "
          f"GPU {it_gpu.mean():.3f} oracle {it_ref.mean():.3f}")
