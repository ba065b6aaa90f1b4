"""Cooperative passes (csrc/models_dense.hpp CoopCtl / DenseRowModel::coop_pass): in the fused kernel a data pass of a problem with
m >= 1024 rows is K ticketed row chunks, each accumulated from zero and folded into the owner's LDS total in TICKET ORDER; a
wave whose work queue is dry takes tickets of its workgroup siblings' passes.  What must hold:
  * whoever computes the chunks, the bits are the same: run-to-run, for any batch size, for any position in the batch —
    also when most waves of a launch have nothing of their own and help from the first microsecond;
  * the chunked sum is a legitimate evaluation of the same Accumulate callback: trajectories equal the oracle's (tie-aware),
    for every chunk count, exactly as with the single-chunk pass (toa_tuning::coop_off)."""
import numpy as np
import pytest
import torch

from parity import check_trajectories, gpu_dict

pytestmark = pytest.mark.gpu


def _run(ta, model, x0, opts):
    x = x0.clone()
    out = ta.Optimize(x, model, opts, history=True)
    torch.cuda.synchronize()
    return x, out


def _bits(t):
    return t.cpu().numpy().tobytes()


@pytest.mark.parametrize("tdt,n,m", [(torch.float32, 50, 2000), (torch.float64, 50, 1200), (torch.float32, 63, 1024), (torch.float32, 34, 1500)])
def test_bits_do_not_depend_on_who_computes_the_chunks(ta, tdt, n, m):
    P = 333   # 84 workgroups: every workgroup is in its tail from the first problems on
    model, x0, _ = ta.DenseRow.synthetic(P, n, m, tdt)
    opts = ta.Options.benchmark()
    x_a, o_a = _run(ta, model, x0, opts)
    for _ in range(3):                                     # run to run
        x_b, o_b = _run(ta, model, x0, opts)
        assert _bits(x_a) == _bits(x_b) and _bits(o_a.errs) == _bits(o_b.errs) and _bits(o_a.num_iters) == _bits(o_b.num_iters)
    for first, S in ((0, 65), (100, 97), (P - 72, 72)):     # batch size / position (65 problems: 3 waves help from the start)
        sub, sx0, _ = ta.DenseRow.synthetic(S, n, m, tdt, problem0=first)
        assert torch.equal(sx0, x0[first:first + S])
        x_s, o_s = _run(ta, sub, sx0, opts)
        assert torch.equal(x_s, x_a[first:first + S])
        assert torch.equal(o_s.errs, o_a.errs[first:first + S]) and torch.equal(o_s.final_cost, o_a.final_cost[first:first + S])
        assert torch.equal(o_s.stop_reason, o_a.stop_reason[first:first + S])


@pytest.mark.parametrize("K", [0, 2, 4, 7])
def test_every_chunk_count_follows_the_oracle(ta, oracle, K):
    P, n, m = 96, 50, 2000
    A, b, x0, _ = oracle.synth_dense_row(P, n, m, np.float32, seed=123)
    opts = ta.Options.benchmark()
    ref = oracle.dense_row_lm(A, b, x0, opts.to_pod(), history=True)
    model = ta.DenseRow.from_arrays(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
    with ta.api.default_context().tuning(coop_off=int(K == 0), coop_chunks=K):       # toa_tuning: typed per-handle state, no environment
        x, out = _run(ta, model, torch.from_numpy(x0).cuda(), opts)
    check_trajectories(gpu_dict(out, x), dict(errs=ref["errs"], succ=ref["succ"], iters=ref["iters"], stop=ref["stop"], x=ref["x"],
                                              cost=ref["cost"], fails=ref["fails"], deltas2=ref["deltas2"]), np.float32, opts.to_pod())


def test_single_chunk_pass_is_the_classic_pass(ta):
    """Below 1024 rows (and with toa_tuning::coop_off) a pass is ONE chunk: the same instruction stream over the same rows as the
    launch-per-iteration form's data pass, so the stepping form still reproduces the fused kernel bit for bit."""
    P, n, m = 80, 50, 600
    model, x0, _ = ta.DenseRow.synthetic(P, n, m, torch.float32)
    opts = ta.Options.benchmark()
    x_f, o_f = _run(ta, model, x0, opts)
    xs = x0.clone()
    o_s = ta.Optimizer(xs, model, opts, history=True)()
    torch.cuda.synchronize()
    assert torch.equal(xs, x_f) and torch.equal(o_s.num_iters, o_f.num_iters) and torch.equal(o_s.final_cost, o_f.final_cost)
