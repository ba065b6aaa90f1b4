"""Team form of the fused kernel (csrc/kernels.hpp DenseRowModel<.., TEAMW = 12>, toa_tuning::team_on): a workgroup
of twelve waves of which only `team_owners` pull problems; the others take ticketed row chunks of the owners' accumulate AND
evaluate-only passes, so that at most 2 problems per compute unit are in flight and their rows stay in the Infinity Cache
between two passes.  It is an A/B arm (measured slower: profiles/r05_ab_log.md §1), but it is a product path and must give the
bits of the classic one-wavefront-per-problem form: which wave computes a chunk never changes the chunk, and the fold order is
the ticket order on both sides.  What must hold, with the same chunk count on both sides:
  * x, StopReason, iteration counts, cost / |dx|^2 / accept histories, the exported Hessian and the pass counters are identical,
    for every number of owners (1 = one problem per compute unit ... 12 = every wave an owner, helpers only in the tail);
  * also with the memo switched off, with option sets that keep problems bouncing between rejected steps (evaluate-only
    passes: the second ticket counter), and for batches smaller than the launch (waves that help from the first microsecond)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(ta, model, x0, opts, **tune):
    from tinyopt_amd.api import default_context
    x = x0.clone()
    with default_context().tuning(**tune):
        out = ta.Optimize(x, model, opts, history=True)
        torch.cuda.synchronize()
    return x, out


def _same(a, b):
    xa, oa = a
    xb, ob = b
    assert torch.equal(xa, xb)
    for f in ("stop_reason", "num_iters", "final_cost", "errs", "deltas2", "successes", "num_failures", "counters"):
        ta_, tb_ = getattr(oa, f), getattr(ob, f)
        assert (ta_ is None) == (tb_ is None), f
        if ta_ is not None:
            assert torch.equal(ta_, tb_), f
    if oa.final_hessian is not None:
        assert torch.equal(oa.final_hessian, ob.final_hessian)


@pytest.mark.parametrize("owners,K", [(2, 6), (1, 6), (3, 4), (2, 2), (12, 2), (2, 12), (5, 7)])
def test_team_form_gives_the_bits_of_the_classic_form(ta, owners, K):
    P, n, m = 700, 50, 2000
    model, x0, _ = ta.DenseRow.synthetic(P, n, m, torch.float32)
    opts = ta.Options.benchmark()
    ref = _run(ta, model, x0, opts, coop_chunks=K)
    got = _run(ta, model, x0, opts, coop_chunks=K, team_on=1, team_owners=owners)
    _same(ref, got)
    assert int(got[1].counters[1]) > 0            # evaluate-only passes happened (ticketed through the second counter)
    got2 = _run(ta, model, x0, opts, coop_chunks=K, team_on=1, team_owners=owners, team_prio=1)
    _same(ref, got2)


def test_team_form_with_rejected_steps_and_without_the_memo(ta):
    """Far starts + a large initial damping factor: many rejected steps, roll-backs, evaluate-only iterations."""
    P, n, m = 300, 50, 1600
    model, x0, _ = ta.DenseRow.synthetic(P, n, m, torch.float32)
    g = torch.Generator(device="cpu").manual_seed(5)
    x0 = x0 + 3.0 * torch.randn(x0.shape, generator=g).to(x0)
    opts = ta.Options.benchmark()
    opts.max_iters = 30
    opts.max_consec_failures = 0
    opts.max_total_failures = 0
    for memo_off in (0, 1):
        ref = _run(ta, model, x0, opts, coop_chunks=5, memo_off=memo_off)
        got = _run(ta, model, x0, opts, coop_chunks=5, memo_off=memo_off, team_on=1, team_owners=2)
        _same(ref, got)
    assert int(ref[1].counters[1]) > P // 2      # plenty of evaluate-only passes
    assert int((ref[1].num_failures > 0).sum()) > P // 2


def test_team_form_small_batches_and_positions(ta):
    n, m = 50, 2000
    opts = ta.Options.benchmark()
    full, x0, _ = ta.DenseRow.synthetic(400, n, m, torch.float32)
    ref = _run(ta, full, x0, opts, coop_chunks=6)
    for first, S in ((0, 65), (7, 70), (100, 130), (301, 99)):   # (below 65 problems the library picks the row-split form: other sums)
        sub, sx0, _ = ta.DenseRow.synthetic(S, n, m, torch.float32, problem0=first)
        x_s, o_s = _run(ta, sub, sx0, opts, coop_chunks=6, team_on=1, team_owners=2)
        assert torch.equal(x_s, ref[0][first:first + S])
        assert torch.equal(o_s.errs, ref[1].errs[first:first + S]) and torch.equal(o_s.stop_reason, ref[1].stop_reason[first:first + S])
        assert torch.equal(o_s.num_iters, ref[1].num_iters[first:first + S])


def test_team_form_is_ignored_where_it_does_not_exist(ta):
    """Other layouts / dtypes have no team instantiation: toa_tuning::team_on changes nothing there."""
    for tdt, n, m in ((torch.float64, 12, 1500), (torch.float32, 34, 1500)):
        model, x0, _ = ta.DenseRow.synthetic(64, n, m, tdt)
        opts = ta.Options.benchmark()
        _same(_run(ta, model, x0, opts), _run(ta, model, x0, opts, team_on=1))
