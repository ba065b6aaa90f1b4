"""The fused solve is one asynchronous launch on the handle's stream (include/tinyopt_amd.h, toa_lm_run), so it can be
captured into a hipGraph and replayed — provided its workspaces (parameter block, memo slots) exist: they are grown on
demand, which needs a stream drain and a hipMalloc, neither legal under capture.  A first un-captured call of the shape
makes them; a capture that would have to grow one is refused with a message that says so."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _problem(ta, P, n, m, dtype):  # noqa: D103
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    return ta.DenseRow.synthetic(P, n, m, tdt)


def test_capture_after_a_warm_call_replays_the_solve(ta):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        model, x0, xstar = _problem(ta, 600, 50, 400, np.float32)
        opts = ta.Options.benchmark()
        x_ref = x0.clone()
        ref = ta.Optimize(x_ref, model, opts)            # also the warm call: the context of this stream, its workspaces
        s.synchronize()
        x = x0.clone()
        out = ta.Optimize(x, model, opts)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            ta.Optimize(x, model, opts, out=out)
        for _ in range(2):
            x.copy_(x0)
            out.num_iters.zero_()
            g.replay()
            s.synchronize()
            assert torch.equal(x, x_ref)
            assert torch.equal(out.num_iters, ref.num_iters) and torch.equal(out.stop_reason, ref.stop_reason)
            assert torch.equal(out.final_cost, ref.final_cost)


def test_capture_that_would_grow_a_workspace_is_refused(ta):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        small, x0s, _ = _problem(ta, 8, 50, 64, np.float32)
        opts = ta.Options.benchmark()
        ta.Optimize(x0s.clone(), small, opts)            # the context exists, with workspaces for 8 problems
        big, x0, _ = _problem(ta, 4000, 50, 64, np.float32)
        x = x0.clone()
        out = ta.api._alloc_output(4000, 50, opts, False, x.device)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with pytest.raises(Exception, match="cannot happen while the stream is being captured"):
            with torch.cuda.graph(g, stream=s):
                ta.Optimize(x, big, opts, out=out)
    torch.cuda.synchronize()
    # the handle is still usable afterwards
    with torch.cuda.stream(s):
        o2 = ta.Optimize(x, big, opts)
        s.synchronize()
        assert bool((o2.stop_reason > 0).all())


def test_the_n_256_pipeline_is_capturable_where_every_stage_is_ours(ta, oracle):
    """fp32, aligned rows, n = 256: rows kernel, Gram, factorisation and the state machine are all kernels of this library
    that skip finished problems, so under capture the whole pass budget is recorded and the graph replays the solve with no
    host in the loop — the bits of the eager call.  fp64 (library Gram) says it cannot."""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        P, n, m = 6, 256, 768
        A, b, x0h, _ = oracle.synth_dense_row(P, n, m, np.float32, seed=3)
        model = ta.DenseRowNatural(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
        x0 = torch.from_numpy(x0h).cuda()
        opts = ta.Options.benchmark()
        x_ref = x0.clone()
        ref = ta.Optimize(x_ref, model, opts)            # eager (and the warm call)
        x = x0.clone()
        out = ta.Optimize(x, model, opts)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            ta.Optimize(x, model, opts, out=out)
        for _ in range(2):
            x.copy_(x0)
            out.num_iters.zero_()
            g.replay()
            s.synchronize()
            assert torch.equal(x, x_ref)
            assert torch.equal(out.num_iters, ref.num_iters) and torch.equal(out.stop_reason, ref.stop_reason)
            assert torch.equal(out.final_cost, ref.final_cost)
        A64, b64, x064, _ = oracle.synth_dense_row(2, 160, 480, np.float64, seed=4)
        m64 = ta.DenseRowNatural(torch.from_numpy(A64).cuda(), torch.from_numpy(b64).cuda())
        x64 = torch.from_numpy(x064).cuda()
        o64 = ta.Optimize(x64.clone(), m64, opts)
        s.synchronize()
        g2 = torch.cuda.CUDAGraph()
        with pytest.raises(Exception, match="can be captured"):
            with torch.cuda.graph(g2, stream=s):
                ta.Optimize(x64, m64, opts, out=o64)
    torch.cuda.synchronize()


def test_a_graph_survives_a_later_larger_eager_call(ta):
    """a captured launch bakes the pointers of the handle's workspaces (memo slots, scratch) into the graph; a later
    eager call with a larger shape makes them grow.  Once a launch has been captured the outgrown block is kept until toa_destroy
    (toa_release_workspace), so the earlier graph still replays — and still gives the bits of its eager twin."""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        opts = ta.Options.benchmark()
        # (a) the fused kernel: memo slots for 40 problems' waves, then a batch that needs every resident slot
        model, x0, _ = _problem(ta, 40, 50, 1100, np.float32)
        x_ref = x0.clone()
        ref = ta.Optimize(x_ref, model, opts)
        x = x0.clone()
        out = ta.Optimize(x, model, opts)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            ta.Optimize(x, model, opts, out=out)
        big, xb, _ = _problem(ta, 5000, 50, 1100, np.float32)
        ob = ta.Optimize(xb.clone(), big, opts)            # the memo block grows here
        s.synchronize()
        assert bool((ob.stop_reason > 0).all())
        for _ in range(2):
            x.copy_(x0)
            out.num_iters.zero_()
            g.replay()
            s.synchronize()
            assert torch.equal(x, x_ref) and torch.equal(out.num_iters, ref.num_iters) and torch.equal(out.final_cost, ref.final_cost)
        # (b) the n > 128 pipeline: its scratch block (rows, Gram partials, factor workspace) outgrown by a larger batch
        rng = np.random.default_rng(7)
        P, n, m = 4, 160, 640
        A = rng.uniform(-1, 1, (P, m, n)).astype(np.float32)
        xs = rng.uniform(-1, 1, (P, n)).astype(np.float32)
        t = np.einsum("pmn,pn->pm", A, xs)
        b = (t + 0.1 * np.sin(t)).astype(np.float32)
        x0n = torch.from_numpy(xs + 0.3 * rng.uniform(-1, 1, (P, n)).astype(np.float32)).cuda()
        mdl = ta.DenseRowNatural(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
        xr = x0n.clone()
        refn = ta.Optimize(xr, mdl, opts)
        xn = x0n.clone()
        outn = ta.Optimize(xn, mdl, opts)
        s.synchronize()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=s):
            ta.Optimize(xn, mdl, opts, out=outn)
        A2 = rng.uniform(-1, 1, (24, 2 * m, n)).astype(np.float32)
        b2 = rng.uniform(-1, 1, (24, 2 * m)).astype(np.float32)
        big2 = ta.DenseRowNatural(torch.from_numpy(A2).cuda(), torch.from_numpy(b2).cuda())
        ta.Optimize(torch.zeros(24, n, device="cuda"), big2, opts)          # scratch grows
        s.synchronize()
        for _ in range(2):
            xn.copy_(x0n)
            outn.num_iters.zero_()
            g2.replay()
            s.synchronize()
            assert torch.equal(xn, xr) and torch.equal(outn.num_iters, refn.num_iters) and torch.equal(outn.final_cost, refn.final_cost)
    torch.cuda.synchronize()


def test_capture_of_an_unbounded_retry_budget_is_refused_with_the_numbers(ta):
    """max_consec_failures == 0 means up to 255 retries per iteration — (max_iters + 2) * 256 passes of ~7 kernels.
    The n > 128 capture records its whole budget, so such options are refused under capture (the eager call runs them)."""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        rng = np.random.default_rng(8)
        P, n, m = 2, 160, 480
        A = rng.uniform(-1, 1, (P, m, n)).astype(np.float32)
        b = rng.uniform(-1, 1, (P, m)).astype(np.float32)
        mdl = ta.DenseRowNatural(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
        opts = ta.Options.benchmark()
        opts.max_consec_failures = 0
        x = torch.zeros(P, n, device="cuda")
        out = ta.Optimize(x, mdl, opts)                   # eager: fine
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with pytest.raises(Exception, match="graph nodes"):
            with torch.cuda.graph(g, stream=s):
                ta.Optimize(x, mdl, opts, out=out)
    torch.cuda.synchronize()


def test_a_run_time_models_second_build_is_refused_under_capture_and_fine_after_a_warm_call(ta):
    """Round 6: a run-time model is built twice — without the M-estimator branch at toa_model_compile, with it the first time it runs on
    a handle that has a loss set.  That second build cannot happen inside a stream capture (TOA_E_UNSUPPORTED, nothing recorded, the
    capture left valid); after one un-captured call the same solve captures and replays."""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        P, items, n = 512, 200, 6
        gen = torch.Generator(device="cuda").manual_seed(3)
        A = torch.rand(P, items, n, dtype=torch.float64, device="cuda", generator=gen) * 2 - 1
        xs = torch.rand(P, n, dtype=torch.float64, device="cuda", generator=gen) * 2 - 1
        t = torch.einsum("pmn,pn->pm", A, xs)
        data = torch.cat([A, (t + 0.1 * torch.sin(t))[..., None]], 2).contiguous()
        x0 = xs + 0.2
        body = "S t = x[0] * p[0];\n" + "".join(f"t = t + x[{j}] * p[{j}];\n" for j in range(1, n)) + f"r[0] = t + T(0.1) * sin(t) - p[{n}];"
        jit = ta.JitResidual(body + "  // capture test: a text of its own, so that no other test has built its variants", n=n, item_scalars=n + 1, dtype=torch.float64)
        opts = ta.Options.benchmark()
        plain = jit.bind(data)
        robust = plain.with_loss("huber", 0.5)
        x = x0.clone()
        out = ta.Optimize(x, plain, opts)                 # the plain build runs (and warms the context of this stream)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with pytest.raises(Exception, match="cannot happen while the stream is being captured"):
            with torch.cuda.graph(g, stream=s):
                ta.Optimize(x, robust, opts, out=out)     # would need the second build
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        x_ref = x0.clone()
        ref = ta.Optimize(x_ref, robust, opts)            # un-captured: builds the variant with the branch
        s.synchronize()
        x.copy_(x0)
        out = ta.Optimize(x, robust, opts)
        s.synchronize()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=s):
            ta.Optimize(x, robust, opts, out=out)
        x.copy_(x0)
        g2.replay()
        s.synchronize()
        assert torch.equal(x, x_ref) and torch.equal(out.num_iters, ref.num_iters)
