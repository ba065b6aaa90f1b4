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
