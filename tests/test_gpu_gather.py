"""C1 at the boundary: `toa_comm_*` + `toa_gather` (one RCCL ncclGather of x / stop_reason / num_iters / final_cost in
native types).  A single-GPU box can only run the one-rank communicator; the multi-rank unpack arithmetic (block
partition, padding to the largest shard) is exercised by feeding the unpack path a hand-built multi-rank layout through
world-size-1 gathers of every shard and by the CPU test of toa_shard_range."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tdt", [torch.float32, torch.float64])
def test_gather_one_rank_roundtrip(ta, tdt):
    P, n, m = 37, 12, 60
    model, x0, _ = ta.DenseRow.synthetic(P, n, m, tdt)
    x = x0.clone()
    out = ta.Optimize(x, model, ta.Options.benchmark())
    ctx = ta.api.default_context()
    comm = ta.Communicator.from_torch(ctx)             # no process group: a one-rank communicator, id made locally
    assert (comm.nranks, comm.rank) == (1, 0)
    res = ta.gather_native(comm, x, out, P_total=P)
    torch.cuda.synchronize()
    assert torch.equal(res["x"], x) and res["x"].dtype == tdt            # native dtype, not a float64 payload
    assert torch.equal(res["stop_reason"], out.stop_reason) and torch.equal(res["num_iters"], out.num_iters)
    assert torch.equal(res["final_cost"], out.final_cost)
    res2 = ta.gather_native(comm, x, out, P_total=P)                      # the communicator and its buffers are reusable
    torch.cuda.synchronize()
    assert torch.equal(res2["x"], x)
    with pytest.raises(ValueError):
        ta.gather_native(comm, x[:5], out, P_total=P)
    comm.close()


def test_gather_se3_xdim_and_errors(ta, oracle):
    """x of a manifold model has 12 stored scalars per problem (n = 6): xdim, not n, travels."""
    data, p0, _ = oracle.synth_se3_reproj(3, 50, np.float64, seed=1)
    model = ta.SE3Reproj(torch.from_numpy(data).cuda(), 50)
    x = torch.from_numpy(p0.copy()).cuda()
    out = ta.Optimize(x, model, ta.Options())
    comm = ta.Communicator.from_torch(ta.api.default_context())
    res = ta.gather_native(comm, x, out, P_total=3)
    torch.cuda.synchronize()
    assert res["x"].shape == (3, 12) and torch.equal(res["x"], x)
    with pytest.raises(ta.ToaError):
        ta.gather_native(comm, x, out, P_total=3, root=1)                 # root out of range


@pytest.mark.parametrize("xdim,tdt", [(200, torch.float32), (1024, torch.float64), (12 * 8 + 3 * 256, torch.float64), (65, torch.float32)])
def test_gather_wide_parameter_blocks(ta, xdim, tdt):
    """DenseRowNatural (n up to 1024) and bundle adjustment (12 C + 3 N stored scalars) have x far wider than one wavefront:
    the record layout is generic in xdim (the round-2 limit of 64 is gone)."""
    from types import SimpleNamespace
    P = 23
    g = torch.Generator(device="cpu").manual_seed(xdim)
    x = torch.randn(P, xdim, generator=g, dtype=tdt).cuda()
    out = SimpleNamespace(stop_reason=torch.randint(-4, 9, (P,), generator=g, dtype=torch.int32).cuda(),
                          num_iters=torch.randint(0, 50, (P,), generator=g, dtype=torch.int32).cuda(),
                          final_cost=torch.rand(P, generator=g, dtype=torch.float64).cuda())
    comm = ta.Communicator.from_torch(ta.api.default_context())
    res = ta.gather_native(comm, x, out, P_total=P)
    torch.cuda.synchronize()
    assert torch.equal(res["x"], x)
    assert torch.equal(res["stop_reason"], out.stop_reason) and torch.equal(res["num_iters"], out.num_iters)
    assert torch.equal(res["final_cost"], out.final_cost)
    comm.close()
