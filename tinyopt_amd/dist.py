"""Multi-GPU sharding for batches of independent problems.

Problems are independent (the reference optimises exactly one x per call, docs/API.md:12), so the
batch shards embarrassingly: rank g owns the contiguous block [g*P/G, (g+1)*P/G) and there is NO
per-iteration communication.  Exactly one collective runs at the end — a gather of the results to
rank 0 (RCCL over xGMI when the backend is "nccl"; gloo on CPU for the tests).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(P: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition of problem ids; sizes differ by at most one."""
    base, rem = divmod(int(P), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def gather_output(x: torch.Tensor, fields: Dict[str, torch.Tensor], P_total: int, dst: int = 0,
                  group=None) -> Optional[Dict[str, torch.Tensor]]:
    """Gather per-problem results of every rank's shard to `dst`, in problem-id order.

    x: [P_local, n]; fields: name -> [P_local, ...] tensors (stop_reason, num_iters, final_cost ...).
    Everything is packed into ONE float64 payload so a single gather moves it (C4: ~2.7 MB/GPU).
    Returns the assembled dict on `dst`, None elsewhere.  Without an initialised process group
    (single process) it returns the inputs unchanged.
    """
    names = sorted(fields)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        out = {"x": x}
        out.update({k: fields[k] for k in names})
        return out
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = x.shape[1]
    cols = [x.to(torch.float64).reshape(x.shape[0], -1)] + [fields[k].to(torch.float64).reshape(x.shape[0], -1) for k in names]
    widths = [c.shape[1] for c in cols]
    payload = torch.cat(cols, dim=1).contiguous()
    W = payload.shape[1]
    # equal-size gather: pad every shard to the largest shard
    max_local = max(shard_range(P_total, r, world)[1] - shard_range(P_total, r, world)[0] for r in range(world))
    buf = torch.zeros(max_local, W, dtype=torch.float64, device=payload.device)
    buf[: payload.shape[0]] = payload
    gathered = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, gathered, dst=dst, group=group)
    if rank != dst:
        return None
    parts = []
    for r in range(world):
        lo, hi = shard_range(P_total, r, world)
        parts.append(gathered[r][: hi - lo])
    full = torch.cat(parts, dim=0)
    out, off = {}, 0
    out["x"] = full[:, off:off + widths[0]].to(x.dtype).reshape(P_total, n)
    off += widths[0]
    for k, w in zip(names, widths[1:]):
        t = full[:, off:off + w].to(fields[k].dtype)
        out[k] = t.reshape((P_total,) + tuple(fields[k].shape[1:]))
        off += w
    return out


class Communicator:
    """RCCL communicator of the C-ABI (`toa_comm_*`, include/tinyopt_amd.h) for the single end-of-job collective.
    One per process / GPU.  The 128-byte unique id is created by one rank and handed to the others by the HOST program;
    `from_torch` uses torch.distributed's broadcast for that (any backend), a C++ host would use MPI_Bcast."""

    def __init__(self, ctx, id_bytes: bytes, nranks: int, rank: int):
        from ._capi import check
        self.ctx, self.nranks, self.rank = ctx, int(nranks), int(rank)
        assert len(id_bytes) == 128
        self._h = C.c_void_p()
        buf = C.create_string_buffer(bytes(id_bytes), 128)
        check(ctx.lib.toa_comm_init_rank(ctx.h, buf, self.nranks, self.rank, C.byref(self._h)))

    @staticmethod
    def unique_id(lib=None) -> bytes:
        from . import _capi
        lib = lib or _capi.load()
        buf = C.create_string_buffer(128)
        _capi.check(lib.toa_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def from_torch(cls, ctx, group=None) -> "Communicator":
        if not (dist.is_available() and dist.is_initialized()):
            return cls(ctx, cls.unique_id(ctx.lib), 1, 0)
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        dev = torch.device("cuda", ctx.device) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        t = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            t.copy_(torch.frombuffer(bytearray(cls.unique_id(ctx.lib)), dtype=torch.uint8))
        dist.broadcast(t, src=0, group=group)
        return cls(ctx, bytes(t.cpu().numpy().tobytes()), world, rank)

    def close(self):
        if getattr(self, "_h", None):
            self.ctx.lib.toa_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def gather_native(comm: Communicator, x: torch.Tensor, out, P_total: int, root: int = 0):
    """`toa_gather`: ONE ncclGather of (x, stop_reason, num_iters, final_cost) in their native types, problem-id
    order on the root.  x: this rank's [P_local, xd] shard (P_local = size of shard_range(P_total, rank, nranks)); out:
    its Output.  Returns a dict of [P_total, ...] tensors on the root, None elsewhere.  Stream-ordered."""
    from ._capi import ToaResults, check
    from .api import _dtype_code
    ctx = comm.ctx
    lo, hi = shard_range(P_total, comm.rank, comm.nranks)
    if x.shape[0] != hi - lo:
        raise ValueError("x does not hold this rank's shard of P_total problems")
    loc = ToaResults()
    loc.stop_reason, loc.num_iters, loc.final_cost = out.stop_reason.data_ptr(), out.num_iters.data_ptr(), out.final_cost.data_ptr()
    res, allr, x_all = None, ToaResults(), None
    if comm.rank == root:
        dev = x.device
        x_all = torch.empty(P_total, x.shape[1], dtype=x.dtype, device=dev)
        res = {"x": x_all, "stop_reason": torch.empty(P_total, dtype=torch.int32, device=dev),
               "num_iters": torch.empty(P_total, dtype=torch.int32, device=dev),
               "final_cost": torch.empty(P_total, dtype=torch.float64, device=dev)}
        allr.stop_reason, allr.num_iters, allr.final_cost = (res["stop_reason"].data_ptr(), res["num_iters"].data_ptr(),
                                                             res["final_cost"].data_ptr())
    check(ctx.lib.toa_gather(ctx.h, comm._h, _dtype_code(x.dtype), x.shape[1], int(P_total), x.data_ptr(), C.byref(loc), int(root),
                             x_all.data_ptr() if x_all is not None else None, C.byref(allr)))
    return res
