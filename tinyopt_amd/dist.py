"""Multi-GPU sharding for batches of independent problems.

Problems are independent (the reference optimises exactly one x per call, docs/API.md:12), so the
batch shards embarrassingly: rank g owns the contiguous block [g*P/G, (g+1)*P/G) and there is NO
per-iteration communication.  Exactly one collective runs at the end — a gather of the results to
rank 0 (RCCL over xGMI when the backend is "nccl"; gloo on CPU for the tests).
"""
from __future__ import annotations

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
