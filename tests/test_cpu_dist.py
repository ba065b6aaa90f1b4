"""CPU tests of the N>1 path: world_size-2 gloo processes exercising shard_range + the single result
gather (the only collective on the path)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    from tinyopt_amd.dist import shard_range
    for P in (0, 1, 7, 8, 100000, 12501):
        for world in (1, 2, 3, 8):
            spans = [shard_range(P, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == P
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, P, n, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tinyopt_amd.dist import gather_output, shard_range
    lo, hi = shard_range(P, rank, world)
    ids = torch.arange(lo, hi, dtype=torch.float64)
    x = (ids[:, None] * 10 + torch.arange(n, dtype=torch.float64)[None, :]).to(torch.float32)
    fields = {"stop_reason": (ids % 5).to(torch.int32), "num_iters": (ids * 2).to(torch.int32), "final_cost": ids * 0.5}
    out = gather_output(x, fields, P_total=P)
    if rank == 0:
        q.put({k: v.numpy() for k, v in out.items()})
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_gather_world2_gloo():
    P, n, world = 11, 3, 2   # uneven shards: 6 + 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, world, port, P, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ids = np.arange(P, dtype=np.float64)
    assert np.array_equal(got["x"], (ids[:, None] * 10 + np.arange(n)[None, :]).astype(np.float32))
    assert np.array_equal(got["stop_reason"], (ids % 5).astype(np.int32))
    assert np.array_equal(got["num_iters"], (ids * 2).astype(np.int32))
    assert np.array_equal(got["final_cost"], ids * 0.5)


def test_gather_world4_uneven_gloo():
    """Four ranks, shards of 4 + 4 + 3 + 3 problems and a wide x (the 2 / 4 / 8-GPU legs of the scaling run use exactly this
    path when the native gather is not available): the root gets every field in problem-id order."""
    P, n, world = 14, 50, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29850 + (os.getpid() % 100)
    procs = [ctx.Process(target=_worker, args=(r, world, port, P, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    ids = np.arange(P, dtype=np.float64)
    assert np.array_equal(got["x"], (ids[:, None] * 10 + np.arange(n)[None, :]).astype(np.float32))
    assert np.array_equal(got["stop_reason"], (ids % 5).astype(np.int32))
    assert np.array_equal(got["num_iters"], (ids * 2).astype(np.int32))
    assert np.array_equal(got["final_cost"], ids * 0.5)


def test_bench_watchdog_line_is_a_complete_bench_line():
    """bench.py at N > 1 arms a watchdog that prints the measured line if the result gather hangs: that line must carry the
    same top-level keys as the normal one — `roofline` included, or the driver would record the scaling point as unmeasured."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    a = src.index("partial = {")
    b = src.index("def _bail():")
    block = src[a:b]
    for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"steps"', '"warmup"', '"ms_per_step"', '"higher_is_better"', '"scaling"',
                '"vs_baseline"', '"dtype"', '"data"', '"config"', '"roofline"', '"frac"', '"achieved"', '"peak"'):
        assert key in block, key


def test_gather_single_process_passthrough():
    from tinyopt_amd.dist import gather_output
    x = torch.zeros(4, 2)
    out = gather_output(x, {"stop_reason": torch.ones(4, dtype=torch.int32)}, P_total=4)
    assert out["x"] is x and out["stop_reason"].sum() == 4


def test_c_abi_shard_range_matches_python(built):
    """toa_shard_range (the C-ABI's block partition, used by toa_gather's unpack) == tinyopt_amd.dist.shard_range."""
    import ctypes as C
    from tinyopt_amd import _capi
    from tinyopt_amd.dist import shard_range
    lib = _capi.load()
    lo, hi = C.c_int64(), C.c_int64()
    for P in (0, 1, 7, 8, 100000, 12501):
        for world in (1, 2, 3, 8):
            for r in range(world):
                assert lib.toa_shard_range(P, r, world, C.byref(lo), C.byref(hi)) == 0
                assert (lo.value, hi.value) == shard_range(P, r, world)
    assert lib.toa_shard_range(5, 2, 2, C.byref(lo), C.byref(hi)) != 0


def _dry_run(world):
    """`bench.py --gpus N` launched EXACTLY as the driver launches the scaling bench (python -m torch.distributed.run ...), with
    --dry-run: gloo + CPU tensors and a stand-in for the solve through the real barriers, MAX / SUM / MIN reductions, result
    gather, watchdog and JSON assembly.  (no multi-GPU box has ever run this code; a shape bug in the N > 1
    plumbing must not be what the first 8-GPU run finds.)"""
    import json
    import subprocess
    port = 29950 + (os.getpid() % 40) + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--dry-run"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout            # rank 0 prints ONE line
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == world and d["steps"] == 3 and d["scaling"] == "weak" and "dry-run" in d["data"]
    P = d["config"]["problems_per_gpu"]
    # the stand-in solves take 7 iterations per problem on even ranks and 8 on odd ones: SUM and MIN / MAX over ranks
    per_rank = [P * (7 + r % 2) for r in range(world)]
    assert d["config"]["lm_iterations_per_step_all_gpus"] == sum(per_rank)
    assert d["config"]["lm_iterations_per_step_per_gpu_min_max"] == [min(per_rank), max(per_rank)]
    assert abs(d["config"]["iters_per_problem"] - sum(per_rank) / (P * world)) < 1e-12
    assert d["value"] > 0 and abs(d["value"] - sum(per_rank) * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-9
    assert "torch.distributed.gather" in d["config"]["gather_impl"] and d["config"]["gather_ms"] > 0
    assert d["roofline"]["unit"] == "GB/s" and d["roofline"]["passes_per_launch"] == 7 * P
    return d


def test_bench_dry_run_two_ranks():
    _dry_run(2)


def test_bench_dry_run_eight_ranks():
    _dry_run(8)
