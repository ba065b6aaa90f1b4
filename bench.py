#!/usr/bin/env python3
"""bench.py — LM iterations/s of the batched dense LM hot path on N MI355X (one process per GPU).

A "step" = one pass of the hot path over one batch: a complete batched LM solve (`toa_lm_run`,
benchmarks/options.h options) of this rank's shard of synthetic DenseRow problems, restarted from the
same x0 every step.  Inputs are generated in HBM before the timed region.  Default workload = the
per-GPU shard of BASELINE config C4 (12 500 problems x n=50 x m=2000, fp32; 8 ranks = the full
100 000-problem C4) — weak scaling.  Problems shard across ranks with no data-path collective; one
result gather (RCCL) runs after the timed region and is reported separately.

Prints ONE JSON line on rank 0 (stdout); diagnostics go to stderr.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (P per GPU, n, m, torch dtype, dtype tag, description)
    "c4": (12500, 50, 2000, torch.float32, "f32", "C4 shard: 12500 problems/GPU x n=50 x m=2000 DenseRow fp32 (8 GPUs = 100k-problem C4)"),
    "c3": (10000, 12, 500, torch.float64, "f64", "C3: 10000 problems/GPU x n=12 x m=500 DenseRow fp64"),
    # beyond one wavefront: rows kernel + batched library GEMM + workgroup Cholesky; MFMA-bound
    "large128": (512, 128, 4096, torch.float32, "f32", "512 problems/GPU x n=128 x m=4096 DenseRow fp32, library-backed path (n > 63)"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# dense MFMA peaks of the dtypes this path computes in: f32-input MFMA 157.3 TF (MI355X_MICROARCH.md), f64 78.6 TF (vendor spec)
MFMA_PEAK_TFLOPS = {"f32": 157.3, "f64": 78.6}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(n, m, np_dtype, pod, budget_problems):
    """Oracle (CPU restatement of the reference algorithm) timed on this host, single thread, on a
    bounded sample of the same workload.  kind = "port" (the real reference needs Eigen; not buildable)."""
    from oracle import pyoracle
    lib = None
    try:  # host-tuned build so the baseline is not handicapped by the portable -march of the shipped .so
        lib = pyoracle.load(pyoracle.build(march="native", out_dir=tempfile.mkdtemp(prefix="toa_oracle_")))
        build = "g++ -O3 -march=native"
    except Exception as e:  # noqa: BLE001
        log(f"[cpu_baseline] native rebuild failed ({e}); using shipped x86-64-v3 build")
        lib = pyoracle.load()
        build = "g++ -O3 -march=x86-64-v3"
    A, b, x0, _ = pyoracle.synth_dense_row(budget_problems, n, m, np_dtype)
    r1 = pyoracle.dense_row_lm(A, b, x0, pod, nthreads=1, lib=lib)
    it1 = int(r1["iters"].sum())
    ncores = os.cpu_count() or 1
    rall = pyoracle.dense_row_lm(A, b, x0, pod, nthreads=ncores, lib=lib)
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": it1 / r1["seconds"], "unit": "LM iterations/s", "cores": 1, "kind": "port",
        "sample": f"{budget_problems} problems of the same workload (n={n}, m={m}), {it1} LM iterations, "
                  f"{r1['seconds']:.1f} s single thread; oracle/lm_oracle.hpp built with {build}",
        "all_cores": {"value": int(rall["iters"].sum()) / rall["seconds"], "cores": ncores, "seconds": rall["seconds"]},
        "host_cpu": model,
    }, r1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--problems", type=int, default=0, help="override problems per GPU (debug; invalidates the metric)")
    ap.add_argument("--cpu-problems", type=int, default=0, help="CPU baseline sample size (0 = auto)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg (profiling runs)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    import tinyopt_amd as ta

    P, n, m, tdt, tag, desc = WORKLOADS[args.workload]
    if args.problems:
        P = args.problems
    ctx = ta.api.default_context(local_rank)
    info = ctx.info()
    opts = ta.Options.benchmark()  # benchmarks/options.h:10-27
    pod = opts.to_pod()

    # ---- inputs resident in HBM before the timed region (weak scaling: rank r owns problems [r*P, (r+1)*P))
    large = n > 63
    if not large:
        model, x0, xstar = ta.DenseRow.synthetic(P, n, m, tdt, problem0=rank * P)
    else:  # natural layout (A then b), generated on the device with the same distributions
        gen = torch.Generator(device="cuda").manual_seed(0x7194 + rank)
        A = torch.rand(P, m, n, dtype=tdt, device="cuda", generator=gen) * 2 - 1
        xstar = torch.rand(P, n, dtype=tdt, device="cuda", generator=gen) * 2 - 1
        t = torch.einsum("pmn,pn->pm", A, xstar)
        bvec = t + 0.1 * torch.sin(t) + 1e-3 * (torch.rand(P, m, dtype=tdt, device="cuda", generator=gen) * 2 - 1)
        x0 = xstar + 0.5 * (torch.rand(P, n, dtype=tdt, device="cuda", generator=gen) * 2 - 1) / (n / 50) ** 0.5
        model = ta.DenseRowNatural(A, bvec)
        del A, bvec, t
    x = x0.clone()
    out = ta.Optimize(x, model, opts)  # allocates result buffers once
    torch.cuda.synchronize()

    def step(acc):
        """One timed unit: restart from x0, run the batched solve, accumulate the step's units on the
        device (no host sync).  Used identically for warmup and timing so that every lazily loaded
        torch kernel (copy, sum, add) is resident before the clock starts."""
        x.copy_(x0)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        ta.Optimize(x, model, opts, out=out)  # one kernel launch on torch's current stream
        e1.record()
        it = out.num_iters.sum(dtype=torch.int64)
        ps = out.counters[0] + out.counters[1]
        ap = out.counters[0].clone()
        if acc is None:
            return (it, ps, [(e0, e1)], ap)
        return (acc[0] + it, acc[1] + ps, acc[2] + [(e0, e1)], acc[3] + ap)

    wacc = None
    for _ in range(max(args.warmup, 1) if args.warmup else 0):
        wacc = step(wacc)
    if wacc is not None:
        _ = int(wacc[0].item())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier(device_ids=[local_rank])
    torch.cuda.synchronize()

    acc = None
    t0 = time.perf_counter()
    for k in range(args.steps):
        acc = step(acc)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier(device_ids=[local_rank])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    iters_total = int(acc[0].item())
    passes_total = int(acc[1].item())
    acc_passes_total = int(acc[3].item())
    kern_ms = [a.elapsed_time(b) for a, b in acc[2]]

    # ---- max over ranks, totals over ranks
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        s = torch.tensor([iters_total, passes_total], dtype=torch.int64, device="cuda")
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        iters_all, passes_all = int(s[0].item()), int(s[1].item())
        # per-GPU imbalance of the data-dependent work (no inter-GPU rebalancing, report the spread)
        lo = torch.tensor([iters_total], dtype=torch.int64, device="cuda")
        hi = lo.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        rank_iters = [int(lo.item()) / args.steps, int(hi.item()) / args.steps]
    else:
        iters_all, passes_all = iters_total, passes_total
        rank_iters = [iters_total / args.steps, iters_total / args.steps]

    # ---- the single end-of-job collective: gather results to rank 0 (timed separately)
    tg = time.perf_counter()
    gathered = ta.gather_output(x, {"stop_reason": out.stop_reason, "num_iters": out.num_iters,
                                    "final_cost": out.final_cost}, P_total=P * world)
    torch.cuda.synchronize()
    gather_ms = (time.perf_counter() - tg) * 1e3

    # ---- correctness guard on the timed work (size-independent property: planted solution recovered)
    stop = out.stop_reason
    ok_frac = float((stop >= 0).double().mean().item())
    max_dev = float((x - xstar).abs().max().item())
    if rank == 0:
        log(f"[bench] device={info['name']} CUs={info['num_cus']} succeeded={ok_frac:.4f} max|x-x*|={max_dev:.2e} "
            f"mean iters/problem={iters_total / args.steps / P:.2f} gather={gather_ms:.2f} ms")
    assert ok_frac == 1.0, "some problems failed"
    assert max_dev < 2e-2, "planted solution not recovered"

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    bytes_per_pass = model.algorithmic_bytes_per_pass  # m (n+1) sizeof(T)
    kern_avg_s = float(np.mean(kern_ms)) * 1e-3
    passes_per_launch = passes_total / args.steps
    achieved = bytes_per_pass * passes_per_launch / kern_avg_s / 1e9
    # measured STREAM-like read ceiling over the same packed buffer, next to the nominal peak
    try:
        stream_read = ctx.hbm_read_GBps(model.packed[: min(model.packed.shape[0], 64)].contiguous() if large else model.packed, reps=5)
    except Exception as e:  # noqa: BLE001
        log(f"[bench] hbm read probe failed: {e}")
        stream_read = None
    # secondary ceiling ("MFMA ceiling reported additionally"): matrix-core flops ISSUED by the accumulate
    # passes (NBM(NBM+1)/2 tiles of v_mfma_*_16x16x4 = 2048 flop per 4 rows) against the dense fp32 / fp64 MFMA peak
    if large:  # the library GEMM computes the full n x n square: 2 m n^2 flop per accumulate pass
        mfma_flop_per_pass = 2 * m * n * n
    else:
        lay = ta.api.dense_row_layout(tdt, n, m)
        mfma_flop_per_pass = (lay["rows_padded"] // 4) * (lay["nb"] * (lay["nb"] + 1) // 2) * 2048
    mfma_tflops = mfma_flop_per_pass * (acc_passes_total / args.steps) / kern_avg_s / 1e12
    mfma_peak = MFMA_PEAK_TFLOPS[tag]
    traffic = None
    pmc_file = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc_file):
        try:
            with open(pmc_file) as f:
                pm = json.load(f)
            if pm.get("workload") == args.workload and pm.get("problems") == P:
                traffic = pm.get("hbm_bytes_per_launch")
        except Exception:  # noqa: BLE001
            traffic = None
    result = {
        "metric": "LM iterations/s (batched dense n<=50)" if not large else f"LM iterations/s (batched dense n={n}, library-backed path)",
        "value": iters_all / elapsed,
        "unit": "LM iterations/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": tag, "data": "synthetic",
        "config": {"workload": desc, "problems_per_gpu": P, "n": n, "m": m,
                   "options": "benchmarks/options.h (max_iters 10, min_error 0, min_rerr_dec 1e-12, min_step_norm2 1e-16, max_consec_failures 3)",
                   "parallelism": f"problem-sharded x{world}, no data-path collective, one result gather",
                   "gather_ms": gather_ms, "lm_iterations_per_step_all_gpus": iters_all / args.steps,
                   "lm_iterations_per_step_per_gpu_min_max": rank_iters,
                   "device": info["name"], "num_cus": info["num_cus"]},
        "roofline": {"bound": "hbm", "kernel": "lm_fused_kernel" if not large else "large_rows_vec_kernel + rocBLAS gemm_batched + large_chol_solve_kernel (whole pass)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "measured_read_ceiling_GBps": stream_read,
                     "frac_of_measured_ceiling": (achieved / stream_read) if stream_read else None,
                     "algorithmic_bytes_per_pass": bytes_per_pass, "passes_per_launch": passes_per_launch,
                     "kernel_ms_avg": kern_avg_s * 1e3, "kernel_ms_all": kern_ms,
                     "mfma_secondary": {"achieved": mfma_tflops, "peak": mfma_peak, "unit": "TFLOP/s",
                                        "frac": mfma_tflops / mfma_peak, "issued_flop_per_accumulate_pass": mfma_flop_per_pass,
                                        "accumulate_passes_per_launch": acc_passes_total / args.steps}},
    }
    if large:  # AI = 2 m n^2 / (m (n + 1) sizeof) = 64 flop/B at n = 128 fp32, beyond the 19.7 flop/B ridge: the MFMA roof bounds this path
        r = result["roofline"]
        sec = r.pop("mfma_secondary")
        result["roofline"] = {"bound": "mfma", "kernel": r["kernel"], "achieved": sec["achieved"], "peak": sec["peak"], "unit": "TFLOP/s",
                              "frac": sec["frac"], "traffic": None, "flop_per_accumulate_pass": sec["issued_flop_per_accumulate_pass"],
                              "accumulate_passes_per_launch": sec["accumulate_passes_per_launch"],
                              "kernel_ms_avg": r["kernel_ms_avg"], "kernel_ms_all": r["kernel_ms_all"],
                              "hbm_secondary": {"achieved": r["achieved"], "peak": r["peak"], "unit": "GB/s", "frac": r["frac"],
                                                "algorithmic_bytes_per_pass": r["algorithmic_bytes_per_pass"],
                                                "passes_per_launch": r["passes_per_launch"]}}
    if not args.no_cpu and world == 1:  # the CPU baseline leg runs on rank 0 at N = 1 only
        np_dtype = np.float32 if tdt == torch.float32 else np.float64
        est = 4e-4 * (n * n * m) / (50 * 50 * 2000) * 8  # rough seconds per problem (8 iterations)
        sample = args.cpu_problems or int(max(16, min(P, 15.0 / max(est, 1e-6))))
        base, _ = cpu_baseline(n, m, np_dtype, pod, sample)
        result["cpu_baseline"] = base
        result["config"]["speedup_vs_cpu_1thread"] = result["value"] / base["value"]
    print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
