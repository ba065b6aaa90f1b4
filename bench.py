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
import ctypes
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
    # a BATCH of C2-sized problems (C2 itself is ONE problem, latency-bound: SINGLE below) — what a GPU is for at this size; the narrow routes of
    # round 6 (an item per lane, the Gram in registers: csrc/models_jet.hpp JetModel over the packed rows)
    "c2_batch_f32": (131072, 6, 1000, torch.float32, "f32", "131072 problems/GPU x n=6 x m=1000 DenseRow fp32 (the C2 shape, batched)"),
    "c2_batch": (65536, 6, 1000, torch.float64, "f64", "65536 problems/GPU x n=6 x m=1000 DenseRow fp64 (the C2 shape, batched)"),
    # the C4 shape through the reference's OTHER doors: the residual handed over as TEXT at run time —
    # with its own Jacobian row (a manual Accumulate callback, docs/API.md:37-57, benchmarks/dense.cpp:57-66,90-99), or as r(x)
    # only, differentiated on the device (optimize_autodiff.h:91-166).  Items = rows [a_i | b_i] in the natural layout.
    "c4_text": (12500, 50, 2000, torch.float32, "f32", "C4 shard shape (12500 problems/GPU x n=50 x m=2000 fp32), DenseRow residual AND its Jacobian row supplied as C++ text at run time (TOA_JIT_ACCUMULATE; csrc/row_model.hpp)"),
    "c4_ad": (12500, 50, 2000, torch.float32, "f32", "C4 shard shape (12500 problems/GPU x n=50 x m=2000 fp32), DenseRow residual supplied as C++ text, Jacobian by device AD (chunked Jets, a row per lane; csrc/row_model.hpp)"),
    # beyond one wavefront: rows kernel + batched library GEMM + workgroup Cholesky; MFMA-bound
    "large128": (512, 128, 4096, torch.float32, "f32", "512 problems/GPU x n=128 x m=4096 DenseRow fp32, workgroup-per-problem kernel (64 <= n <= 128)"),
    "large256": (128, 256, 8192, torch.float32, "f32", "128 problems/GPU x n=256 x m=8192 DenseRow fp32, launch-per-stage pipeline (n > 128): rows kernel + hand-written MFMA Gram (operand-sharing deal of the 36 tiles) + one-workgroup blocked Cholesky, passes enqueued ahead"),
}
# single-problem configs of BASELINE.json (latency-bound: SURVEY §8d "report us/iter and GB/s"); replicas only at N > 1
SINGLE = {
    "c1": "C1 sqrt2: scalar x, 1 residual (tests/sqrt2.cpp), starts {1, -0.3, 3.2}, fp64",
    "c2": "C2: ONE problem, n=6, m=1000 DenseRow fp64 (benchmarks/dense.cpp extension)",
    "c5": "C5: ONE SE3 pose, 25000 points = 50000 reprojection residuals, fp64",
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# dense MFMA peaks of the dtypes this path computes in: f32-input MFMA 157.3 TF (MI355X_MICROARCH.md), f64 78.6 TF (vendor spec)
MFMA_PEAK_TFLOPS = {"f32": 157.3, "f64": 78.6}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def eigen_probe():
    """"if find_package(Eigen3) succeeds on the box, additionally build an Eigen-backed variant ...; otherwise state
    'Eigen unavailable; baseline is restatement'".  The reference itself (header-only over Eigen 3.4) does not travel to the GPU
    box and no Eigen-backed variant is shipped, so this only REPORTS what the box has."""
    import glob
    hits = []
    for pat in ("/usr/include/eigen3/Eigen/Core", "/usr/local/include/eigen3/Eigen/Core", "/opt/*/include/eigen3/Eigen/Core",
                "/usr/include/Eigen/Core", "/opt/rocm/include/Eigen/Core"):
        hits += glob.glob(pat)
    if not hits:
        return "unavailable: no Eigen/Core on this box; the baseline is the restatement (oracle/lm_oracle.hpp, kind = port)"
    return f"present at {os.path.dirname(os.path.dirname(hits[0]))} but unused: the reference's sources do not travel with the repo; baseline is the restatement"


def cpu_baseline(n, m, np_dtype, pod, budget_problems, loss=None, th=0.0):
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
    r1 = pyoracle.dense_row_lm(A, b, x0, pod, nthreads=1, lib=lib, loss=loss, th2=th * th)
    it1 = int(r1["iters"].sum())
    ncores = os.cpu_count() or 1
    rall = pyoracle.dense_row_lm(A, b, x0, pod, nthreads=ncores, lib=lib) if loss is None else r1   # (the oracle's loss hook is single-threaded)
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
        "value": it1 / r1["seconds"], "unit": "LM iterations/s", "cores": 1, "kind": "port", "eigen": eigen_probe(),
        "sample": f"{budget_problems} problems of the same workload (n={n}, m={m}), {it1} LM iterations, "
                  f"{r1['seconds']:.1f} s single thread; oracle/lm_oracle.hpp built with {build}",
        "all_cores": {"value": int(rall["iters"].sum()) / rall["seconds"], "cores": ncores, "seconds": rall["seconds"]},
        "host_cpu": model,
    }, r1


class _DryEvent:
    """torch.cuda.Event's interface on the host clock (bench.py --dry-run)."""

    def __init__(self, enable_timing=True):
        self.t = 0.0

    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return max((other.t - self.t) * 1e3, 1e-3)


class _DryModel:
    def __init__(self, P, n, m, tdt, xstar):
        self.P, self.n, self.m, self.dtype, self.xstar = P, n, m, tdt, xstar
        self.algorithmic_bytes_per_pass = m * (n + 1) * torch.empty(0, dtype=tdt).element_size()
        self.packed = torch.zeros(1, dtype=tdt)


def _dry_optimize(rank):
    """Stand-in for ta.Optimize in a --dry-run: writes the planted solution into x and an Output-shaped object with
    rank-dependent counts (so that the MIN / MAX / SUM reductions over ranks have something to tell apart)."""
    from types import SimpleNamespace

    def optimize(x, model, opts, out=None):
        P = x.shape[0]
        if out is None:
            out = SimpleNamespace(stop_reason=torch.zeros(P, dtype=torch.int32), num_iters=torch.zeros(P, dtype=torch.int32),
                                  final_cost=torch.zeros(P, dtype=torch.float64), counters=torch.zeros(8, dtype=torch.int64))
        x.copy_(model.xstar)
        out.stop_reason.fill_(3)
        out.num_iters.fill_(7 + rank % 2)
        out.final_cost.copy_(torch.arange(P, dtype=torch.float64) + 1000.0 * rank)
        out.counters.zero_()
        out.counters[0] = 6 * P
        out.counters[1] = P
        out.counters[4] = P
        return out
    return optimize


def run_single(args, ta, rank, world, local_rank):
    """BASELINE configs C1 / C2 / C5: ONE problem (C1: three scalar starts) per GPU — latency-bound, "replicas only" at
    N > 1.  A step = one whole solve from x0 (`toa_lm_run`, benchmarks/options.h options); the line reports
    us per solve and per LM iteration (HIP events on the launch stream), the algorithmic GB/s of the data passes and the
    oracle on the same problem on one host core."""
    from tinyopt_amd import synth
    wl = args.workload
    opts = ta.Options.benchmark()
    pod = opts.to_pod()
    tdt = torch.float64
    if wl == "c2":
        n, m, P = 6, 1000, 1
        model, x0, xstar = ta.DenseRow.synthetic(P, n, m, tdt, problem0=rank)
        bytes_per_pass = model.algorithmic_bytes_per_pass
    elif wl == "c5":
        n, npts, P = 6, 25000, 1
        data, p0, pstar = synth.synth_se3_reproj(P, npts, np.float64, seed=4 + rank)
        model = ta.SE3Reproj(torch.from_numpy(data).cuda(), npts)
        x0, xstar = torch.from_numpy(p0).cuda(), torch.from_numpy(pstar).cuda()
        m = 2 * npts
        bytes_per_pass = model.algorithmic_bytes_per_pass
    else:  # c1: the reference test's own options (tests/sqrt2.cpp:106-112), not the benchmark's
        opts = ta.Options()
        opts.max_iters = 20
        opts.max_consec_failures = 0
        pod = opts.to_pod()
        n, m, P = 1, 1, 3
        model = ta.Sqrt2(P, tdt)
        x0 = torch.tensor([[1.0], [-0.3], [3.2]], dtype=tdt, device="cuda")
        xstar = None
        bytes_per_pass = 0
    ctx = ta.api.default_context(local_rank)
    info = ctx.info()
    # A solve here is 50-100 us, so anything else on the stream would be a large part of what the line reports: every step
    # gets its own start vector and its own Output, set up (and zeroed) BEFORE the timed region, and the iteration counts
    # are read after it — the timed region holds the solves and nothing else.
    nwarm = max(args.warmup, 1)
    xs = [x0.clone() for _ in range(nwarm + args.steps)]
    outs = [ta.Optimize(x0.clone(), model, opts) for _ in range(nwarm + args.steps)]
    for o in outs:
        o.counters.zero_()
    torch.cuda.synchronize()

    def step(i):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ta.Optimize(xs[i], model, opts, out=outs[i], zero_counters=False)
        e1.record()
        return e0, e1

    for i in range(nwarm):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier(device_ids=[local_rank])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    recs = [step(nwarm + i) for i in range(args.steps)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier(device_ids=[local_rank])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timed = outs[nwarm:]
    iters_total = int(sum(int(o.num_iters.sum().item()) for o in timed))
    passes_total = int(sum(int((o.counters[0] + o.counters[1]).item()) for o in timed))
    x, out = xs[-1], outs[-1]
    for o in timed:
        assert bool((o.stop_reason >= 0).all()), "solve failed"
    kern_us = [r[0].elapsed_time(r[1]) * 1e3 for r in recs]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        sm = torch.tensor([iters_total], dtype=torch.int64, device="cuda")
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        iters_all = int(sm.item())
    else:
        iters_all = iters_total
    assert bool((out.stop_reason >= 0).all()), "solve failed"
    for x in xs[nwarm:]:     # every timed solve, not only the last one
        if wl == "c1":
            assert float((x.abs() - 2.0 ** 0.5).abs().max()) < 1e-5
        else:
            assert float((x - xstar).abs().max()) < (5e-3 if wl == "c2" else 5e-4), "planted solution not recovered"
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    solve_us = float(np.median(kern_us))
    its_per_solve = iters_total / args.steps
    passes_per_solve = passes_total / args.steps
    achieved = bytes_per_pass * passes_per_solve / (solve_us * 1e-6) / 1e9 if bytes_per_pass else 0.0
    result = {
        "metric": "LM iterations/s (single problem, latency-bound)", "value": iters_all / elapsed, "unit": "LM iterations/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": SINGLE[wl], "problems_per_gpu": P, "n": n, "m": m,
                   "options": "benchmarks/options.h" if wl != "c1" else "tests/sqrt2.cpp:106-112 (defaults, max_iters 20, max_consec_failures 0)",
                   "parallelism": f"replicas x{world} (one problem does not shard, SURVEY §8e)",
                   "us_per_solve_device": solve_us, "us_per_lm_iteration_device": solve_us / max(its_per_solve / P, 1e-9),
                   "us_per_solve_host_wall": elapsed / args.steps * 1e6,
                   "lm_iterations_per_solve": its_per_solve / P, "data_passes_per_solve": passes_per_solve / P,
                   "device": info["name"], "num_cus": info["num_cus"]},
        "roofline": {"bound": "latency", "kernel": "wide_team_kernel / wide_persistent_kernel (one launch per solve)" if wl != "c1" else "lm_fused_kernel<Sqrt2Model>",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "note": "one problem cannot fill the chip: the solve is a serial chain of ~(data pass, fold, LDL^T, judge) "
                             "hand-overs; the figure of merit is us per LM iteration, GB/s is reported for scale only",
                     "algorithmic_bytes_per_pass": bytes_per_pass, "kernel_us_all": kern_us},
    }
    if not args.no_cpu and world == 1:
        from oracle import pyoracle   # the checker / CPU baseline only: the inputs above came from the package's own generators
        lib = pyoracle.load(pyoracle.build(march="native", out_dir=tempfile.mkdtemp(prefix="toa_oracle_")))
        x0h = x0.cpu().numpy()
        if wl == "c2":
            A, b, x0o, _ = pyoracle.synth_dense_row(1, n, m, np.float64, problem0=0)   # the problem rank 0 solved

            def solve_once():
                return int(pyoracle.dense_row_lm(A, b, x0o, pod, lib=lib)["iters"].sum())
        elif wl == "c5":
            stop, iters, cost = np.zeros(1, np.int32), np.zeros(1, np.int32), np.zeros(1)

            def solve_once():
                xh = np.array(x0h, copy=True)
                lib.oracle_se3_reproj_lm(1, 1, npts, data.ctypes.data, xh.ctypes.data, pod, stop.ctypes.data, iters.ctypes.data,
                                         cost.ctypes.data, None, None)
                return int(iters[0])
        else:
            def solve_once():
                return int(pyoracle.sqrt2_lm(x0h[:, 0].copy(), pod)["iters"].sum())
        solve_once()
        reps, t_cpu, it_cpu = 0, 0.0, 0
        while t_cpu < 2.0 and reps < 100000:
            tc = time.perf_counter()
            it_cpu += solve_once()
            t_cpu += time.perf_counter() - tc
            reps += 1
        result["cpu_baseline"] = {"value": it_cpu / t_cpu, "unit": "LM iterations/s", "cores": 1, "kind": "port", "eigen": eigen_probe(),
                                  "us_per_solve": t_cpu / reps * 1e6, "lm_iterations_per_solve": it_cpu / reps / P,
                                  "sample": f"the same problem solved {reps} times by oracle/lm_oracle.hpp (g++ -O3 -march=native), {t_cpu:.2f} s, one thread"}
        result["config"]["speedup_vs_cpu_1thread"] = result["value"] / result["cpu_baseline"]["value"]
    print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


BA_DESC = "bundle adjustment: 1024 scenes/GPU x 8 SE3 cameras x 256 points (4096 residuals, 816 unknowns per scene), fp64, points eliminated (Schur)"


def run_ba(args, ta, rank, world, local_rank):
    """batched bundle adjustment with the points eliminated (toa_ba_run).  The reference would run
    Optimize on the dense (6C + 3N)^2 system (math.h:232-240) — that is the CPU baseline (oracle/ba.hpp)."""
    from tinyopt_amd import synth
    P, C, N = (args.problems or 1024), 8, 256
    opts = ta.Options.benchmark()
    pod = opts.to_pod()
    data, x0h, xsh = synth.synth_ba(P, C, N, np.float64, seed=0x71940917 + rank)
    model = ta.BundleAdjustment(torch.from_numpy(data).cuda(), C, N)
    x0 = torch.from_numpy(x0h).cuda()
    ctx = ta.api.default_context(local_rank)
    info = ctx.info()
    x = x0.clone()
    out = ta.Optimize(x, model, opts)
    torch.cuda.synchronize()

    def step():
        x.copy_(x0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ta.Optimize(x, model, opts, out=out)
        e1.record()
        return e0, e1, out.num_iters.sum(dtype=torch.int64), (out.counters[0] + out.counters[1]).clone()

    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier(device_ids=[local_rank])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    recs = [step() for _ in range(args.steps)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier(device_ids=[local_rank])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    iters_total = int(sum(int(r[2].item()) for r in recs))
    passes_total = int(sum(int(r[3].item()) for r in recs))
    kern_ms = [r[0].elapsed_time(r[1]) for r in recs]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        sm = torch.tensor([iters_total], dtype=torch.int64, device="cuda")
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        iters_all = int(sm.item())
    else:
        iters_all = iters_total
    assert bool((out.stop_reason >= 0).all()), "solve failed"
    # size-independent property: the reprojection RMS ends at the planted noise level (0.5 px uniform -> 0.29 px rms per coordinate)
    rms = float((out.final_cost / (2 * C * N)).sqrt().max())
    assert rms < 0.5, f"reprojection rms {rms}"
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    bytes_per_pass = model.algorithmic_bytes_per_pass
    kern_s = float(np.mean(kern_ms)) * 1e-3
    achieved = bytes_per_pass * (passes_total / args.steps) / kern_s / 1e9
    traffic = traffic_raw = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest_ba.json")) as f:
            pm = json.load(f)
        if pm.get("workload") == "ba" and pm.get("scenes") == P:
            traffic, traffic_raw = pm.get("hbm_bytes_per_launch"), pm.get("hbm_bytes_per_launch_raw")
    except Exception:  # noqa: BLE001
        pass
    result = {
        "metric": "LM iterations/s (batched bundle adjustment, Schur complement)", "value": iters_all / elapsed, "unit": "LM iterations/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": BA_DESC, "problems_per_gpu": P, "cameras": C, "points": N, "n": 6 * C + 3 * N, "m": 2 * C * N,
                   "options": "benchmarks/options.h", "parallelism": f"scene-sharded x{world}, no data-path collective",
                   "iters_per_problem": iters_all / args.steps / (P * world), "final_reprojection_rms_px_max": rms,
                   "device": info["name"], "num_cus": info["num_cus"]},
        "roofline": {"bound": "hbm", "kernel": "ba_schur_kernel (one workgroup per scene, the whole solve)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_raw_counters": traffic_raw,
                     "traffic_source": ("profiles/pmc_latest_ba.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same workload and "
                                        "binary, collected by tools/refresh_profiles.sh; not this run)") if traffic else None,
                     "algorithmic_bytes_per_pass": bytes_per_pass, "passes_per_launch": passes_total / args.steps,
                     "note": "algorithmic bytes = observations (u, v, visibility) + points per pass; `traffic` = FETCH_SIZE x the C4 stream "
                             "calibration + WRITE_SIZE of the whole launch, per-scene work arrays included (`traffic_raw_counters`: "
                             "uncalibrated).  With the W blocks materialised (18 values per observation, written once and read twice "
                             "per iteration, 151 MB over the 512 resident scenes) it was 6.0 GB = 21x algorithmic "
                             "(profiles/r02_pmc_ba_before.json); the kernel is latency / issue-bound now, DESIGN.md 4c",
                     "kernel_ms_avg": kern_s * 1e3, "kernel_ms_all": kern_ms},
    }
    if not args.no_cpu and world == 1:
        from oracle import pyoracle
        lib = pyoracle.load(pyoracle.build(march="native", out_dir=tempfile.mkdtemp(prefix="toa_oracle_")))
        S = args.cpu_problems or 6
        tc = time.perf_counter()
        r = pyoracle.ba_lm(data[:S], x0h[:S], C, N, pod, history=False, lib=lib)
        t_cpu = time.perf_counter() - tc
        it_cpu = int(r["iters"].sum())
        result["cpu_baseline"] = {"value": it_cpu / t_cpu, "unit": "LM iterations/s", "cores": 1, "kind": "port", "eigen": eigen_probe(),
                                  "sample": f"{S} scenes of the same workload solved the reference's way — dense (6C + 3N)^2 = 816^2 Hessian + "
                                            f"dense LDL^T (oracle/ba.hpp; math.h:232-240) — {it_cpu} LM iterations, {t_cpu:.1f} s, one thread",
                                  "iters_per_problem": it_cpu / S}
        result["config"]["speedup_vs_cpu_1thread"] = result["value"] / result["cpu_baseline"]["value"]
    print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


BALISTS_DESC = "bundle adjustment with visibility lists: 4 scenes/GPU x 64 cameras x 5000 points x 6 observations per point (30 000 observations, 15 384 unknowns per scene), fp64"


def run_balists(args, ta, rank, world, local_rank):
    """SURVEY §8(f) rank 4 at a realistic shape: tens of cameras, each point seen by a few (toa_ba_lists_run).  A step = one
    batched solve from x0 with the benchmark options.  The reference's route — tinyopt::Optimize on the dense (6C + 3N)^2
    system (math.h:232-240): a 1.9 GB Hessian and ~1e12 flops per LDL^T per scene — is not runnable at this size; the CPU
    baseline is the dense oracle on the same generator at 16 cameras x 300 points and is labelled as such."""
    from tinyopt_amd import synth
    P, C, N, K = (args.problems or 4), 64, 5000, 6
    opts = ta.Options.benchmark()
    pod = opts.to_pod()
    intr, oc, op, ouv, x0h, xsh = synth.synth_ba_lists(P, C, N, K, np.float64, seed=0x71940917 + rank)
    model = ta.BundleAdjustmentLists(torch.from_numpy(intr).cuda(), torch.from_numpy(oc).cuda(), torch.from_numpy(op).cuda(),
                                     torch.from_numpy(ouv).cuda(), C, N)
    x0 = torch.from_numpy(x0h).cuda()
    ctx = ta.api.default_context(local_rank)
    info = ctx.info()
    x = x0.clone()
    out = ta.Optimize(x, model, opts)
    torch.cuda.synchronize()

    def step():
        x.copy_(x0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ta.Optimize(x, model, opts, out=out)
        e1.record()
        return e0, e1, out.num_iters.sum(dtype=torch.int64), (out.counters[0] + out.counters[1]).clone()

    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier(device_ids=[local_rank])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    recs = [step() for _ in range(args.steps)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier(device_ids=[local_rank])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    iters_total = int(sum(int(r[2].item()) for r in recs))
    passes_total = int(sum(int(r[3].item()) for r in recs))
    kern_ms = [r[0].elapsed_time(r[1]) for r in recs]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        sm = torch.tensor([iters_total], dtype=torch.int64, device="cuda")
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        iters_all = int(sm.item())
    else:
        iters_all = iters_total
    assert bool((out.stop_reason >= 0).all()), "solve failed"
    rms = float((out.final_cost / (2 * N * K)).sqrt().max())
    assert rms < 0.5, f"reprojection rms {rms}"
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    bytes_per_pass = model.algorithmic_bytes_per_pass
    kern_s = float(np.mean(kern_ms)) * 1e-3
    achieved = bytes_per_pass * (passes_total / args.steps) / kern_s / 1e9
    traffic_bl, traffic_src = None, None
    try:   # HBM bytes per batched solve, summed over every kernel of the pipeline (separate --pmc passes, tools/pmc_sum.sh; not this run)
        with open(os.path.join(ROOT, "profiles", "pmc_latest_balists.json")) as f:
            pm = json.load(f)
        if pm.get("problems") == P:
            traffic_bl = pm.get("hbm_bytes_per_launch")
            traffic_src = ("profiles/pmc_latest_balists.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over all kernels of the pipeline, same workload; "
                           "~80x the input bytes, ~5x the intermediates it must write and read once (J_c, J_p, r per observation, the reduced system per scene): "
                           "the Schur-complement kernel gathers 64-128 byte records of other cameras' observations; the pipeline is latency-bound at this "
                           "size, see the note)")
    except Exception:  # noqa: BLE001
        pass
    result = {
        "metric": "LM iterations/s (bundle adjustment with visibility lists, Schur complement)", "value": iters_all / elapsed,
        "unit": "LM iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": BALISTS_DESC, "problems_per_gpu": P, "cameras": C, "points": N, "observations": N * K,
                   "n": 6 * C + 3 * N, "m": 2 * N * K, "options": "benchmarks/options.h",
                   "parallelism": f"scene-sharded x{world}, no data-path collective",
                   "iters_per_problem": iters_all / args.steps / (P * world), "ms_per_lm_iteration": elapsed / max(iters_all / (P * world), 1) * 1e3,
                   "final_reprojection_rms_px_max": rms, "device": info["name"], "num_cus": info["num_cus"]},
        "roofline": {"bound": "latency", "kernel": "bl_* pipeline (10 launches per Build + Solve attempt, enqueued two passes ahead of the stop flag; one-workgroup blocked Cholesky of the 384 x 384 reduced camera system per scene, in place, one launch)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic_bl,
                     "traffic_source": traffic_src, "algorithmic_bytes_per_pass": bytes_per_pass, "passes_per_launch": passes_total / args.steps,
                     "note": "launch / latency-bound at this size (a few scenes, ~1 MB of observations each): the figure of merit is ms per LM "
                             "iteration; GB/s is reported for scale only",
                     "kernel_ms_avg": kern_s * 1e3, "kernel_ms_all": kern_ms},
    }
    if not args.no_cpu and world == 1:
        from oracle import pyoracle
        lib = pyoracle.load(pyoracle.build(march="native", out_dir=tempfile.mkdtemp(prefix="toa_oracle_")))
        Cs, Ns = 16, 300
        ds, x0s, _ = synth.synth_ba(1, Cs, Ns, np.float64, seed=5)
        tc = time.perf_counter()
        r = pyoracle.ba_lm(ds, x0s, Cs, Ns, pod, history=False, lib=lib)
        t_cpu = time.perf_counter() - tc
        it_cpu = int(r["iters"].sum())
        result["cpu_baseline"] = {"value": it_cpu / t_cpu, "unit": "LM iterations/s", "cores": 1, "kind": "port", "eigen": eigen_probe(),
                                  "sample": f"NOT the bench shape (a 15 384^2 dense Hessian is 1.9 GB and ~1e12 flops per LDL^T): the same generator at "
                                            f"{Cs} cameras x {Ns} points, all visible ({6 * Cs + 3 * Ns} unknowns), solved the reference's way — dense "
                                            f"Hessian + dense LDL^T (oracle/ba.hpp; math.h:232-240) — {it_cpu} LM iterations, {t_cpu:.1f} s, one thread",
                                  "iters_per_problem": float(it_cpu)}
    print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS) + sorted(SINGLE) + ["ba", "balists"])
    ap.add_argument("--problems", type=int, default=0, help="override problems per GPU (debug; invalidates the metric)")
    ap.add_argument("--cpu-problems", type=int, default=0, help="CPU baseline sample size (0 = auto)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg (profiling runs)")
    ap.add_argument("--loss", default="", help="an M-estimator on every residual of a packed DenseRow workload (c4, c3), e.g. huber:0.5 — "
                                               "losses::Huber(n2, th2, true) inside the cost functor (robust_norms.h:20-26); recorded in the line")
    ap.add_argument("--dry-run", action="store_true",
                    help="N > 1 plumbing check WITHOUT a GPU: gloo + CPU tensors and a stand-in for the solve go through the very same barriers, "
                         "reductions, result gather, watchdog and JSON assembly as a real run (tests/test_cpu_dist.py runs it at N = 2 and 8, "
                         "launched exactly as the driver launches the scaling bench); the line says data = dry-run and is not a measurement")
    ap.add_argument("--tuning", default="", help="A/B arms of the library as toa_tuning fields, e.g. coop_off=1,memo_off=1 (recorded in the line; "
                                                 "the default line is measured with none)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    gpu = not args.dry_run
    if gpu:
        torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl" if gpu else "gloo", rank=rank, world_size=world)

    import tinyopt_amd as ta

    if not gpu:
        if args.workload not in WORKLOADS or WORKLOADS[args.workload][1] > 63:
            raise SystemExit("--dry-run covers the sharded dense workloads (c4, c3)")
        args.no_cpu = True
    if args.tuning and gpu:   # typed per-handle state (include/tinyopt_amd.h toa_tuning): the library reads no environment variable
        ta.api.default_context(local_rank).set_tuning(**{k: int(v) for k, v in (kv.split("=") for kv in args.tuning.split(","))})
    if args.workload in SINGLE:
        return run_single(args, ta, rank, world, local_rank)
    if args.workload == "ba":
        return run_ba(args, ta, rank, world, local_rank)
    if args.workload == "balists":
        return run_balists(args, ta, rank, world, local_rank)
    P, n, m, tdt, tag, desc = WORKLOADS[args.workload]
    if args.problems:
        P = args.problems
    opts = ta.Options.benchmark()  # benchmarks/options.h:10-27
    pod = opts.to_pod()
    dev = "cuda" if gpu else "cpu"
    sync = torch.cuda.synchronize if gpu else (lambda: None)
    barrier = (lambda: dist.barrier(device_ids=[local_rank])) if gpu else dist.barrier
    Event = torch.cuda.Event if gpu else _DryEvent
    if gpu:
        ctx = ta.api.default_context(local_rank)
        info = ctx.info()
        optimize = ta.Optimize
    else:   # stand-ins with the shapes, dtypes and fields of the real objects (everything AFTER the solve is the real code path)
        P = args.problems or 13
        ctx, info = None, {"name": "dry-run (no device)", "num_cus": 0}
        optimize = _dry_optimize(rank)

    # ---- inputs resident in HBM before the timed region (weak scaling: rank r owns problems [r*P, (r+1)*P))
    large = n > 63
    if not gpu:
        xstar = torch.arange(rank * P, (rank + 1) * P, dtype=tdt)[:, None] * 0.001 + torch.arange(n, dtype=tdt)[None, :]
        x0 = torch.zeros(P, n, dtype=tdt)
        model = _DryModel(P, n, m, tdt, xstar)
    elif args.workload in ("c4_text", "c4_ad"):
        # the same distributions as DenseRow.synthetic, rows in the natural layout: an item = [a_i (n) | b_i]
        gen = torch.Generator(device="cuda").manual_seed(0x7194 + rank)
        A = torch.rand(P, m, n, dtype=tdt, device="cuda", generator=gen) * 2 - 1
        xstar = torch.rand(P, n, dtype=tdt, device="cuda", generator=gen) * 2 - 1
        t = torch.einsum("pmn,pn->pm", A, xstar)
        bvec = t + 0.1 * torch.sin(t) + 1e-3 * (torch.rand(P, m, dtype=tdt, device="cuda", generator=gen) * 2 - 1)
        x0 = xstar + 0.5 * (torch.rand(P, n, dtype=tdt, device="cuda", generator=gen) * 2 - 1)
        items = torch.cat([A, bvec[..., None]], dim=2).contiguous()
        del A, bvec, t
        if args.workload == "c4_text":
            body = (f"T t = x[0] * p[0];\nfor (int j = 1; j < {n}; ++j) t += x[j] * p[j];\nT sn, cs; sincos_t(t, &sn, &cs);\n"
                    f"r[0] = t + T(0.1) * sn - p[{n}];\nif (want_grad) {{\n  const T sc = T(1) + T(0.1) * cs;\n#pragma unroll\n"
                    f"  for (int j = 0; j < {n}; ++j) J[0][j] = sc * p[j];\n}}")
            jit = ta.JitResidual(body, n=n, item_scalars=n + 1, dtype=tdt, kind="accumulate", ctx=ctx)
        else:
            body = f"S t = x[0] * p[0];\n#pragma unroll 2\nfor (int j = 1; j < {n}; ++j) t = t + x[j] * p[j];\nr[0] = t + T(0.1) * sin(t) - p[{n}];"
            jit = ta.JitResidual(body, n=n, item_scalars=n + 1, dtype=tdt, ctx=ctx)
        model = jit.bind(items)
        jit_build = jit.stats()
    elif not large:
        model, x0, xstar = ta.DenseRow.synthetic(P, n, m, tdt, problem0=rank * P)
        if args.loss:
            model = model.with_loss(args.loss.split(":")[0], float(args.loss.split(":")[1]))
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
    out = optimize(x, model, opts)  # allocates result buffers once
    sync()

    def step(acc):
        """One timed unit: restart from x0, run the batched solve, accumulate the step's units on the
        device (no host sync).  Used identically for warmup and timing so that every lazily loaded
        torch kernel (copy, sum, add) is resident before the clock starts."""
        x.copy_(x0)
        e0 = Event(enable_timing=True)
        e1 = Event(enable_timing=True)
        e0.record()
        optimize(x, model, opts, out=out)  # one kernel launch on torch's current stream
        e1.record()
        it = out.num_iters.sum(dtype=torch.int64)
        ps = out.counters[0] + out.counters[1]   # passes that STREAMED the rows (accumulate + evaluate-only)
        ap = out.counters[0].clone()
        mp = out.counters[4].clone()              # Builds served from the memo of the last accepted point (no data pass)
        if acc is None:
            return (it, ps, [(e0, e1)], ap, mp)
        return (acc[0] + it, acc[1] + ps, acc[2] + [(e0, e1)], acc[3] + ap, acc[4] + mp)

    wacc = None
    for _ in range(max(args.warmup, 1) if args.warmup else 0):
        wacc = step(wacc)
    if wacc is not None:
        _ = int(wacc[0].item())
    sync()
    if world > 1:
        barrier()
    sync()

    acc = None
    t0 = time.perf_counter()
    for k in range(args.steps):
        acc = step(acc)
    sync()
    if world > 1:
        barrier()
    sync()
    elapsed = time.perf_counter() - t0
    elapsed_local = elapsed
    iters_total = int(acc[0].item())
    passes_total = int(acc[1].item())
    acc_passes_total = int(acc[3].item())
    memo_builds_total = int(acc[4].item())
    kern_ms = [a.elapsed_time(b) for a, b in acc[2]]
    # roofline of this rank's kernel, formed BEFORE the gather so that the watchdog's line carries it too
    bytes_per_pass = model.algorithmic_bytes_per_pass  # m (n+1) sizeof(T)
    kern_avg_s = float(np.mean(kern_ms)) * 1e-3
    passes_per_launch = passes_total / args.steps
    achieved = bytes_per_pass * passes_per_launch / kern_avg_s / 1e9

    # ---- max over ranks, totals over ranks
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        s = torch.tensor([iters_total, passes_total], dtype=torch.int64, device=dev)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        iters_all, passes_all = int(s[0].item()), int(s[1].item())
        # per-GPU imbalance of the data-dependent work (no inter-GPU rebalancing, report the spread)
        lo = torch.tensor([iters_total], dtype=torch.int64, device=dev)
        hi = lo.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        rank_iters = [int(lo.item()) / args.steps, int(hi.item()) / args.steps]
        # every rank's own kernel time and wall time, so that the first real multi-GPU run shows stragglers (one all_gather of 2 doubles)
        mine = torch.tensor([kern_avg_s * 1e3, elapsed_local * 1e3 / args.steps], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = {"kernel_ms_avg": [float(v[0].item()) for v in allr], "ms_per_step": [float(v[1].item()) for v in allr]}
    else:
        iters_all, passes_all = iters_total, passes_total
        rank_iters = [iters_total / args.steps, iters_total / args.steps]
        per_rank = None

    # ---- the single end-of-job collective: gather results to rank 0 (timed separately)
    # N > 1: `toa_gather` of the C-ABI — one ncclGather of (x, stop_reason, num_iters, final_cost) in native types
    # (2.7 MB per GPU at C4) on a communicator of its own (id broadcast through torch.distributed); if that fails the
    # torch.distributed gather of tinyopt_amd.dist is used and the line says so.  A watchdog guarantees that a hung
    # collective (this code has never run on more than one GPU before the driver's scaling run) cannot take the bench
    # line with it: after 120 s every rank leaves, rank 0 printing the measured line first.
    gather_impl, comm_init_ms, gather_ms = "none (single process)", None, 0.0
    watchdog = None
    if world > 1:
        import threading
        partial = {"metric": "LM iterations/s (batched dense n<=50)", "value": iters_all / elapsed, "unit": "LM iterations/s",
                   "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                   "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": tag, "data": "synthetic",
                   "config": {"workload": desc, "problems_per_gpu": P, "n": n, "m": m, "gather_ms": None,
                              "gather_error": "result gather did not complete within 120 s (line printed by the watchdog)"},
                   "roofline": {"bound": "hbm", "kernel": "lm_fused_kernel (rank 0's launches)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                                "algorithmic_bytes_per_pass": bytes_per_pass, "passes_per_launch": passes_per_launch,
                                "kernel_ms_avg": kern_avg_s * 1e3}}

        def _bail():
            if rank == 0:
                print(json.dumps(partial), flush=True)
            os._exit(0)

        watchdog = threading.Timer(120.0, _bail)
        watchdog.daemon = True
        watchdog.start()
    try:
        if world > 1 and not large and gpu and not os.environ.get("TOA_BENCH_TORCH_GATHER"):
            tc = time.perf_counter()
            comm = ta.Communicator.from_torch(ctx)
            sync()
            comm_init_ms = (time.perf_counter() - tc) * 1e3
            ta.gather_native(comm, x, out, P_total=P * world)      # warm (buffers, RCCL channel set-up)
            sync()
            barrier()
            tg = time.perf_counter()
            gathered = ta.gather_native(comm, x, out, P_total=P * world)
            sync()
            gather_ms = (time.perf_counter() - tg) * 1e3
            gather_impl = "toa_gather (C-ABI): one ncclGather, native dtypes"
            if rank == 0:
                assert gathered["x"].shape == (P * world, n) and torch.equal(gathered["x"][:P], x)
                assert torch.equal(gathered["num_iters"][:P], out.num_iters)
        else:
            raise RuntimeError("torch path selected")
    except Exception as e:  # noqa: BLE001
        if world > 1:
            log(f"[bench] native gather not used ({e}); falling back to torch.distributed.gather")
        tg = time.perf_counter()
        gathered = ta.gather_output(x, {"stop_reason": out.stop_reason, "num_iters": out.num_iters,
                                        "final_cost": out.final_cost}, P_total=P * world)
        sync()
        gather_ms = (time.perf_counter() - tg) * 1e3
        if world > 1:
            gather_impl = "torch.distributed.gather (float64 payload)"
            if rank == 0:   # the root holds every rank's shard in problem-id order
                assert gathered["x"].shape == (P * world, n) and torch.equal(gathered["x"][:P].to(x.dtype), x)
                assert gathered["num_iters"].shape[0] == P * world and torch.equal(gathered["num_iters"][:P].to(out.num_iters.dtype), out.num_iters)
    if watchdog is not None:
        watchdog.cancel()

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

    # measured STREAM-like read ceiling over the same packed buffer, next to the nominal peak
    try:
        if not gpu:
            raise RuntimeError("dry run")
        probe_src = model.packed
        if large:   # (a slice of whole problems, at most ~1 GiB; the natural layout's rows are 4 (n + 1)-byte records: cut on a 16-byte boundary)
            rows = max(1, min(probe_src.shape[0], (1 << 30) // max(1, probe_src.shape[1] * probe_src.element_size())))
            probe_src = probe_src[:rows]
        nb = probe_src.numel() * probe_src.element_size() // 16 * 16
        out_gbs = ctypes.c_double()
        ta.api.check(ctx.lib.toa_hbm_read_probe(ctx.h, ctypes.c_void_p(probe_src.data_ptr()), nb, 5, ctypes.byref(out_gbs)))
        stream_read = out_gbs.value
    except Exception as e:  # noqa: BLE001
        log(f"[bench] hbm read probe failed: {e}")
        stream_read = None
    # the same for a working set that stays on-die (the 256 MiB Infinity Cache): the ceiling of re-reads, next to HBM's
    try:
        if not gpu:
            raise RuntimeError("dry run")
        llc_read = ctx.llc_read_GBps(model.packed)
    except Exception as e:  # noqa: BLE001
        log(f"[bench] llc read probe failed: {e}")
        llc_read = None
    # secondary ceiling ("MFMA ceiling reported additionally"): matrix-core flops ISSUED by the accumulate
    # passes (NBM(NBM+1)/2 tiles of v_mfma_*_16x16x4 = 2048 flop per 4 rows) against the dense fp32 / fp64 MFMA peak
    if large:
        # ALGORITHMIC flops of one accumulate pass: the symmetric Gram of [J | r], (n + 1)(n + 2) / 2 multiply-adds per row.
        # (Round 1 priced this path at 2 m n^2, the FULL square its rocBLAS GEMM computed; large_fused_kernel only computes
        # the lower block triangle, so that accounting would now credit flops nobody performs.)
        mfma_flop_per_pass = m * (n + 1) * (n + 2)
        nbl = (n + 15) // 16
        if n <= 128:
            mfma_issued_per_pass = 4 * ((((m + 3) // 4 + 3) // 4)) * (nbl * (nbl + 1) // 2) * 2048   # 4 waves x steps x tiles x 16*16*4*2
        else:   # large_gram_kernel: 64 x 64 blocks of the lower block triangle, 16 tiles each, every 4-row step
            nb64 = (n + 63) // 64
            mfma_issued_per_pass = ((m + 3) // 4) * (nb64 * (nb64 + 1) // 2) * 16 * 2048
    else:
        lay = ta.api.dense_row_layout(tdt, n, m)
        mfma_flop_per_pass = (lay["rows_padded"] // 4) * (lay["nb"] * (lay["nb"] + 1) // 2) * 2048
    mfma_tflops = mfma_flop_per_pass * (acc_passes_total / args.steps) / kern_avg_s / 1e12
    mfma_peak = MFMA_PEAK_TFLOPS[tag]
    traffic = None
    pmc_file = os.path.join(ROOT, "profiles", "pmc_latest.json" if args.workload == "c4" else f"pmc_latest_{args.workload}.json")
    if os.path.exists(pmc_file):
        try:
            with open(pmc_file) as f:
                pm = json.load(f)
            if pm.get("workload") == args.workload and pm.get("problems") == P:
                traffic = pm.get("hbm_bytes_per_launch")
        except Exception:  # noqa: BLE001
            traffic = None
    result = {
        "metric": "LM iterations/s (batched dense n<=50)" if not large else f"LM iterations/s (batched dense n={n}, {'64 <= n <= 128' if n <= 128 else 'n > 128'} path)",
        "value": iters_all / elapsed,
        "unit": "LM iterations/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": tag, "data": "synthetic" if gpu else "dry-run (no GPU work: N > 1 plumbing check, NOT a measurement)",
        "config": {"workload": desc, "problems_per_gpu": P, "n": n, "m": m,
                   "options": "benchmarks/options.h (max_iters 10, min_error 0, min_rerr_dec 1e-12, min_step_norm2 1e-16, max_consec_failures 3)",
                   "parallelism": f"problem-sharded x{world}, no data-path collective, one result gather",
                   "gather_ms": gather_ms, "gather_impl": gather_impl, "comm_init_ms": comm_init_ms,
                   "lm_iterations_per_step_all_gpus": iters_all / args.steps,
                   "iters_per_problem": iters_all / args.steps / (P * world),
                   "lm_iterations_per_step_per_gpu_min_max": rank_iters,
                   "per_rank": per_rank,
                   "device": info["name"], "num_cus": info["num_cus"]},
        "roofline": {"bound": "hbm", "kernel": "lm_fused_kernel" if not large else ("large_fused_kernel (data pass + fold + blocked LDL^T + step: the whole launch)" if n <= 128 else "large_gram_kernel + the rest of the n > 128 pipeline (rows, reduce, pre, stage, one-workgroup blocked Cholesky, post: the whole batched solve, two staggered lanes)"),
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": (os.path.relpath(pmc_file, ROOT) + " (rocprofv3 --pmc passes of the same workload and binary, collected by tools/refresh_profiles.sh; not this run)") if traffic else None,
                     "measured_read_ceiling_GBps": stream_read,
                     "llc_read_ceiling_GBps": llc_read,
                     "frac_of_measured_ceiling": (achieved / stream_read) if stream_read else None,
                     "algorithmic_bytes_per_pass": bytes_per_pass, "passes_per_launch": passes_per_launch,
                     "passes_accounting": "only the passes that streamed the rows count (accumulate + evaluate-only, device counters); "
                                          "Builds served from the memo of the last accepted point read ~10 KB instead of a pass "
                                          "and are NOT credited",
                     "builds_from_memo_per_launch": memo_builds_total / args.steps,
                     # SURVEY.md §8(d) prices the roof per problem-ITERATION (m (n + 1) sizeof(T) bytes each: "100 % HBM = 19.6 M
                     # it/s" at C4).  With the memo an iteration no longer costs a pass, so this figure is NOT the DRAM rate of the
                     # kernel (that is `achieved`, from the passes actually streamed); it is the iteration rate on SURVEY's scale.
                     "per_iteration": {"bytes_per_iteration": bytes_per_pass,
                                       "iterations_per_launch": iters_all / args.steps / world,
                                       "achieved": bytes_per_pass * (iters_all / args.steps / world) / kern_avg_s / 1e9,
                                       "ceiling_iterations_per_s": HBM_PEAK_GBS * 1e9 / bytes_per_pass,
                                       "frac": bytes_per_pass * (iters_all / args.steps / world) / kern_avg_s / 1e9 / HBM_PEAK_GBS},
                     "kernel_ms_avg": kern_avg_s * 1e3, "kernel_ms_all": kern_ms,
                     "mfma_secondary": {"achieved": mfma_tflops, "peak": mfma_peak, "unit": "TFLOP/s",
                                        "frac": mfma_tflops / mfma_peak, "issued_flop_per_accumulate_pass": mfma_flop_per_pass,
                                        "accumulate_passes_per_launch": acc_passes_total / args.steps}},
    }
    if args.loss:
        result["metric"] = "LM iterations/s (batched dense n<=50, an M-estimator on every residual)"
        result["config"]["loss"] = args.loss
        result["config"]["tuning"] = args.tuning
    if args.workload in ("c4_text", "c4_ad"):
        result["metric"] = "LM iterations/s (batched dense n<=50, residual supplied as text at run time)"
        result["roofline"]["kernel"] = "lm_fused_kernel<RowModel<float, 3, 3, ...>> (hiprtc build of the user's functor)"
        result["config"]["jit_build"] = jit_build   # resident workgroups / CU, LDS per workgroup, registers, scratch of the run-time build
    if large:  # AI = (n + 2) / sizeof = 32.5 flop/B at n = 128 fp32, beyond the 19.7 flop/B ridge: the MFMA roof bounds this path
        r = result["roofline"]
        sec = r.pop("mfma_secondary")
        result["roofline"] = {"bound": "mfma", "kernel": r["kernel"], "achieved": sec["achieved"], "peak": sec["peak"], "unit": "TFLOP/s",
                              "frac": sec["frac"], "traffic": r["traffic"], "traffic_source": r["traffic_source"],
                              "flop_per_accumulate_pass": sec["issued_flop_per_accumulate_pass"],
                              "flop_accounting": "algorithmic: m (n+1)(n+2) per accumulate pass (symmetric Gram of [J | r]); evaluate passes, "
                                                 "the LDL^T and the step are in the time but not in the flops",
                              "issued_mfma_flop_per_accumulate_pass": mfma_issued_per_pass,
                              "frac_issued": mfma_issued_per_pass * sec["accumulate_passes_per_launch"] / (r["kernel_ms_avg"] * 1e-3) / 1e12 / sec["peak"],
                              "achieved_full_square_accounting_r01": 2 * m * n * n * sec["accumulate_passes_per_launch"] / (r["kernel_ms_avg"] * 1e-3) / 1e12,
                              "accumulate_passes_per_launch": sec["accumulate_passes_per_launch"],
                              # Builds served from the memo of the last accepted linearisation (round 5) run no data pass: they are in the
                              # time, not in `achieved`.  Priced per BUILD — every Build of the loop credited with a pass's flops, the
                              # scale on which the launches of earlier rounds (no memo: every Build streamed) were quoted:
                              "builds_from_memo_per_launch": r["builds_from_memo_per_launch"],
                              "per_build": {"builds_per_launch": sec["accumulate_passes_per_launch"] + r["builds_from_memo_per_launch"],
                                            "frac": sec["frac"] * (sec["accumulate_passes_per_launch"] + r["builds_from_memo_per_launch"]) / max(sec["accumulate_passes_per_launch"], 1e-9)},
                              "kernel_ms_avg": r["kernel_ms_avg"], "kernel_ms_all": r["kernel_ms_all"],
                              "hbm_secondary": {"achieved": r["achieved"], "peak": r["peak"], "unit": "GB/s", "frac": r["frac"],
                                                "algorithmic_bytes_per_pass": r["algorithmic_bytes_per_pass"],
                                                "passes_per_launch": r["passes_per_launch"],
                                                "measured_read_ceiling_GBps": r["measured_read_ceiling_GBps"],
                                                "llc_read_ceiling_GBps": r["llc_read_ceiling_GBps"]}}
    if args.workload == "large128" and gpu and world == 1 and not args.no_cpu:   # (--no-cpu = a profiling run: only the launches of the timed workload)
        # 512 problems = 256 CUs x 2 resident workgroups: every slot holds exactly ONE problem, so the launch lasts as long as the CU
        # with the two longest ones (iterations per problem: mean 7.3, max 10-11).  The kernel's own rate shows on a batch that keeps
        # the slots refilled — the same generator at four times the problems, after the timed region (profiles/r06_ab_log.md §4):
        itp = out.num_iters.to(torch.float64)
        result["roofline"]["one_problem_per_slot"] = {"iterations_per_problem_mean": float(itp.mean()), "iterations_per_problem_max": float(itp.max()),
                                                      "mean_over_max": float(itp.mean() / itp.max())}
        del model, x, x0, out
        torch.cuda.empty_cache()
        Pb = 4 * P
        gen = torch.Generator(device="cuda").manual_seed(0x7194 + 977)
        A = torch.rand(Pb, m, n, dtype=tdt, device="cuda", generator=gen) * 2 - 1
        xs_b = torch.rand(Pb, n, dtype=tdt, device="cuda", generator=gen) * 2 - 1
        tb = torch.einsum("pmn,pn->pm", A, xs_b)
        bb = tb + 0.1 * torch.sin(tb) + 1e-3 * (torch.rand(Pb, m, dtype=tdt, device="cuda", generator=gen) * 2 - 1)
        x0b = xs_b + 0.5 * (torch.rand(Pb, n, dtype=tdt, device="cuda", generator=gen) * 2 - 1) / (n / 50) ** 0.5
        mb = ta.DenseRowNatural(A, bb)
        del A, bb, tb
        xb = x0b.clone()
        ob = optimize(xb, mb, opts)
        sync()
        tms, accp = [], 0
        for _ in range(4):
            xb.copy_(x0b)
            e0, e1 = Event(enable_timing=True), Event(enable_timing=True)
            e0.record(); optimize(xb, mb, opts, out=ob); e1.record(); sync()
            tms.append(e0.elapsed_time(e1)); accp = int(ob.counters[0])
        tb_s = sum(tms[1:]) / len(tms[1:]) * 1e-3
        result["roofline"]["balanced_batch"] = {"problems": Pb, "kernel_ms_avg": tb_s * 1e3, "accumulate_passes_per_launch": accp,
                                                "lm_iterations_per_s": float(ob.num_iters.sum()) / tb_s,
                                                "frac": accp * result["roofline"]["flop_per_accumulate_pass"] / tb_s / 1e12 / result["roofline"]["peak"]}
        del mb, xb, x0b, ob
    if not args.no_cpu and world == 1:  # the CPU baseline leg runs on rank 0 at N = 1 only
        np_dtype = np.float32 if tdt == torch.float32 else np.float64
        est = 4e-4 * (n * n * m) / (50 * 50 * 2000) * 8  # rough seconds per problem (8 iterations)
        sample = args.cpu_problems or int(max(16, min(P, 15.0 / max(est, 1e-6))))
        base, r1 = cpu_baseline(n, m, np_dtype, pod, sample, *((args.loss.split(":")[0], float(args.loss.split(":")[1])) if args.loss else ()))
        base["iters_per_problem"] = float(r1["iters"].mean())   # next to config.iters_per_problem of the device
        result["cpu_baseline"] = base
        result["config"]["speedup_vs_cpu_1thread"] = result["value"] / base["value"]
        # the metric counts iterations, and the device's blocked fp32 sums resolve a few more last-bit steps than the oracle's
        # sequential sum takes: the same launches priced at the ORACLE's iteration count per problem
        result["config"]["value_at_oracle_iters"] = result["value"] * base["iters_per_problem"] / result["config"]["iters_per_problem"]
    print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
