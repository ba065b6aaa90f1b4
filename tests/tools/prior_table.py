#!/usr/bin/env python3
"""Cross-check with the reference's ONLY published table (docs/benchmark-ceres-table.png, README.md:125):
mean wall time of one whole Optimize() call on the Gaussian-prior problem r = (x - y)/sigma, m = n, fp64,
manual-Jacobian callback, benchmarks/options.h options.  Prints, per n: published tinyopt us (unknown x86 CPU),
the CPU oracle us on this host (1 thread), and the batched GPU time per solve (1 MI355X, P problems in flight)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import tinyopt_amd as ta
from oracle import pyoracle

PUBLISHED = {3: 1.49, 6: 2.47, 12: 5.15, 33: 25.68, 50: 56.26}   # "Dense VecX Prior n" rows, us
opts = ta.Options.benchmark()
print("| n | published tinyopt (us, unknown CPU) | CPU oracle here (us/solve, 1 thread) | MI355X batched (us/solve) | solves/s/GPU | mean iters |")
print("|---|---|---|---|---|---|")
for n, pub in PUBLISHED.items():
    Pc = 20000
    y, s, x0 = pyoracle.synth_gaussian_prior(Pc, n, np.float64)
    r = pyoracle.gaussian_prior_lm(y, s, x0, opts.to_pod())
    cpu_us = r["seconds"] / Pc * 1e6
    P = 1 << 20
    g = torch.Generator(device="cuda").manual_seed(1)
    yd = torch.rand(P, n, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    sd = torch.rand(P, n, dtype=torch.float64, device="cuda", generator=g) + 0.5
    xd0 = torch.rand(P, n, dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    model = ta.GaussianPrior(yd, sd)
    x = xd0.clone()
    out = ta.Optimize(x, model, opts)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        x.copy_(xd0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ta.Optimize(x, model, opts, out=out); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    assert (out.stop_reason >= 0).all() and (x - yd).abs().max() < 1e-9
    print(f"| {n} | {pub} | {cpu_us:.2f} | {best * 1e3 / P:.4f} | {P / best * 1e3:.3e} | {out.num_iters.double().mean().item():.2f} |")
