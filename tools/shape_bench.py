#!/usr/bin/env python3
"""Whole-solve time of the fused kernel on a synthetic DenseRow batch of any shape (A/B of variant libraries on shapes
bench.py has no workload for).   usage: python tools/shape_bench.py <n> <m> <P> <f32|f64>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import tinyopt_amd as ta

n, m, P = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dt = torch.float32 if sys.argv[4] == "f32" else torch.float64
model, x0, xs = ta.DenseRow.synthetic(P, n, m, dt)
opts = ta.Options.benchmark()
x = x0.clone()
out = ta.Optimize(x, model, opts)
torch.cuda.synchronize()
assert bool((out.stop_reason >= 0).all()) and float((x - xs).abs().max()) < 2e-2
ts = []
for _ in range(10):
    x.copy_(x0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ta.Optimize(x, model, opts, out=out)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
print(f"n={n} m={m} P={P} {sys.argv[4]}: median {ts[len(ts) // 2]:.4f} ms  min {ts[0]:.4f} ms  iters/problem {float(out.num_iters.double().mean()):.2f}")
