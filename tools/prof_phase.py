#!/usr/bin/env python3
"""Runs ONE phase kernel a few times (for rocprofv3 --pmc passes).  usage: prof_phase.py {acc|eval|solve|fused} [c4|c3]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tinyopt_amd as ta
phase = sys.argv[1]
wl = sys.argv[2] if len(sys.argv) > 2 else "c4"
P, n, m, dt = (12500, 50, 2000, torch.float32) if wl == "c4" else (10000, 12, 500, torch.float64)
model, x0, xs = ta.DenseRow.synthetic(P, n, m, dt)
g, H, c, _ = ta.accumulate(model, x0, True)
opts = ta.Options.benchmark()
x = x0.clone()
out = ta.Optimize(x, model, opts)
torch.cuda.synchronize()
for _ in range(3):
    if phase == "acc":
        ta.accumulate(model, x0, True)
    elif phase == "eval":
        ta.accumulate(model, x0, False)
    elif phase == "solve":
        ta.solve_damped(H, g, 1.0001)
    else:
        x.copy_(x0); ta.Optimize(x, model, opts, out=out)
torch.cuda.synchronize()
