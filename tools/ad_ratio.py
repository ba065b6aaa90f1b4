#!/usr/bin/env python3
"""Throughput of device AD against the hand-derived MFMA path on the same problems: the DenseRow residual
written as r(x) only (TOA_MODEL_DENSE_ROW_AD, JetRowModel: chunked Jets in the MFMA operand layout) vs DenseRowModel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import tinyopt_amd as ta


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts)


for (P, n, m, dt) in ((10000, 12, 500, torch.float64), (4096, 50, 2000, torch.float32)):
    gen = torch.Generator(device="cuda").manual_seed(5)
    A = torch.rand(P, m, n, dtype=dt, device="cuda", generator=gen) * 2 - 1
    xs = torch.rand(P, n, dtype=dt, device="cuda", generator=gen) * 2 - 1
    t = torch.einsum("pmn,pn->pm", A, xs)
    b = t + 0.1 * torch.sin(t)
    x0 = xs + 0.3 * (torch.rand(P, n, dtype=dt, device="cuda", generator=gen) * 2 - 1)
    an = ta.DenseRow.from_arrays(A, b)
    ad = ta.DenseRowAD(A, b)
    opts = ta.Options.benchmark()
    res = {}
    for name, model in (("analytic", an), ("ad", ad)):
        x = x0.clone()
        out = ta.Optimize(x, model, opts)

        def run():
            x.copy_(x0)
            ta.Optimize(x, model, opts, out=out)
        ms = timeit(run)
        its = int(out.num_iters.sum())
        res[name] = (ms, its, its / ms * 1e3, timeit(lambda: ta.accumulate(model, x0, True)))
    print(f"P={P} n={n} m={m} {str(dt)[6:]}: analytic {res['analytic'][0]:.3f} ms ({res['analytic'][2] / 1e6:.2f} M it/s, accumulate "
          f"{res['analytic'][3]:.3f} ms)   device AD {res['ad'][0]:.3f} ms ({res['ad'][2] / 1e6:.2f} M it/s, accumulate {res['ad'][3]:.3f} ms)   "
          f"AD / analytic: {res['ad'][0] / res['analytic'][0]:.1f}x the time, {res['ad'][3] / res['analytic'][3]:.1f}x in the accumulate pass", flush=True)
