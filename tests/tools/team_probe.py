#!/usr/bin/env python3
"""Team form (one workgroup per small problem) against one wavefront per problem, batches of C2-sized problems.
Run twice: plain, and with argv[1] = 1 (toa_tuning::wide_no_autosplit: forces one wavefront per problem)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tinyopt_amd as ta

ta.api.default_context().set_tuning(wide_no_autosplit=int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for n, m in ((6, 1000), (12, 2000)):
    for P in (1, 16, 64, 128, 256, 512, 1024, 2048):
        model, x0, xs = ta.DenseRow.synthetic(P, n, m, torch.float64)
        opts = ta.Options.benchmark()
        x = x0.clone(); out = ta.Optimize(x, model, opts); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            x.copy_(x0)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); ta.Optimize(x, model, opts, out=out); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f"n={n} m={m} P={P}: {min(ts) * 1e3:.0f} us  max|x-x*|={float((x - xs).abs().max()):.1e}", flush=True)
