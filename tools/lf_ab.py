#!/usr/bin/env python3
"""Same-process timing of the 64 <= n <= 128 kernel at n = 128, m = 4096 for several batch sizes (argv[1] = 1: the row-split data pass, toa_tuning::large_row_split)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tinyopt_amd as ta
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
from large_n_bench import synth
opts = ta.Options.benchmark()
ROWSPLIT = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ta.api.default_context().set_tuning(large_row_split=ROWSPLIT)
for P in (512, 1024, 2048):
    A, b, x0, xs = synth(P, 128, 4096, torch.float32)
    model = ta.DenseRowNatural(A, b)
    x = x0.clone(); out = ta.Optimize(x, model, opts); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        x.copy_(x0); torch.cuda.synchronize(); t0 = time.perf_counter(); ta.Optimize(x, model, opts, out=out); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    it = int(out.num_iters.sum()); cnt = out.counters.cpu().numpy()
    t = min(ts)
    print(f"row_split={ROWSPLIT} P={P}: {t*1e3:.3f} ms  {it/t/1e3:.0f} k it/s  iters/problem {it/P:.3f}  acc passes {int(cnt[0])}  TFLOP/s(sym) {int(cnt[0])*4096*129*130/t/1e12:.1f}  max|x-x*| {float((x-xs).abs().max()):.2e}")
