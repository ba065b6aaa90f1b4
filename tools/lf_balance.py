#!/usr/bin/env python3
"""The 64 <= n <= 128 kernel (n = 128, m = 4096, fp32) against the batch size: how much of the distance between the 512-problem line
of bench.py --workload large128 and the kernel's own rate is the one-problem-per-slot finish (512 = 256 CUs x 2 resident workgroups: the
launch lasts as long as the CU that holds the two longest problems).  Prints time, rate, the fraction of the f32 MFMA peak by the passes
run, and the per-problem iteration / pass histogram with the balance bound mean / max that follows from it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tinyopt_amd as ta
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
from large_n_bench import synth
PEAK = 157.3e12
opts = ta.Options.benchmark()
for P in (256, 512, 1024, 2048, 4096):
    A, b, x0, xs = synth(P, 128, 4096, torch.float32)
    model = ta.DenseRowNatural(A, b)
    x = x0.clone(); out = ta.Optimize(x, model, opts); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        x.copy_(x0); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ta.Optimize(x, model, opts, out=out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e-3)
    it = out.num_iters.cpu().numpy().astype(np.int64); cnt = out.counters.cpu().numpy()
    t = min(ts)
    acc = int(cnt[0])
    hist = np.bincount(it, minlength=11)
    print(f"P={P:5d}: {t*1e3:7.3f} ms  {it.sum()/t/1e3:7.0f} k it/s  {acc*4096*129*130/t/PEAK:.3f} of the f32 peak by passes run   iterations per problem: mean {it.mean():.2f}, "
          f"max {it.max()}  (mean / max = {it.mean()/it.max():.3f})   histogram 0..10: {hist.tolist()}", flush=True)
