#!/usr/bin/env python3
"""Throughput of the n > 63 path (TOA_MODEL_DENSE_ROW_NATURAL: the workgroup-per-problem persistent kernel of
large_fused.hip up to n = 128, rows kernel + rocBLAS GEMM + workgroup LDL^T / rocSOLVER Cholesky + LM state
machine kernels beyond) on synthetic DenseRow problems, next to the one-wavefront fused kernel at n = 63 and the CPU
restatement (1 thread) on a bounded sample.   usage: python tools/large_n_bench.py [--no-cpu]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import tinyopt_amd as ta


def synth(P, n, m, dt, seed=1):
    g = torch.Generator(device="cuda").manual_seed(seed)
    A = torch.rand(P, m, n, dtype=dt, device="cuda", generator=g) * 2 - 1
    xs = torch.rand(P, n, dtype=dt, device="cuda", generator=g) * 2 - 1
    t = torch.einsum("pmn,pn->pm", A, xs)
    b = t + 0.1 * torch.sin(t) + 1e-3 * (torch.rand(P, m, dtype=dt, device="cuda", generator=g) * 2 - 1)
    x0 = xs + 0.5 * (torch.rand(P, n, dtype=dt, device="cuda", generator=g) * 2 - 1) / np.sqrt(n / 50)
    return A, b, x0, xs


def main():
    no_cpu = "--no-cpu" in sys.argv
    opts = ta.Options.benchmark()
    print("| dtype | n | m | problems | ms / solve of the batch | LM it/s | algorithmic GB/s (m (n+1) per pass) | TFLOP/s, full-square 2 m n^2 accounting (symmetric: half) | CPU oracle it/s (1 thread) |")
    print("|---|---|---|---|---|---|---|---|---|")
    for dt, n, m, P in ((torch.float32, 63, 2000, 2048), (torch.float32, 64, 2000, 2048), (torch.float32, 96, 3000, 1024),
                        (torch.float32, 128, 4096, 512), (torch.float32, 128, 4096, 2048),
                        (torch.float32, 256, 8192, 128), (torch.float32, 512, 8192, 64), (torch.float32, 768, 8192, 32), (torch.float32, 1024, 8192, 16),
                        (torch.float32, 128, 65536, 1), (torch.float32, 128, 16384, 4), (torch.float32, 96, 20000, 1),   # a few huge problems: the row-split pipeline
                        (torch.float64, 256, 4096, 64), (torch.float64, 64, 2000, 1024),
                        (torch.float64, 96, 3000, 512), (torch.float64, 128, 4096, 256)):
        A, b, x0, xs = synth(P, n, m, dt)
        model = ta.DenseRowNatural(A, b) if n > 63 else ta.DenseRow.from_arrays(A, b)
        x = x0.clone()
        out = ta.Optimize(x, model, opts)
        torch.cuda.synchronize()
        assert bool((out.stop_reason >= 0).all()) and float((x - xs).abs().max()) < 2e-2, (n, float((x - xs).abs().max()))
        ts = []
        for _ in range(3):
            x.copy_(x0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ta.Optimize(x, model, opts, out=out)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        t = min(ts)
        iters = int(out.num_iters.sum())
        cnt = out.counters.cpu().numpy()
        passes = int(cnt[0] + cnt[1])
        es = A.element_size()
        gbs = passes * m * (n + 1) * es / t / 1e9
        tf = int(cnt[0]) * 2.0 * m * n * n / t / 1e12
        cpu = ""
        if not no_cpu:
            from oracle import pyoracle
            S = max(1, min(P, int(2e9 / (m * n * n * 8))))
            r = pyoracle.dense_row_lm(A[:S].cpu().numpy(), b[:S].cpu().numpy(), x0[:S].cpu().numpy(), opts.to_pod())
            cpu = f"{r['iters'].sum() / r['seconds']:.0f}"
        print(f"| {'f32' if dt == torch.float32 else 'f64'} | {n} | {m} | {P} | {t * 1e3:.2f} | {iters / t:.0f} | {gbs:.0f} | {tf:.1f} | {cpu} |", flush=True)


if __name__ == "__main__":
    main()
