#!/usr/bin/env python3
"""Throughput map (GPU): batched DenseRow LM solves over the supported range of n, both dtypes, benchmark options.
Prints a markdown table: LM iterations/s, data passes/s, algorithmic TB/s and fraction of the 8 TB/s HBM peak."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tinyopt_amd as ta


def run(n, m, tdt, target_bytes=4.0e9):
    model0, _, _ = ta.DenseRow.synthetic(1, n, m, tdt)
    bpp = model0.algorithmic_bytes_per_pass
    P = int(max(2048, min(200000, target_bytes // bpp)))
    model, x0, xs = ta.DenseRow.synthetic(P, n, m, tdt)
    opts = ta.Options.benchmark()
    x = x0.clone()
    out = ta.Optimize(x, model, opts)
    ts = []
    for _ in range(5):
        x.copy_(x0)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); ta.Optimize(x, model, opts, out=out); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    t = sorted(ts)[1]
    iters = int(out.num_iters.sum().item())
    passes = int((out.counters[0] + out.counters[1]).item())
    ok = bool((out.stop_reason >= 0).all().item())
    err = float((x - xs).abs().max().item())
    return P, t, iters / t, passes * bpp / t / 1e12, ok, err


def main():
    print("| dtype | n | m | layout (NBM, THIN) | problems | ms / launch | M LM it/s | algorithmic TB/s | % of 8 TB/s | max abs(x - x*) |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for tdt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        for n, m in ((6, 1000), (12, 500), (16, 600), (18, 700), (31, 1000), (34, 1200), (50, 2000), (63, 2000)):
            lay = ta.api.dense_row_layout(tdt, n, m)
            P, t, its, tbs, ok, err = run(n, m, tdt)
            print(f"| {tag} | {n} | {m} | ({lay['nb']}, {lay['thin']}) | {P} | {t * 1e3:.2f} | {its / 1e6:.1f} | {tbs:.2f} | {100 * tbs / 8:.0f} | {err:.1e}{'' if ok else ' FAILED'} |")


if __name__ == "__main__":
    main()
