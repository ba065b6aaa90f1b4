#!/usr/bin/env python3
"""VERDICT r04 #1 go / no-go (ii): the existing row-split form (toa_lm_run_split: one data-pass launch + one step launch per
LM iteration, `splits` waves per problem) on a C4-shaped batch whose rows FIT the 256 MiB Infinity Cache (512 problems =
209 MB) against one that does not (4096 problems = 1.67 GB).  Same kernels, same waves per problem, same occupancy per
launch from 512 x 6 = 3072 waves upward: the per-problem time of a data pass is what the cache changes.
   python tools/llc_split_probe.py [splits]        (run it under rocprofv3 --kernel-trace --stats for per-kernel times)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tinyopt_amd as ta


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts)


def main():
    splits = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    only = [int(a) for a in sys.argv[2:]] or [256, 512, 640, 768, 1024, 2048, 4096]
    n, m = 50, 2000
    opts = ta.Options.benchmark()
    for P in only:
        model, x0, _ = ta.DenseRow.synthetic(P, n, m, torch.float32)
        x = x0.clone()
        out = ta.Optimize(x, model, opts, splits=splits)

        def run():
            x.copy_(x0)
            ta.Optimize(x, model, opts, out=out, splits=splits)

        t = timed(run)
        cnt = out.counters.cpu().numpy()
        its = int(out.num_iters.sum().item())
        # accumulate seam over the same batch, back to back (every launch streams every row once)
        t_acc = timed(lambda: ta.accumulate(model, x0, True))
        t_ev = timed(lambda: ta.accumulate(model, x0, False))
        print(f"P={P:5d} ({P * 408000 / 2**20:7.1f} MB) splits={splits}: solve {t:8.3f} ms = {t / P * 1e3:7.3f} us/problem, "
              f"{its / P:.2f} it/problem, passes acc={cnt[0]} eval={cnt[1]}  |  one-wave seam: accumulate {t_acc / P * 1e3:.3f} "
              f"us/problem ({P * 408000 / t_acc / 1e6:.0f} GB/s), evaluate {t_ev / P * 1e3:.3f} us/problem ({P * 408000 / t_ev / 1e6:.0f} GB/s)")


if __name__ == "__main__":
    main()
