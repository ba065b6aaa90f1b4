#!/usr/bin/env python3
"""Phase breakdown probe (GPU): times the K1/K2 seam, the K3 seam and the fused kernel separately so
that the fused kernel's time can be attributed (accumulate vs evaluate vs LDL^T vs queue tail)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tinyopt_amd as ta


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts), sum(ts) / len(ts)


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
    if wl == "shape":   # probe.py shape n m f32|f64 P [P ...]
        n, m, dt = int(sys.argv[2]), int(sys.argv[3]), (torch.float64 if sys.argv[4] == "f64" else torch.float32)
        Ps = tuple(int(v) for v in sys.argv[5:])
        wl = f"n={n} m={m} {sys.argv[4]}"
    else:
        n, m, dt = (50, 2000, torch.float32) if wl == "c4" else (12, 500, torch.float64)
        Ps = (12288, 12500) if wl == "c4" else (8192, 10000)
    for P in Ps:
        model, x0, xs = ta.DenseRow.synthetic(P, n, m, dt)
        bpp = model.algorithmic_bytes_per_pass
        t_acc = timeit(lambda: ta.accumulate(model, x0, True))
        t_ev = timeit(lambda: ta.accumulate(model, x0, False))
        g, H, c, _ = ta.accumulate(model, x0, True)
        t_sol = timeit(lambda: ta.solve_damped(H, g, 1.0001))
        opts = ta.Options.benchmark()
        x = x0.clone()
        out = ta.Optimize(x, model, opts)
        def run():
            x.copy_(x0); ta.Optimize(x, model, opts, out=out)
        t_f = timeit(run)
        cnt = out.counters.cpu().numpy()
        print(f"[{wl}] P={P}: accumulate {t_acc[0]:.3f} ms ({bpp*P/t_acc[0]/1e6:.0f} GB/s)  evaluate {t_ev[0]:.3f} ms "
              f"({bpp*P/t_ev[0]/1e6:.0f} GB/s)  solve_damped {t_sol[0]:.3f} ms  fused {t_f[0]:.3f} ms  "
              f"counters acc={cnt[0]} eval={cnt[1]} solves={cnt[2]}  "
              f"sum-of-phases estimate {(cnt[0]*t_acc[0] + cnt[1]*t_ev[0] + cnt[2]*t_sol[0])/P:.3f} ms")


if __name__ == "__main__":
    main()
