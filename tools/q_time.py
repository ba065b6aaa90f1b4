#!/usr/bin/env python3
"""Times the K1 / K2 seam (toa_accumulate with and without gradient) at the C4 shape: median and minimum of 30 launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tinyopt_amd as ta
P, n, m = 12500, 50, 2000
model, x0, xs = ta.DenseRow.synthetic(P, n, m, torch.float32)
bpp = model.algorithmic_bytes_per_pass
def t(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]
a = t(lambda: ta.accumulate(model, x0, True))
e = t(lambda: ta.accumulate(model, x0, False))
print(f"{sys.argv[1] if len(sys.argv) > 1 else ''} accumulate median {a[0]:.4f} min {a[1]:.4f} ms ({bpp*P/a[0]/1e6:.0f} GB/s)   evaluate median {e[0]:.4f} min {e[1]:.4f} ms ({bpp*P/e[0]/1e6:.0f} GB/s)")
