#!/usr/bin/env python3
"""Where a fused launch loses time, from per-problem start / end stamps (toa_debug_timeline): problems in flight over the launch,
and the duration of a problem by the round it started in (first round: every wave starts in the same phase; last: the drain).
   python tools/timeline_rounds.py c3|c4 [problems]"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import tinyopt_amd as ta
from tinyopt_amd.api import default_context

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
P, n, m, dt = (12500, 50, 2000, torch.float32) if wl == "c4" else (10000, 12, 500, torch.float64)
if len(sys.argv) > 2:
    P = int(sys.argv[2])
model, x0, _ = ta.DenseRow.synthetic(P, n, m, dt)
opts = ta.Options.benchmark()
x = x0.clone()
out = ta.Optimize(x, model, opts)
for _ in range(3):
    x.copy_(x0)
    ta.Optimize(x, model, opts, out=out)
torch.cuda.synchronize()
path = os.path.join(tempfile.mkdtemp(), "tl.txt")
ctx = default_context()
ctx.debug_timeline(path)
x.copy_(x0)
ta.Optimize(x, model, opts, out=out)
torch.cuda.synchronize()
ctx.debug_timeline(None)
rows = [tuple(int(v) for v in line.split()) for line in open(path) if not line.startswith("#")]
t = np.array(rows, dtype=np.float64)
t0 = t[:, 0].min()
s, e = (t[:, 0] - t0) * 1e-2, (t[:, 1] - t0) * 1e-2     # us
T = e.max()
d = e - s
print(f"[{wl}] P={P}: launch {T:.1f} us between the first start and the last end; problem time mean {d.mean():.1f} us (min {d.min():.1f}, max {d.max():.1f})")
order = np.argsort(s, kind="stable")
W = 3072
for r in range((P + W - 1) // W):
    idx = order[r * W:(r + 1) * W]
    print(f"  started as #{r * W}..{r * W + len(idx) - 1}: start {s[idx].min():7.1f} .. {s[idx].max():7.1f} us, duration mean {d[idx].mean():6.1f} us "
          f"(p10 {np.percentile(d[idx], 10):6.1f}, p90 {np.percentile(d[idx], 90):6.1f}), last end {e[idx].max():7.1f}")
bins = 30
edges = np.linspace(0, T, bins + 1)
for i in range(bins):
    mid = 0.5 * (edges[i] + edges[i + 1])
    active = int(((s <= mid) & (e > mid)).sum())
    print("%7.1f us  in flight %5d  %s" % (mid, active, "#" * (active // 64)))
