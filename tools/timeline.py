#!/usr/bin/env python3
"""Reads a toa_debug_timeline file (ctx.debug_timeline(path)) (per-problem start / end stamps of fused launches, 100 MHz ticks) and prints how many
problems were in flight over the last launch's duration — where a launch loses time (ramp, steady state, drain)."""
import sys
import numpy as np

launches, cur = [], None
for line in open(sys.argv[1]):
    if line.startswith("#"):
        cur = []
        launches.append((line.strip(), cur))
    else:
        a, b = line.split()
        cur.append((int(a), int(b)))
hdr, rows = launches[-1]
t = np.array(rows, dtype=np.float64)
t0 = t[:, 0].min()
s, e = (t[:, 0] - t0) * 1e-5, (t[:, 1] - t0) * 1e-5     # ms
T = e.max()
print(hdr, "duration %.3f ms, mean problem time %.3f ms (min %.3f max %.3f)" % (T, (e - s).mean(), (e - s).min(), (e - s).max()))
bins = 40
edges = np.linspace(0, T, bins + 1)
for i in range(bins):
    mid = 0.5 * (edges[i] + edges[i + 1])
    active = int(((s <= mid) & (e > mid)).sum())
    started = int(((s >= edges[i]) & (s < edges[i + 1])).sum())
    print("%6.2f ms  in flight %5d  started %5d  %s" % (mid, active, started, "#" * (active // 64)))
