#!/usr/bin/env python3
"""Accuracy of the device Accumulate (g, H, cost) against an fp64 numpy reference, at x0 and near x*."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import tinyopt_amd as ta
from oracle import pyoracle

def ref64(A, b, x):
    A = A.astype(np.float64); b = b.astype(np.float64); x = x.astype(np.float64)
    t = np.einsum("pmn,pn->pm", A, x)
    r = t + 0.1 * np.sin(t) - b
    J = (1 + 0.1 * np.cos(t))[:, :, None] * A
    return np.einsum("pmn,pm->pn", J, r), np.einsum("pmi,pmj->pij", J, J), (r * r).sum(1)

P, n, m = 8, 50, 2000
A, b, x0, xs = pyoracle.synth_dense_row(P, n, m, np.float32)
model = ta.DenseRow.from_arrays(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
for name, x in (("x0", x0), ("x*", xs.astype(np.float32))):
    g64, H64, c64 = ref64(A, b, x)
    g, H, c, _ = ta.accumulate(model, torch.from_numpy(x).cuda())
    go, Ho, co, _ = pyoracle.dense_row_accumulate(A, b, x)
    def rel(a, r): return np.abs(a - r).max() / np.abs(r).max()
    print(f"[{os.environ.get('TINYOPT_AMD_LIB','default')[-24:]}] {name}: device vs fp64: g {rel(g.cpu().numpy(), g64):.2e} H {rel(H.cpu().numpy(), H64):.2e} "
          f"c {np.abs(c.cpu().numpy()/c64-1).max():.2e} | oracle(f32) vs fp64: g {rel(go, g64):.2e} H {rel(Ho, H64):.2e} c {np.abs(co/c64-1).max():.2e}  |g|max {np.abs(g64).max():.3e}")
