#!/usr/bin/env python3
"""Seam check of the LDS-staged fp32 pass (n = 48..51): (g, H, cost) of toa_accumulate and the cost-only call against the oracle,
ragged row counts; then the pass timings of tools/probe.py's C4 shape.  Run with TINYOPT_AMD_LIB=<variant built with
-DTOA_STAGEDQ_SEAM> and without, same box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import tinyopt_amd as ta
from oracle import pyoracle as oracle

bad = 0
for n in (48, 49, 50, 51):
    for m in (5, 16, 17, 63, 64, 203, 2000):
        P = 5
        A, b, x0, _ = oracle.synth_dense_row(P, n, m, np.float32, seed=100 * n + m)
        g_ref, H_ref, c_ref, _ = oracle.dense_row_accumulate(A.astype(np.float64), b.astype(np.float64), x0.astype(np.float64))
        model = ta.DenseRow.from_arrays(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
        x = torch.from_numpy(x0).cuda()
        g, H, c, nres = ta.accumulate(model, x, want_grad=True)
        _, _, c2, _ = ta.accumulate(model, x, want_grad=False)
        eg = np.abs(g.cpu().numpy() - g_ref).max() / np.abs(g_ref).max()
        eH = np.abs(H.cpu().numpy() - H_ref).max() / np.abs(H_ref).max()
        ec = np.abs(c.cpu().numpy() - c_ref).max() / np.abs(c_ref).max()
        ec2 = np.abs(c2.cpu().numpy() - c_ref).max() / np.abs(c_ref).max()
        same = bool((c.cpu().numpy() == c2.cpu().numpy()).all())
        ok = eg < 3e-5 and eH < 3e-5 and ec < 3e-5 and ec2 < 3e-5
        bad += not ok
        print(f"n={n} m={m}: g {eg:.1e} H {eH:.1e} cost {ec:.1e} cost-only {ec2:.1e} acc==eval bits {same} {'ok' if ok else 'FAIL'}", flush=True)
print("FAILURES:", bad)
