#!/usr/bin/env python3
"""A BATCH of SE3 reprojection problems (n = 6 on the SE3 manifold: tests/sophus.cpp's `Optimize(pose, lambda)`, BASELINE C5's residual)
with the residual supplied as text — device AD through pose * exp(d) — against the compiled-in SE3Reproj family.
usage: se3_batch_probe.py [P] [points] [f32|f64]      ($TOA_JIT_FLAGS=-DTOA_JIT_ROW_MIN=13: the register-Gram model of rounds 3-5)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import tinyopt_amd as ta
from oracle import pyoracle
from row_model_bench import timeit
from test_gpu_jit import REPROJ
P = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dtype, tdt = (np.float32, torch.float32) if (len(sys.argv) > 3 and sys.argv[3] == "f32") else (np.float64, torch.float64)
data, p0, pstar = pyoracle.synth_se3_reproj(P, N, dtype, seed=4)
d = torch.from_numpy(data).cuda()
opts = ta.Options()
jit = ta.JitResidual(REPROJ, n=6, item_scalars=5, residuals_per_item=2, header_scalars=8, dtype=tdt, manifold="se3")
text = jit.bind(d[:, 8:].reshape(P, N, 5).contiguous(), header=d[:, :8].contiguous())
x0 = torch.from_numpy(p0.copy()).cuda()
res = {}
for name, model in (("compiled-in SE3Reproj", ta.SE3Reproj(d, N)), ("text, device AD", text)):
    x = x0.clone(); out = ta.Optimize(x, model, opts)
    def run():
        x.copy_(x0); ta.Optimize(x, model, opts, out=out)
    ms = timeit(run); its = int(out.num_iters.sum())
    res[name] = x.clone()
    st = f"  build: {jit.stats()}" if model is text else ""
    print(f"{name:24s} P={P} points={N} {str(tdt)[6:]}: {ms:8.3f} ms  {its / ms / 1e3:7.3f} M it/s  iters/problem {its / P:.2f}  "
          f"max|pose - truth| {float((x.double().cpu() - torch.from_numpy(pstar).double()).abs().max()):.2e}{st}", flush=True)
print(f"max|text - compiled-in| {float((res['text, device AD'] - res['compiled-in SE3Reproj']).abs().max()):.2e}")
