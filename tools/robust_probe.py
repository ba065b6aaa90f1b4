#!/usr/bin/env python3
"""DenseRow with an M-estimator at the C4 shape: the compiled-in family (launch-per-iteration form, one chunk per problem) against the
same residual + Jacobian supplied as text (RowModel: the loss inside the fused kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import tinyopt_amd as ta
from test_gpu_row_models import manual_body
from row_model_bench import timeit
P, n, m, dt = 12500, 50, 2000, torch.float32
if len(sys.argv) > 3: P, n, m = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
gen = torch.Generator(device="cuda").manual_seed(5)
A = torch.rand(P, m, n, dtype=dt, device="cuda", generator=gen) * 2 - 1
xs = torch.rand(P, n, dtype=dt, device="cuda", generator=gen) * 2 - 1
t = torch.einsum("pmn,pn->pm", A, xs)
b = t + 0.1 * torch.sin(t)
b[:, ::20] += 2.0     # planted outliers
x0 = xs + 0.3 * (torch.rand(P, n, dtype=dt, device="cuda", generator=gen) * 2 - 1)
items = torch.cat([A, b[..., None]], dim=2).contiguous()
opts = ta.Options.benchmark()
for loss in (None, ("huber", 0.5)):
    for name, model in (("compiled-in", ta.DenseRow.from_arrays(A, b)), ("text+J", ta.JitResidual(manual_body(n), n=n, item_scalars=n + 1, dtype=dt, kind="accumulate").bind(items))):
        if loss: model = model.with_loss(*loss)
        x = x0.clone(); out = ta.Optimize(x, model, opts)
        def run():
            x.copy_(x0); ta.Optimize(x, model, opts, out=out)
        ms = timeit(run); its = int(out.num_iters.sum())
        print(f"{name:12s} loss={loss}: {ms:8.3f} ms  {its / ms / 1e3:7.3f} M it/s  iters/problem {its / P:.2f}  err {float((x - xs).abs().max()):.2e}", flush=True)
