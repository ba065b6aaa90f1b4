#!/usr/bin/env python3
"""Runs ONE phase kernel a few times (for rocprofv3 --pmc passes).  usage: prof_phase.py {acc|eval|solve|fused} [c4|c3|c4_text|c4_ad]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import tinyopt_amd as ta
phase = sys.argv[1]
wl = sys.argv[2] if len(sys.argv) > 2 else "c4"
P, n, m, dt = (10000, 12, 500, torch.float64) if wl == "c3" else (12500, 50, 2000, torch.float32)
if wl in ("c4_text", "c4_ad"):   # the C4 shape with the residual supplied as text (bench.py --workload c4_text / c4_ad: the same generator)
    from test_gpu_row_models import ad_body, manual_body
    gen = torch.Generator(device="cuda").manual_seed(0x7194)
    A = torch.rand(P, m, n, dtype=dt, device="cuda", generator=gen) * 2 - 1
    xs = torch.rand(P, n, dtype=dt, device="cuda", generator=gen) * 2 - 1
    t = torch.einsum("pmn,pn->pm", A, xs)
    b = t + 0.1 * torch.sin(t) + 1e-3 * (torch.rand(P, m, dtype=dt, device="cuda", generator=gen) * 2 - 1)
    x0 = xs + 0.5 * (torch.rand(P, n, dtype=dt, device="cuda", generator=gen) * 2 - 1)
    items = torch.cat([A, b[..., None]], dim=2).contiguous()
    del A, b, t
    jit = ta.JitResidual(manual_body(n), n=n, item_scalars=n + 1, dtype=dt, kind="accumulate") if wl == "c4_text" else ta.JitResidual(ad_body(n), n=n, item_scalars=n + 1, dtype=dt)
    model = jit.bind(items)
else:
    model, x0, xs = ta.DenseRow.synthetic(P, n, m, dt)
g, H, c, _ = ta.accumulate(model, x0, True)
opts = ta.Options.benchmark()
x = x0.clone()
out = ta.Optimize(x, model, opts)
torch.cuda.synchronize()
for _ in range(3):
    if phase == "acc":
        ta.accumulate(model, x0, True)
    elif phase == "eval":
        ta.accumulate(model, x0, False)
    elif phase == "solve":
        ta.solve_damped(H, g, 1.0001)
    else:
        x.copy_(x0); ta.Optimize(x, model, opts, out=out)
torch.cuda.synchronize()
