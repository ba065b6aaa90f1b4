#!/usr/bin/env python3
"""Where a row model's pass spends its time: the accumulate seam of the C4 shape as text, with parts of the pass ablated through
$TOA_JIT_FLAGS (csrc/row_model.hpp TOA_ROW_ABL: 1 no functor, 2 no Gram steps, 4 no [J | r] image).  Not a correctness tool."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

import tinyopt_amd as ta
from test_gpu_row_models import ad_body, manual_body


def timeit(fn, reps=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts)


P = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
n, m, dt = 50, 2000, torch.float32
gen = torch.Generator(device="cuda").manual_seed(5)
A = torch.rand(P, m, n, dtype=dt, device="cuda", generator=gen) * 2 - 1
xs = torch.rand(P, n, dtype=dt, device="cuda", generator=gen) * 2 - 1
t = torch.einsum("pmn,pn->pm", A, xs)
b = t + 0.1 * torch.sin(t)
items = torch.cat([A, b[..., None]], dim=2).contiguous()
x0 = xs.clone()
an = ta.DenseRow.from_arrays(A, b)
gb = P * m * (n + 1) * 4 / 1e9
print(f"compiled-in: accumulate {timeit(lambda: ta.accumulate(an, x0, True)):.3f} ms, cost-only {timeit(lambda: ta.accumulate(an, x0, False)):.3f} ms   ({gb:.2f} GB per pass)")
for kind, body in (("accumulate", manual_body(n)), ("residual", ad_body(n))):
    for abl in (0, 1, 2, 3, 7):
        os.environ["TOA_JIT_FLAGS"] = f"-DTOA_ROW_ABL={abl}"
        mod = ta.JitResidual(body, n=n, item_scalars=n + 1, dtype=dt, kind=kind).bind(items)
        ta_acc = timeit(lambda: ta.accumulate(mod, x0, True))
        ta_ev = timeit(lambda: ta.accumulate(mod, x0, False))
        print(f"{kind:10s} ABL={abl}: accumulate {ta_acc:7.3f} ms = {gb / ta_acc * 1e3:6.0f} GB/s   cost-only {ta_ev:7.3f} ms = {gb / ta_ev * 1e3:6.0f} GB/s", flush=True)
