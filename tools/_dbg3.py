import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import tinyopt_amd as ta
from test_gpu_row_models import manual_body
from row_model_bench import timeit
P, n, m, dt = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 50, 2000, torch.float32
gen = torch.Generator(device="cuda").manual_seed(5)
A = torch.rand(P, m, n, dtype=dt, device="cuda", generator=gen) * 2 - 1
xs = torch.rand(P, n, dtype=dt, device="cuda", generator=gen) * 2 - 1
t = torch.einsum("pmn,pn->pm", A, xs); b = t + 0.1 * torch.sin(t)
x0 = xs + 0.3 * (torch.rand(P, n, dtype=dt, device="cuda", generator=gen) * 2 - 1)
items = torch.cat([A, b[..., None]], dim=2).contiguous()
models = [("compiled-in", ta.DenseRow.from_arrays(A, b)), ("text+J", ta.JitResidual(manual_body(n), n=n, item_scalars=n + 1, dtype=dt, kind="accumulate").bind(items))]
for name, model in models:
    for mi in (0, 1, 2, 4, 6):
        opts = ta.Options.benchmark(); opts.max_iters = mi
        x = x0.clone(); out = ta.Optimize(x, model, opts)
        def run():
            x.copy_(x0); ta.Optimize(x, model, opts, out=out)
        ms = timeit(run)
        cn = [int(v) for v in out.counters[:5].cpu()]
        print(f"{name:12s} max_iters={mi}: {ms:7.3f} ms   iterations {int(out.num_iters.sum()) / P:.2f}/problem  streamed {cn[0] / P:.2f} cost-only {cn[1] / P:.2f} memo {cn[4] / P:.2f} solves {cn[2] / P:.2f}", flush=True)
