#!/usr/bin/env python3
"""The headline shape (n = 50, m = 2000, fp32) through four doors: the compiled-in DenseRow family, the same residual with its
Jacobian supplied as TEXT (RowModel over the user's functor), as text WITHOUT a Jacobian (row-per-lane chunked Jets), and the
compiled-in AD model.  usage: python tools/row_model_bench.py [P] [n] [m] [f32|f64]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

import tinyopt_amd as ta
from test_gpu_row_models import ad_body, manual_body


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts)


def narrow_bodies(n):
    """n <= 12 (JetModel: x and the Jets are REGISTER arrays of the lane): loops over x[j] have to unroll completely — a running
    index into a register array is scratch memory (144 / 272 B per lane and 1.5-3 x the time with the row models' bodies)."""
    man = (f"T t = x[0] * p[0];\n#pragma unroll\nfor (int j = 1; j < {n}; ++j) t += x[j] * p[j];\nT sn, cs; sincos_t(t, &sn, &cs);\n"
           f"r[0] = t + T(0.1) * sn - p[{n}];\nif (want_grad) {{\n  const T sc = T(1) + T(0.1) * cs;\n#pragma unroll\n  for (int j = 0; j < {n}; ++j) J[0][j] = sc * p[j];\n}}")
    ad = f"S t = x[0] * p[0];\n#pragma unroll\nfor (int j = 1; j < {n}; ++j) t = t + x[j] * p[j];\nr[0] = t + T(0.1) * sin(t) - p[{n}];"
    return man, ad


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    m = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
    dt = torch.float64 if (len(sys.argv) > 4 and sys.argv[4] == "f64") else torch.float32
    gen = torch.Generator(device="cuda").manual_seed(5)
    A = torch.rand(P, m, n, dtype=dt, device="cuda", generator=gen) * 2 - 1
    xs = torch.rand(P, n, dtype=dt, device="cuda", generator=gen) * 2 - 1
    t = torch.einsum("pmn,pn->pm", A, xs)
    b = t + 0.1 * torch.sin(t)
    x0 = xs + 0.3 * (torch.rand(P, n, dtype=dt, device="cuda", generator=gen) * 2 - 1)
    items = torch.cat([A, b[..., None]], dim=2).contiguous()
    mb, ab = narrow_bodies(n) if n <= 12 else (manual_body(n), ad_body(n))
    models = [("compiled-in", ta.DenseRow.from_arrays(A, b)),
              ("text+J", ta.JitResidual(mb, n=n, item_scalars=n + 1, dtype=dt, kind="accumulate").bind(items)),
              ("text AD", ta.JitResidual(ab, n=n, item_scalars=n + 1, dtype=dt).bind(items))]
    if n in (12, 50):
        models.append(("built-in AD", ta.DenseRowAD(A, b)))
    only = os.environ.get("ROWBENCH_ONLY")
    if only:
        models = [mm for mm in models if mm[0] in only.split(",")]
    opts = ta.Options.benchmark()
    base = None
    for name, model in models:
        x = x0.clone()
        out = ta.Optimize(x, model, opts)

        def run():
            x.copy_(x0)
            ta.Optimize(x, model, opts, out=out)
        ms = timeit(run)
        its = int(out.num_iters.sum())
        acc = timeit(lambda: ta.accumulate(model, x0, True))
        ev = timeit(lambda: ta.accumulate(model, x0, False))
        base = base or ms
        cn = [int(v) for v in out.counters[:5].cpu()]
        st = f"   build: {model.res.stats()}" if hasattr(model, "res") else ""
        if os.environ.get("ROWBENCH_SHORT"):
            print(f"{name:12s} P={P} n={n} m={m} {str(dt)[6:]}: solve {ms:8.3f} ms  {its / ms * 1e3 / 1e6:7.3f} M it/s  acc {acc:7.3f} ms  cost-only {ev:7.3f} ms{st}", flush=True)
            continue
        gbs = P * m * (n + 1) * A.element_size() / (acc * 1e-3) / 1e9
        print(f"{name:12s} P={P} n={n} m={m} {str(dt)[6:]}: solve {ms:8.3f} ms  {its / ms * 1e3 / 1e6:7.3f} M it/s  ({ms / base:5.2f}x compiled-in)   "
              f"accumulate seam {acc:7.3f} ms = {gbs:6.0f} GB/s, cost-only {ev:7.3f} ms   err {float((x - xs).abs().max()):.2e}   passes per iteration: {cn[0] / its:.3f} streamed + {cn[1] / its:.3f} cost-only, {cn[4] / its:.3f} from the memo{st}", flush=True)


if __name__ == "__main__":
    main()
