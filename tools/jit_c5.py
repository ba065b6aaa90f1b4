"""BASELINE C5 shape (ONE SE3 pose, 25 000 points = 50 000 residuals, fp64) with the reprojection residual supplied as TEXT at run
time (toa_jit_lm_run -> the row-split form) beside the compiled-in SE3Reproj model: ms per solve, median of 20."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinyopt_amd as ta
from tinyopt_amd import synth

REPROJ = """
const S X = x[0] * p[0] + x[1] * p[1] + x[2] * p[2] + x[9];
const S Y = x[3] * p[0] + x[4] * p[1] + x[5] * p[2] + x[10];
const S Z = x[6] * p[0] + x[7] * p[1] + x[8] * p[2] + x[11];
r[0] = h[0] * X / Z + h[1] - p[3];
r[1] = h[0] * Y / Z + h[2] - p[4];
"""


def timed(fn, x, x0, reps=20):
    ts = []
    for _ in range(reps + 3):
        x.copy_(x0); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts[3:])) * 1e3


def main():
    npts, P = 25000, 1
    data, p0, _ = synth.synth_se3_reproj(P, npts, np.float64, seed=4)
    d = torch.from_numpy(data).cuda(); x0 = torch.from_numpy(p0).cuda(); x = x0.clone()
    o = ta.Options.benchmark()
    built_in = ta.SE3Reproj(d, npts)
    t0 = time.perf_counter()
    fit = ta.JitResidual(REPROJ, n=6, item_scalars=5, residuals_per_item=2, header_scalars=8, dtype=torch.float64, manifold="se3")
    t_compile = time.perf_counter() - t0
    jit = fit.bind(d[:, 8:].reshape(P, npts, 5).contiguous(), header=d[:, :8].contiguous())
    out_b = ta.Optimize(x, built_in, o); it_b = int(out_b.num_iters.sum())
    x.copy_(x0); t0 = time.perf_counter(); out_j = ta.Optimize(x, jit, o); torch.cuda.synchronize(); t_first = time.perf_counter() - t0
    it_j = int(out_j.num_iters.sum())
    ctx = ta.api.default_context()
    print(f"compile (fused + accumulate kernels) {t_compile:.2f} s from_cache={fit.from_cache}; first split call (wide module) {t_first:.2f} s")
    print(f"compiled-in SE3Reproj          : {timed(lambda: ta.Optimize(x, built_in, o, out=out_b), x, x0):.3f} ms / solve ({it_b} iterations)")
    print(f"run-time text, row-split (auto): {timed(lambda: ta.Optimize(x, jit, o, out=out_j), x, x0):.3f} ms / solve ({it_j} iterations)")
    with ctx.tuning(wide_multilaunch=1):
        print(f"run-time text, launch per iter : {timed(lambda: ta.Optimize(x, jit, o, out=out_j), x, x0):.3f} ms / solve")
    with ctx.tuning(wide_no_autosplit=1):
        print(f"run-time text, ONE wavefront   : {timed(lambda: ta.Optimize(x, jit, o, out=out_j), x, x0):.3f} ms / solve")


if __name__ == "__main__":
    main()
