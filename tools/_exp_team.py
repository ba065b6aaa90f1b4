import sys, os
sys.path.insert(0, os.getcwd())
import torch, tinyopt_amd as ta
def timeit(fn, reps=8):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts), sum(ts) / len(ts)
opts = ta.Options.benchmark()
for (P, n, m, dt) in ((10000, 12, 500, torch.float64), (10000, 12, 512, torch.float64), (2560, 12, 2000, torch.float64)):
    model, x0, _ = ta.DenseRow.synthetic(P, n, m, dt)
    x = x0.clone(); out = ta.Optimize(x, model, opts)
    def fused():
        x.copy_(x0); ta.Optimize(x, model, opts, out=out)
    os.environ["TOA_NO_AUTOSPLIT"] = "1"
    tf = timeit(fused)
    xs = x.clone()
    def team():
        x.copy_(x0); ta.Optimize(x, model, opts, out=out, splits=0)
    tt = timeit(team)
    print(f"P={P} n={n} m={m}: fused {tf[0]:.3f} ms  team(8 waves/WG) {tt[0]:.3f} ms  max|dx| {float((x-xs).abs().max()):.2e}", flush=True)
