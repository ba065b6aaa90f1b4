import sys, os
sys.path.insert(0, os.getcwd())
import torch, tinyopt_amd as ta
wl = sys.argv[1]
P, n, m, dt = (12500, 50, 2000, torch.float32) if wl == "c4" else (10000, 12, 500, torch.float64)
model, x0, _ = ta.DenseRow.synthetic(P, n, m, dt)
opts = ta.Options.benchmark()
x = x0.clone(); out = ta.Optimize(x, model, opts)
for _ in range(3):
    x.copy_(x0); ta.Optimize(x, model, opts, out=out)
torch.cuda.synchronize()
