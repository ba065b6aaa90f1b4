import sys, os
sys.path.insert(0, os.getcwd())
import torch, tinyopt_amd as ta
sys.path.insert(0, 'tools')
from large_n_bench import synth
for dt, n, m, P in ((torch.float32, 128, 4096, 512), (torch.float32, 64, 2000, 2048)):
    A, b, x0, xs = synth(P, n, m, dt)
    model = ta.DenseRowNatural(A, b)
    x = x0.clone()
    out = ta.Optimize(x, model, ta.Options.benchmark())
    x.copy_(x0)
    out = ta.Optimize(x, model, ta.Options.benchmark())
    torch.cuda.synchronize()
