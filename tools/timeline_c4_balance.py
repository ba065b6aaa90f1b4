import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import tinyopt_amd as ta
from tinyopt_amd.api import default_context
P = 12500
model, x0, _ = ta.DenseRow.synthetic(P, 50, 2000, torch.float32)
opts = ta.Options.benchmark()
x = x0.clone(); out = ta.Optimize(x, model, opts)
for _ in range(2):
    x.copy_(x0); ta.Optimize(x, model, opts, out=out)
torch.cuda.synchronize()
path = os.path.join(tempfile.mkdtemp(), "tl.txt")
ctx = default_context(); ctx.debug_timeline(path)
x.copy_(x0); ta.Optimize(x, model, opts, out=out); torch.cuda.synchronize(); ctx.debug_timeline(None)
rows = [tuple(int(v) for v in l.split()) for l in open(path) if not l.startswith("#")]
t = np.array(rows, dtype=np.float64)
t0 = t[:,0].min(); s=(t[:,0]-t0)*1e-2; e=(t[:,1]-t0)*1e-2; d=e-s
it = out.num_iters.cpu().numpy()
print("launch %.0f us; sum of problem durations / 3072 = %.0f us (a perfectly balanced finish); queue dry at %.0f us" % (e.max(), d.sum()/3072, np.sort(s)[-1]))
print("duration per iteration: mean %.1f us; by iterations:" % (d.sum()/it.sum()), {int(k): round(float(d[it==k].mean()),0) for k in np.unique(it)})
b = (np.arange(3072)//4)
print("first round by age group:", [round(float(d[:3072][(b//256)==g].mean()),0) for g in range(3)], "by XCD:", [round(float(d[:3072][(b%8)==g].mean()),0) for g in range(8)])
late = s > np.sort(s)[-1] - 1500
print("problems started in the last 1.5 ms before the queue ran dry: n=%d mean duration %.0f us; overall mean %.0f us" % (late.sum(), d[late].mean(), d.mean()))
