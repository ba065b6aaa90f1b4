import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import tinyopt_amd as ta
from tinyopt_amd.api import default_context
P = int(sys.argv[1]); outp = sys.argv[2]
model, x0, _ = ta.DenseRow.synthetic(P, 12, 500, torch.float64)
opts = ta.Options.benchmark()
x = x0.clone(); out = ta.Optimize(x, model, opts)
for _ in range(3):
    x.copy_(x0); ta.Optimize(x, model, opts, out=out)
torch.cuda.synchronize()
path = os.path.join(tempfile.mkdtemp(), "tl.txt")
ctx = default_context(); ctx.debug_timeline(path)
x.copy_(x0); ta.Optimize(x, model, opts, out=out); torch.cuda.synchronize(); ctx.debug_timeline(None)
rows = [tuple(int(v) for v in l.split()) for l in open(path) if not l.startswith("#")]
np.save(outp, np.array(rows, dtype=np.int64))
