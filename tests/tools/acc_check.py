"""Debug helper: accumulate seam vs the oracle over a grid of shapes (prints max relative errors)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tinyopt_amd as ta
from oracle import pyoracle

def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))

for dtype in (np.float64, np.float32):
    for n in (16, 17, 18, 19, 34, 50, 12):
        for m in (16, 32, 64, 80, 96, 150):
            A, b, x0, xs = pyoracle.synth_dense_row(3, n, m, dtype, seed=5)
            g_ref, H_ref, c_ref, _ = pyoracle.dense_row_accumulate(A, b, x0)
            model = ta.DenseRow.from_arrays(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
            g, H, c, nres = ta.accumulate(model, torch.from_numpy(x0).cuda())
            c0 = ta.accumulate(model, torch.from_numpy(x0).cuda(), want_grad=False)[2]
            torch.cuda.synchronize()
            print(np.dtype(dtype).name, n, m, "g %.1e H %.1e c %.1e c0 %.1e" % (rel(g.cpu().numpy(), g_ref), rel(H.cpu().numpy(), H_ref),
                  rel(c.cpu().numpy(), c_ref), rel(c0.cpu().numpy(), c_ref)))
