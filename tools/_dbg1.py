import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import tinyopt_amd as ta
from oracle import pyoracle as oracle
oracle.load()
from test_gpu_row_models import manual_body, _items
for (n, m, dtype) in ((50, 130, np.float64), (50, 128, np.float64), (50, 32, np.float64), (49, 130, np.float64), (51,130,np.float64), (34,130,np.float64)):
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    P = 3
    A, b, x0, xs = oracle.synth_dense_row(P, n, m, dtype, seed=700 + n + m)
    fit = ta.JitResidual(manual_body(n), n=n, item_scalars=n + 1, dtype=tdt, kind="accumulate")
    model = fit.bind(_items(A, b))
    x = torch.from_numpy(x0.copy()).cuda()
    g, H, c, nres = ta.accumulate(model, x)
    g_ref, H_ref, c_ref, _ = oracle.dense_row_accumulate(A, b, x0)
    eg = np.abs(g.cpu().numpy() - g_ref).max() / np.abs(g_ref).max()
    eH = np.abs(H.cpu().numpy() - H_ref)
    print(n, m, "g", eg, "H", eH.max() / np.abs(H_ref).max(), "c", np.abs(c.cpu().numpy() - c_ref).max() / c_ref.max())
    bad = np.argwhere(eH[0] > 1e-8 * np.abs(H_ref).max())
    print("  bad H entries of problem 0:", len(bad), bad[:12].tolist())
