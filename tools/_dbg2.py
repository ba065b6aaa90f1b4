import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import tinyopt_amd as ta
from oracle import pyoracle as oracle
oracle.load()
from test_gpu_row_models import manual_body, _items
os.makedirs("/root/repo/gpurun_out/jitcache", exist_ok=True)
ta.JitResidual.set_cache_dir("/root/repo/gpurun_out/jitcache")
for (n, m, dtype) in ((50, 64, np.float64),):
    tdt = torch.float64
    P = 1
    A, b, x0, xs = oracle.synth_dense_row(P, n, m, dtype, seed=700 + n + m)
    fit = ta.JitResidual(manual_body(n), n=n, item_scalars=n + 1, dtype=tdt, kind="accumulate")
    model = fit.bind(_items(A, b))
    x = torch.from_numpy(x0.copy()).cuda()
    g, H, c, nres = ta.accumulate(model, x)
    g_ref, H_ref, c_ref, _ = oracle.dense_row_accumulate(A, b, x0)
    print("g", g.cpu().numpy()[0][:8], g_ref[0][:8])
    print("c", c.cpu().numpy(), c_ref)
    Hd = H.cpu().numpy()[0]
    print("nan count H", np.isnan(Hd).sum(), "of", Hd.size, " g nan", np.isnan(g.cpu().numpy()).sum())
    # per-half check: the first 32 rows alone
    A2, b2 = A[:, :32].copy(), b[:, :32].copy()
    g2, H2, c2, _ = ta.accumulate(fit.bind(_items(A2, b2)), x)
    g2r, H2r, c2r, _ = oracle.dense_row_accumulate(A2, b2, x0)
    print("first 32 rows ok:", np.abs(g2.cpu().numpy() - g2r).max())
    A3, b3 = A[:, 32:].copy(), b[:, 32:].copy()
    g3, H3, c3, _ = ta.accumulate(fit.bind(_items(A3, b3)), x)
    g3r, H3r, c3r, _ = oracle.dense_row_accumulate(A3, b3, x0)
    print("last 32 rows ok:", np.abs(g3.cpu().numpy() - g3r).max())
