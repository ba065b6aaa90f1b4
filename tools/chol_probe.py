#!/usr/bin/env python3
"""Time (H + lambda diag H) dx = g through toa_solve_damped at the two shapes whose pipelines wait on large_chol_solve_kernel:
n = 384 fp64 x 4 (the reduced camera system of `bench.py --workload balists`) and n = 256 fp32 x 128 (`--workload large256`).
Prints ms per call (wall clock over REPS calls, one sync), the residual against numpy's solve, and a checksum of the bits
(to compare binaries: TINYOPT_AMD_LIB=... python tools/chol_probe.py).  With a -DTOA_CHOL_TIMING build the kernel prints its phases."""
import hashlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinyopt_amd as ta  # noqa: E402


def spd(P, n, dtype, seed):
    rng = np.random.default_rng(seed)
    B = rng.uniform(-1, 1, (P, n, n + 8))
    H = (B @ B.transpose(0, 2, 1) / n + 0.5 * np.eye(n)).astype(dtype)
    g = rng.uniform(-1, 1, (P, n)).astype(dtype)
    return H, g


def main():
    reps = int(os.environ.get("REPS", "200"))
    shapes = [(np.float64, 384, 4), (np.float32, 256, 128), (np.float64, 200, 4), (np.float32, 500, 16), (np.float32, 1024, 8)]
    if os.environ.get("SHAPES"):
        shapes = shapes[:int(os.environ["SHAPES"])]
    for dtype, n, P in shapes:
        H, g = spd(P, n, dtype, n)
        Hd, gd = torch.from_numpy(H).cuda(), torch.from_numpy(g).cuda()
        dx, ok = ta.solve_damped(Hd, gd, 1.0 + 1e-4)
        torch.cuda.synchronize()
        Hn = H.astype(np.float64).copy()
        idx = np.arange(n)
        Hn[:, idx, idx] *= 1.0 + 1e-4
        ref = np.linalg.solve(Hn, g.astype(np.float64)[..., None])[..., 0]
        err = np.abs(dx.double().cpu().numpy() + ref).max() / np.abs(ref).max()
        for _ in range(10):
            ta.solve_damped(Hd, gd, 1.0 + 1e-4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ta.solve_damped(Hd, gd, 1.0 + 1e-4)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        digest = hashlib.sha1(dx.cpu().numpy().tobytes()).hexdigest()[:12]
        print(f"{np.dtype(dtype).name} n={n} P={P}: {ms * 1e3:8.1f} us per call  ok={ok.cpu().numpy().tolist()[:4]}  rel err {err:.2e}  bits {digest}", flush=True)


if __name__ == "__main__":
    main()
