#!/usr/bin/env python3
"""K3 crossover: the one-wavefront register LDL^T against rocSOLVER's batched Cholesky
(potrf + potrs, strided batched) on the same damped systems, n = 8 .. 63, and the library path alone beyond.
Each implementation runs in its own process (toa_tuning::large_library_solver selects the library).

usage: python tools/k3_crossover.py            (prints a markdown table)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, json, numpy as np, torch
sys.path.insert(0, %r)
import tinyopt_amd as ta
res = {}
for dt in (torch.float32, torch.float64):
    for n in %r:
        P = max(64, min(16384, int(2e8 // (n * n * 8))))
        g = torch.rand(P, n, dtype=dt, device='cuda') - 0.5
        J = torch.rand(P, 2 * n, n, dtype=dt, device='cuda') - 0.5
        H = torch.bmm(J.transpose(1, 2), J) + 0.01 * n * torch.eye(n, dtype=dt, device='cuda')   # safely definite in fp32 too
        ta.solve_damped(H, g, 1.0001); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); dx, ok = ta.solve_damped(H, g, 1.0001); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        assert int(ok.sum()) == P
        res[f"{'f32' if dt == torch.float32 else 'f64'}:{n}"] = (P, min(ts))
print(json.dumps(res))
"""


def run(force, ns):
    child = (CHILD % (ROOT, ns)).replace("import tinyopt_amd as ta", "import tinyopt_amd as ta\nta.api.default_context().set_tuning(large_library_solver=%s)" % force, 1)
    r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-3000:])
        raise SystemExit(1)
    return json.loads(r.stdout.strip().splitlines()[-1])


def main():
    small = [8, 12, 16, 24, 32, 48, 50, 63]
    large = [64, 96, 128, 256]
    wave = run("0", small + [64, 96, 128])   # 64..128: the workgroup LDL^T (ldlt_wg.hpp)
    lib = run("1", small + large)
    print("| dtype | n | matrices | own kernel (n <= 63: one wavefront per matrix; 64..128: one workgroup): ns / solve | rocSOLVER potrf+potrs batched: ns / solve | ratio |")
    print("|---|---|---|---|---|---|")
    for dt in ("f32", "f64"):
        for n in small + large:
            k = f"{dt}:{n}"
            P, tl = lib[k]
            if k in wave:
                _, tw = wave[k]
                print(f"| {dt} | {n} | {P} | {tw * 1e6 / P:.0f} | {tl * 1e6 / P:.0f} | {tl / tw:.1f}x |")
            else:
                print(f"| {dt} | {n} | {P} | - | {tl * 1e6 / P:.0f} | |")


if __name__ == "__main__":
    main()
