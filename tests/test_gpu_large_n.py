"""K3 for n > 63 (rocSOLVER batched Cholesky behind toa_solve_damped): parity with a float64
host solve of the same damped systems, and agreement with the one-wavefront path where both apply."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _spd_batch(P, n, dtype, seed):
    rng = np.random.default_rng(seed)
    J = rng.uniform(-1, 1, size=(P, 3 * n, n))
    H = np.einsum("pki,pkj->pij", J, J)
    g = rng.uniform(-1, 1, size=(P, n))
    return H.astype(dtype), g.astype(dtype)


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-9), (np.float32, 2e-3)])
@pytest.mark.parametrize("n", [64, 65, 100, 200])
def test_large_n_solve_matches_host(dtype, tol, n):
    import tinyopt_amd as ta
    P = 7
    H, g = _spd_batch(P, n, dtype, seed=n)
    scale = 1.0 + 1e-3
    dx, ok = ta.solve_damped(torch.from_numpy(H).cuda(), torch.from_numpy(g).cuda(), scale)
    torch.cuda.synchronize()
    assert ok.cpu().numpy().tolist() == [1] * P
    Hd = H.astype(np.float64).copy()
    idx = np.arange(n)
    Hd[:, idx, idx] = (H[:, idx, idx].astype(np.float64) * scale).astype(dtype).astype(np.float64)  # lm.h:108-117: scaled in double, stored in Scalar
    ref = -np.linalg.solve(Hd, g.astype(np.float64)[..., None])[..., 0]
    err = np.abs(dx.cpu().numpy() - ref).max() / np.abs(ref).max()
    assert err < tol, err


def test_large_n_not_positive_definite_is_a_solver_failure():
    import tinyopt_amd as ta
    P, n = 3, 96
    H, g = _spd_batch(P, n, np.float64, seed=5)
    H[1, 10, 10] = -1.0  # indefinite: gn.h:150-171 returns nullopt, the LM loop treats it as a failed solve
    dx, ok = ta.solve_damped(torch.from_numpy(H).cuda(), torch.from_numpy(g).cuda(), 1.0)
    torch.cuda.synchronize()
    assert ok.cpu().numpy().tolist() == [1, 0, 1]
    assert float(dx[1].abs().max()) == 0.0


def test_library_path_agrees_with_wavefront_path_below_64(monkeypatch):
    """Same systems through both implementations of K3 (the crossover measurement relies on them being interchangeable)."""
    import subprocess, sys, os, json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, json, numpy as np, torch; sys.path.insert(0, %r); import tinyopt_amd as ta\n"
        "rng = np.random.default_rng(3); n = 50; P = 5\n"
        "J = rng.uniform(-1, 1, size=(P, 150, n)); H = np.einsum('pki,pkj->pij', J, J); g = rng.uniform(-1, 1, size=(P, n))\n"
        "dx, ok = ta.solve_damped(torch.from_numpy(H).cuda(), torch.from_numpy(g).cuda(), 1.0001)\n"
        "print(json.dumps({'dx': dx.cpu().numpy().tolist(), 'ok': ok.cpu().numpy().tolist()}))\n" % root)
    outs = []
    for force in ("0", "1"):
        env = dict(os.environ, TOA_FORCE_ROCSOLVER=force)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    a, b = np.array(outs[0]["dx"]), np.array(outs[1]["dx"])
    assert outs[0]["ok"] == outs[1]["ok"] == [1] * 5
    assert np.abs(a - b).max() / np.abs(a).max() < 1e-10
