"""CPU tests of the tie-aware trajectory comparator (tests/parity.py) itself: two runs of the ORACLE on the same
problems with the residual rows in a different order differ exactly the way the HIP path and the oracle do (same
mathematics, different summation order), so the comparator must accept them — and it must reject a trajectory whose
decisions differ away from the round-off floor."""
import numpy as np
import pytest

from parity import check_trajectories


def _run(oracle, A, b, x0, pod):
    r = oracle.dense_row_lm(A, b, x0, pod, history=True)
    return dict(errs=r["errs"], succ=r["succ"], iters=r["iters"], stop=r["stop"], x=r["x"], cost=r["cost"],
                fails=r["fails"], deltas2=r["deltas2"])


@pytest.mark.parametrize("dtype,n,m", [(np.float64, 12, 500), (np.float64, 50, 300), (np.float32, 50, 2000), (np.float32, 12, 500)])
@pytest.mark.parametrize("which", ["benchmark", "default"])
def test_row_permuted_oracle_runs_agree_up_to_ties(oracle, dtype, n, m, which):
    from tinyopt_amd.api import Options
    P = 48
    A, b, x0, _ = oracle.synth_dense_row(P, n, m, dtype, seed=4242)
    opts = Options.benchmark() if which == "benchmark" else Options()
    pod = opts.to_pod()
    perm = np.random.default_rng(1).permutation(m)
    a = _run(oracle, A, b, x0, pod)
    c = _run(oracle, np.ascontiguousarray(A[:, perm]), np.ascontiguousarray(b[:, perm]), x0, pod)
    st = check_trajectories(a, c, dtype, pod, label=f"{which}")
    assert st["full"] + st["ties"] == P
    # a tie can only occur once the cost has stopped decreasing: never in the first two iterations
    assert all(j >= 2 for j in st["tie_iters"]), st


def test_comparator_rejects_a_real_difference(oracle):
    from tinyopt_amd.api import Options
    A, b, x0, _ = oracle.synth_dense_row(8, 12, 200, np.float64, seed=7)
    pod = Options().to_pod()
    a = _run(oracle, A, b, x0, pod)
    c = {k: np.array(v, copy=True) for k, v in a.items()}
    c["succ"][3, 1] ^= 1                      # flip an accept/reject decision far from the floor
    with pytest.raises(AssertionError):
        check_trajectories(a, c, np.float64, pod)
    c = {k: np.array(v, copy=True) for k, v in a.items()}
    c["iters"][2] -= 1                        # one run stops an iteration early with nothing on a threshold
    with pytest.raises(AssertionError):
        check_trajectories(a, c, np.float64, pod)
    c = {k: np.array(v, copy=True) for k, v in a.items()}
    c["errs"][5, 1] *= 1.01                   # a different cost at the same iteration
    with pytest.raises(AssertionError):
        check_trajectories(a, c, np.float64, pod)
    c = {k: np.array(v, copy=True) for k, v in a.items()}
    c["stop"][0] = 5                          # identical trajectory, different StopReason
    with pytest.raises(AssertionError):
        check_trajectories(a, c, np.float64, pod)
