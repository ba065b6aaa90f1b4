"""GPU parity tests proper: the HIP path (through the C-ABI) vs the CPU oracle on the same seeded
inputs.  Tolerances: fp64 rel 1e-10 on g/H/cost, 1e-8 on final x, identical
StopReason / iteration counts; fp32 rel 1e-4 on g/H/cost (math.h:297-301 FloatEpsilon class)."""
import numpy as np
import pytest
import torch

from parity import check_trajectories, gpu_dict

pytestmark = pytest.mark.gpu


def _ref_dict(ref):
    return dict(errs=ref["errs"], succ=ref["succ"], iters=ref["iters"], stop=ref["stop"], x=ref["x"], cost=ref["cost"],
                fails=ref["fails"], deltas2=ref["deltas2"])


def _rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


ACC_CASES = [
    # dtype, n, m, P          (BASELINE configs C2, C3, C4 shapes + ragged / edge shapes)
    (np.float64, 6, 1000, 3),
    (np.float64, 12, 500, 9),
    (np.float32, 50, 2000, 5),
    (np.float64, 1, 1, 4),
    (np.float64, 1, 7, 4),
    (np.float32, 3, 10, 6),
    (np.float64, 15, 33, 5),   # n+1 == 16: exactly one block
    (np.float64, 16, 35, 5),   # NB = 2
    (np.float32, 31, 101, 5),  # NB = 2 full
    (np.float32, 33, 64, 5),   # NB = 3
    (np.float64, 47, 97, 3),   # NB = 3 full
    (np.float64, 50, 203, 3),  # NB = 4, m % 4 != 0
    (np.float32, 63, 130, 3),  # largest n
    (np.float32, 16, 40, 4),   # thin tail: b only
    (np.float64, 17, 40, 4),   # thin tail: 1 column + b
    (np.float32, 35, 77, 4),   # NB = 2 + thin 4
    (np.float64, 48, 90, 3),   # NB = 3 + thin 1
    (np.float64, 50, 90, 3),   # NB = 3 + thin 3 (fp64)
    (np.float32, 51, 90, 3),   # NB = 3 + thin 4
]


@pytest.mark.parametrize("dtype,n,m,P", ACC_CASES)
def test_accumulate_matches_oracle(ta, oracle, dtype, n, m, P):
    A, b, x0, _ = oracle.synth_dense_row(P, n, m, dtype, seed=1234)
    g_ref, H_ref, c_ref, nres_ref = oracle.dense_row_accumulate(A, b, x0)
    model = ta.DenseRow.from_arrays(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
    x = torch.from_numpy(x0).cuda()
    g, H, c, nres = ta.accumulate(model, x, want_grad=True)
    torch.cuda.synchronize()
    tol = 1e-10 if dtype == np.float64 else 1e-4
    for p in range(P):
        assert _rel(g[p].cpu().numpy(), g_ref[p]) < tol
        Hg = H[p].cpu().numpy()
        assert _rel(Hg, H_ref[p]) < tol
        assert np.array_equal(Hg, Hg.T), "device H must be exactly symmetric"
    assert _rel(c.cpu().numpy(), c_ref) < tol
    assert (nres.cpu().numpy() == nres_ref).all()
    # cost-only call (grad == nullptr)
    _, _, c2, nres2 = ta.accumulate(model, x, want_grad=False)
    assert _rel(c2.cpu().numpy(), c_ref) < tol
    assert (nres2.cpu().numpy() == nres_ref).all()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n", [1, 2, 5, 12, 33, 50, 63])
def test_solve_damped_spd(ta, oracle, dtype, n):
    rng = np.random.default_rng(n)
    P = 7
    J = rng.uniform(-1, 1, (P, 3 * n + 2, n))
    H = np.einsum("pij,pik->pjk", J, J).astype(dtype)
    g = rng.uniform(-1, 1, (P, n)).astype(dtype)
    for scale in (1.0, 1.0001, 3.0):
        dx_ref, ok_ref = oracle.solve_damped(H, g, scale)
        dx, ok = ta.solve_damped(torch.from_numpy(H).cuda(), torch.from_numpy(g).cuda(), scale)
        torch.cuda.synchronize()
        assert (ok.cpu().numpy() == ok_ref).all() and ok_ref.all()
        tol = 1e-9 if dtype == np.float64 else 2e-3   # conditioning-dependent
        assert _rel(dx.cpu().numpy(), dx_ref) < tol
        # residual check independent of the oracle: (H_damped) dx = -g
        Hd = H.astype(np.float64).copy()
        idx = np.arange(n)
        Hd[:, idx, idx] = (Hd[:, idx, idx] * scale).astype(dtype)
        r = np.einsum("pij,pj->pi", Hd, dx.cpu().numpy().astype(np.float64)) + g
        assert np.abs(r).max() < (1e-9 if dtype == np.float64 else 2e-3)


def test_solve_damped_acceptance_rule(ta, oracle):
    """math.h:236: fail iff info()!=Success || !isPositive(); semi-definite passes (pseudo-inverse)."""
    mats = [
        np.diag([-1.0, -2.0, -3.0]),            # negative definite -> fail
        np.diag([1.0, -2.0, 3.0]),              # indefinite -> fail
        np.diag([2.0, 0.0, 5.0]),               # zero pivot, semi-definite -> ok, that component 0
        np.zeros((3, 3)),                       # all zero -> ok, dx = 0
        np.array([[0, 1, 0], [1, 0, 0], [0, 0, 0.0]]),  # zero diagonal, non-zero off-diagonal -> fail
        np.array([[4, 2, 0.6], [2, 2, 0.5], [0.6, 0.5, 3.0]]),  # SPD
        np.array([[1, 2, 0], [2, 1, 0], [0, 0, 1.0]]),  # indefinite via off-diagonals -> fail
        np.array([[1e-30, 0, 0], [0, 1, 0], [0, 0, 1e30]]),  # badly scaled SPD -> pivoting
    ]
    H = np.stack(mats)
    g = np.tile(np.array([1.0, -2.0, 0.5]), (len(mats), 1))
    dx_ref, ok_ref = oracle.solve_damped(H, g, 1.0)
    dx, ok = ta.solve_damped(torch.from_numpy(H).cuda(), torch.from_numpy(g).cuda(), 1.0)
    torch.cuda.synchronize()
    assert list(ok_ref) == [0, 0, 1, 1, 0, 1, 0, 1]
    assert (ok.cpu().numpy() == ok_ref).all()
    good = ok_ref == 1
    assert np.allclose(dx.cpu().numpy()[good], dx_ref[good], rtol=1e-12, atol=1e-300)


LM_CASES = [
    (np.float64, 12, 500, 40),
    (np.float64, 6, 1000, 8),
    (np.float64, 1, 5, 8),
    (np.float64, 16, 60, 16),
    (np.float64, 50, 300, 6),
    (np.float32, 50, 2000, 12),
    (np.float32, 12, 500, 16),
    (np.float64, 18, 80, 8),    # thin tail
    (np.float32, 34, 150, 8),   # thin tail
    # fp64 n <= 15: the fused kernel's row-per-lane pass through LDS (pass16s) — odd and even row widths (even ones read their
    # rows with LDS bank conflicts, not with different results), the full 16 columns, rows short of / past a 64-row super-batch
    (np.float64, 2, 70, 8),
    (np.float64, 7, 129, 8),
    (np.float64, 13, 64, 8),
    (np.float64, 14, 333, 8),
    (np.float64, 15, 200, 8),
]


@pytest.mark.parametrize("dtype,n,m,P", LM_CASES)
@pytest.mark.parametrize("which", ["benchmark", "default"])
def test_lm_run_matches_oracle(ta, oracle, dtype, n, m, P, which):
    A, b, x0, xs = oracle.synth_dense_row(P, n, m, dtype, seed=99)
    opts = ta.Options.benchmark() if which == "benchmark" else ta.Options()
    opts.hessian.save_last = True
    ref = oracle.dense_row_lm(A, b, x0, opts.to_pod(), history=True)
    model = ta.DenseRow.from_arrays(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
    x = torch.from_numpy(x0.copy()).cuda()
    out = ta.Optimize(x, model, opts, history=True)
    torch.cuda.synchronize()
    xg = x.cpu().numpy()
    stop = out.stop_reason.cpu().numpy()
    iters = out.num_iters.cpu().numpy()
    assert (stop >= 0).all() and (ref["stop"] >= 0).all()
    # trajectory parity, fp64 AND fp32: cost / accept-reject identical iteration by iteration, StopReason / iteration
    # counts / failure counts identical unless the runs part at a decision PROVEN to sit on the round-off floor
    # (tests/parity.py; optimizer.h:428-460, 519-534)
    st = check_trajectories(gpu_dict(out, x), _ref_dict(ref), dtype, opts.to_pod(), label=f"{n}x{m} {which}")
    assert st["full"] + st["ties"] == P
    assert all(j >= 2 or n == 1 for j in st["tie_iters"]), st     # never before the cost has stopped decreasing
    if dtype == np.float64:
        Hf = out.final_hessian.cpu().numpy()
        assert _rel(Hf, ref["H"]) < 1e-9
    assert iters.max() <= opts.max_iters + 1
    # planted solution recovered to the noise level (independent of the oracle)
    assert np.abs(xg - xs).max() < 5e-3


def test_lm_counters_and_history_shapes(ta, oracle):
    A, b, x0, _ = oracle.synth_dense_row(10, 12, 500, np.float64, seed=5)
    opts = ta.Options.benchmark()
    model = ta.DenseRow.from_arrays(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
    x = torch.from_numpy(x0.copy()).cuda()
    out = ta.Optimize(x, model, opts, history=True)
    torch.cuda.synchronize()
    cnt = out.counters.cpu().numpy()
    iters = out.num_iters.cpu().numpy()
    assert cnt[3] == 10
    assert cnt[0] + cnt[1] >= iters.sum()          # every iteration runs at least one pass
    assert out.final_hessian is None               # benchmark options: save_last = false
    # history entries beyond num_iters stay zero (reference: vectors have exactly num_iters entries)
    errs = out.errs.cpu().numpy()
    for p in range(10):
        assert (errs[p, iters[p]:] == 0).all()


def test_synth_device_matches_oracle(ta, oracle):
    for dtype, tdt, n, m in ((np.float64, torch.float64, 12, 50), (np.float32, torch.float32, 50, 37),
                             (np.float32, torch.float32, 33, 9)):
        P = 6
        A, b, x0, xs = oracle.synth_dense_row(P, n, m, dtype, seed=77, problem0=1000)
        model, x0d, xsd = ta.DenseRow.synthetic(P, n, m, tdt, seed=77, problem0=1000)
        ref_model = ta.DenseRow.from_arrays(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
        torch.cuda.synchronize()
        lay = ta.api.dense_row_layout(tdt, n, m)
        got = model.packed.cpu().numpy().reshape(P, lay["rows_padded"], lay["row_stride"])
        want = ref_model.packed.cpu().numpy().reshape(P, lay["rows_padded"], lay["row_stride"])
        # the packer itself: every A / b entry sits where the layout says, everything else is zero
        assert np.array_equal(want[:, :m, lay["pos_cols"]], A)
        assert np.array_equal(want[:, :m, lay["pos_b"]], b)
        mask = np.ones(lay["row_stride"], bool)
        mask[lay["pos_cols"]] = False
        mask[lay["pos_b"]] = False
        assert not want[:, :, mask].any() and not want[:, m:, :].any()
        # device generator vs oracle generator
        assert np.array_equal(got[:, :, lay["pos_cols"]], want[:, :, lay["pos_cols"]]), "A must be bit-identical"
        assert np.allclose(got[:, :, lay["pos_b"]], want[:, :, lay["pos_b"]], rtol=0,
                           atol=1e-6 if dtype == np.float32 else 1e-14)
        assert not got[:, :, mask].any()
        assert np.array_equal(x0d.cpu().numpy(), x0)
        assert np.allclose(xsd.cpu().numpy(), xs.astype(dtype))


def test_option_variants(ta, oracle):
    A, b, x0, xs = oracle.synth_dense_row(12, 12, 200, np.float64, seed=3)
    model = ta.DenseRow.from_arrays(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
    variants = []
    o = ta.Options(); o.solver_type = ta.Options.GaussNewton; variants.append(o)
    o = ta.Options(); o.cost.downscale_by_2 = True; variants.append(o)
    o = ta.Options(); o.cost.normalize = True; variants.append(o)
    o = ta.Options(); o.cost.use_squared_norm = False; variants.append(o)
    o = ta.Options(); o.check_final_cost = True; o.max_iters = 3; variants.append(o)
    o = ta.Options(); o.max_iters = 1; variants.append(o)
    o = ta.Options(); o.lm.damping_init = 10.0; variants.append(o)
    o = ta.Options(); o.use_step_quality_approx = True; variants.append(o)
    o = ta.Options(); o.grad_clipping = 0.5; o.max_iters = 5; variants.append(o)
    o = ta.Options(); o.hessian.check_min_H_diag = 1e9; variants.append(o)   # forces Build failure -> kSolverFailed
    o = ta.Options(); o.min_error = 1e3; variants.append(o)                  # immediate kMinError
    o = ta.Options(); o.max_total_failures = 1; o.max_consec_failures = 0; o.min_step_norm2 = 0; o.min_rerr_dec = 0
    o.min_grad_norm2 = 0; o.min_error = 0; o.max_iters = 30; variants.append(o)
    for i, o in enumerate(variants):
        ref = oracle.dense_row_lm(A, b, x0, o.to_pod(), history=True)
        x = torch.from_numpy(x0.copy()).cuda()
        out = ta.Optimize(x, model, o, history=True)
        torch.cuda.synchronize()
        # every option branch: the same trajectory as the oracle, parting only at proven ties (the last variant, with
        # every convergence test disabled, ends at the FIRST round-off-level cost increase: a tie by construction)
        st = check_trajectories(gpu_dict(out, x), _ref_dict(ref), np.float64, o.to_pod(), label=f"variant {i}")
        assert st["full"] + st["ties"] == 12
        assert np.abs(x.cpu().numpy() - ref["x"]).max() < 1e-6, f"variant {i}"
        assert _rel(out.final_cost.cpu().numpy(), ref["cost"]) < 1e-8, f"variant {i}"


def test_argument_errors(ta):
    model, x0, _ = ta.DenseRow.synthetic(4, 12, 50, torch.float64)
    with pytest.raises(TypeError):
        ta.Optimize(x0, lambda x: x)            # host callable cannot run on the GPU path
    with pytest.raises(ValueError):
        ta.Optimize(x0[:, :5].contiguous(), model)
    o = ta.Options(); o.solver_type = 2         # GradientDescent is not on this path (optimize.h:75 throws)
    with pytest.raises(ta.ToaError):
        ta.Optimize(x0, model, o)


def test_use_ldlt_false_unchecked_inverse(ta, oracle):
    """gn.h:157-162: with use_ldlt = false the step is -H.inverse() * g, unchecked.  Same trajectories as the oracle's LU
    inverse (n = 12 fp64, n = 50 fp32), through the fused, the row-split and the stepping form."""
    for dtype, n, m, P in ((np.float64, 12, 200, 10), (np.float32, 50, 600, 6)):
        A, b, x0, xs = oracle.synth_dense_row(P, n, m, dtype, seed=5)
        model = ta.DenseRow.from_arrays(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
        for o in (ta.Options(), ta.Options.benchmark()):
            o.hessian.use_ldlt = False
            ref = oracle.dense_row_lm(A, b, x0, o.to_pod(), history=True)
            for splits in (0, 4):
                x = torch.from_numpy(x0.copy()).cuda()
                out = ta.Optimize(x, model, o, history=True, splits=splits)
                torch.cuda.synchronize()
                st = check_trajectories(gpu_dict(out, x), _ref_dict(ref), dtype, o.to_pod(), label=f"use_ldlt=false n={n} splits={splits}")
                assert st["full"] + st["ties"] == P, st
                assert np.abs(x.cpu().numpy() - xs).max() < 2e-2


def test_failure_paths_match_reference_semantics(ta, oracle):
    """tests/basic.cpp:147-218: NaN / Inf in residuals or Jacobians -> kSystemHasNaNOrInf, at most one iteration,
    EMPTY history; healthy problems in the same batch are unaffected (per-problem divergence)."""
    P, n, m = 6, 12, 40
    A, b, x0, xs = oracle.synth_dense_row(P, n, m, np.float64, seed=8)
    b[1, 3] = np.nan          # NaN residual
    A[2, 5, 7] = np.inf       # Inf in the Jacobian
    b[4, 0] = -np.inf         # Inf residual
    o = ta.Options()
    ref = oracle.dense_row_lm(A, b, x0, o.to_pod(), history=True)
    model = ta.DenseRow.from_arrays(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
    x = torch.from_numpy(x0.copy()).cuda()
    out = ta.Optimize(x, model, o, history=True)
    torch.cuda.synchronize()
    stop = out.stop_reason.cpu().numpy()
    iters = out.num_iters.cpu().numpy()
    assert list(ref["stop"][[1, 2, 4]]) == [-2, -2, -2]
    assert np.array_equal(stop, ref["stop"]) and np.array_equal(iters, ref["iters"])
    for p in (1, 2, 4):
        assert iters[p] <= 1                                           # FailureChecks: num_iters <= max_iters(1)
        assert not out.errs.cpu().numpy()[p].any()                     # errs / deltas2 / successes empty
        assert not out.successes.cpu().numpy()[p].any()
        assert np.array_equal(x.cpu().numpy()[p], x0[p])               # x untouched
    for p in (0, 3, 5):
        assert stop[p] >= 1 and np.abs(x.cpu().numpy()[p] - xs[p]).max() < 5e-3


def test_tiny_and_empty_batches(ta, oracle):
    """Ragged edges: fewer rows than one MFMA step (m < 4), m < n (rank-deficient -> LM damping still solves),
    a single problem, and an empty batch."""
    for n, m in ((3, 1), (3, 2), (5, 3), (12, 7)):
        A, b, x0, _ = oracle.synth_dense_row(5, n, m, np.float64, seed=n * 10 + m)
        o = ta.Options()
        ref = oracle.dense_row_lm(A, b, x0, o.to_pod())
        model = ta.DenseRow.from_arrays(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
        x = torch.from_numpy(x0.copy()).cuda()
        out = ta.Optimize(x, model, o)
        torch.cuda.synchronize()
        stop = out.stop_reason.cpu().numpy()
        # rank-deficient normal matrices: same verdict class (success / solver failure) as the reference algorithm
        assert np.array_equal(stop >= 0, ref["stop"] >= 0), (n, m, stop, ref["stop"])
        good = (stop >= 0) & (stop == ref["stop"])
        if good.any():
            assert np.allclose(out.final_cost.cpu().numpy()[good], ref["cost"][good], rtol=1e-6, atol=1e-12)
    model, x0, _ = ta.DenseRow.synthetic(1, 12, 50, torch.float64)
    out = ta.Optimize(x0.clone(), model)
    assert out.stop_reason.shape == (1,) and int(out.stop_reason[0]) >= 1
    lib = ta.load()
    ctx = ta.api.default_context()
    import ctypes as C
    pod = ta.Options().to_pod()
    res = ta.ToaResults()
    dummy = torch.zeros(1, dtype=torch.int32, device="cuda")
    res.stop_reason = res.num_iters = dummy.data_ptr()
    res.final_cost = torch.zeros(1, dtype=torch.float64, device="cuda").data_ptr()
    assert lib.toa_lm_run(ctx.h, 1, 1, 12, 50, 0, model.packed.data_ptr(), x0.data_ptr(), C.byref(pod), C.byref(res), None) == 0


def test_solver_one_shot_reference(ta):
    """tests/solvers.cpp:74-110 (`SolverLM<Mat2>`: grad = x - y with y = (4, 5), H = I, x = 0): Build's Marquardt damping
    H_ii *= (1 + lambda_0) followed by Solve gives dx == y to 1e-2."""
    H = torch.eye(2, dtype=torch.float64, device="cuda")[None]
    g = torch.tensor([[-4.0, -5.0]], dtype=torch.float64, device="cuda")
    dx, ok = ta.solve_damped(H, g, 1.0 + 1e-4)      # options.h:133 damping_init = 1e-4
    torch.cuda.synchronize()
    assert int(ok[0]) == 1
    assert abs(float(dx[0, 0]) - 4.0) < 1e-2 and abs(float(dx[0, 1]) - 5.0) < 1e-2
    assert abs(float(dx[0, 0]) - 4.0 / 1.0001) < 1e-12


@pytest.mark.parametrize("dtype,tdt", [(np.float64, torch.float64), (np.float32, torch.float32)])
def test_every_n_layout_accumulate_and_solve(ta, oracle, dtype, tdt):
    """Every supported n (1..63: all MFMA block counts, every thin-tail width, b in the main block or in the tail) at
    ragged row counts: the Accumulate seam and a short LM solve against the oracle."""
    tol = 1e-10 if dtype == np.float64 else 3e-5
    opts = ta.Options.benchmark()
    for n in range(1, 64):
        for m in (n + 3, 4 * n + 37 if n % 7 == 0 else 2 * n + 5):
            P = 3
            A, b, x0, xs = oracle.synth_dense_row(P, n, m, dtype, seed=1000 + n)
            g_ref, H_ref, c_ref, _ = oracle.dense_row_accumulate(A, b, x0)
            model = ta.DenseRow.from_arrays(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
            g, H, c, nres = ta.accumulate(model, torch.from_numpy(x0).cuda())
            assert np.abs(g.cpu().numpy() - g_ref).max() <= tol * np.abs(g_ref).max(), (n, m)
            assert np.abs(H.cpu().numpy() - H_ref).max() <= tol * np.abs(H_ref).max(), (n, m)
            assert np.allclose(c.cpu().numpy(), c_ref, rtol=tol), (n, m)
            assert (nres.cpu().numpy() == m).all()
        if n % 3 == 0 or n in (1, 2, 16, 17, 31, 32, 47, 48, 49, 50, 63):
            m = 6 * n + 11
            A, b, x0, xs = oracle.synth_dense_row(4, n, m, dtype, seed=7 * n)
            ref = oracle.dense_row_lm(A, b, x0, opts.to_pod())
            model = ta.DenseRow.from_arrays(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
            x = torch.from_numpy(x0.copy()).cuda()
            out = ta.Optimize(x, model, opts)
            torch.cuda.synchronize()
            assert (out.stop_reason.cpu().numpy() >= 0).all() and (ref["stop"] >= 0).all(), n
            assert np.abs(x.cpu().numpy() - ref["x"]).max() < (1e-8 if dtype == np.float64 else 3e-3), n


@pytest.mark.parametrize("scale", [1e2, 1e3, 1e4])
@pytest.mark.parametrize("n,m", [(50, 2000), (12, 500), (33, 300)])
def test_fp32_pass_at_large_arguments(ta, oracle, scale, n, m):
    """VERDICT r05 "weak" #1 / "next" #4: the compiled-in fp32 pass evaluates sin / cos with v_sin_f32 / v_cos_f32 (inputs in
    revolutions), whose absolute error grows with |t| while libm's does not, and every other parity test lives at |a_i . x| < ~50.
    Here |a_i . x| is swept up to `scale` (the rows are unit-range, x is scaled).  Yardstick: the SAME fp32 data evaluated in
    float64 (numpy).  At |t| ~ 10^4 one fp32 ulp of t itself is 10^-3, so the fp32 CPU oracle — libm sin, sequential sum —
    is itself only that close to the float64 value; the device must be no further from it than a small multiple of the oracle's
    own distance (+ the fp32 tolerance class of SURVEY §8c, math.h:297-301), for the cost, J^T r and J^T J of an Accumulate pass,
    for the cost-only pass, and for the run-time row model's polynomial-free path (same functor as text).  Then an LM solve from a
    start near the planted solution recovers it."""
    P = 4
    rng = np.random.default_rng(int(scale) + n)
    A = rng.uniform(-1, 1, (P, m, n)).astype(np.float32)
    xs = (scale / np.sqrt(n / 3.0) * rng.uniform(-1, 1, (P, n))).astype(np.float32)          # |a . x*| of the order of `scale`
    t64 = np.einsum("pmn,pn->pm", A.astype(np.float64), xs.astype(np.float64))
    assert np.abs(t64).max() > 0.8 * scale
    b = (t64 + 0.1 * np.sin(t64)).astype(np.float32)
    x = (xs + rng.uniform(-1, 1, xs.shape) * 0.05).astype(np.float32)                        # residuals of order 0.1 .. 1

    def f64(xv):
        t = np.einsum("pmn,pn->pm", A.astype(np.float64), xv.astype(np.float64))
        r = t + 0.1 * np.sin(t) - b.astype(np.float64)
        J = (1 + 0.1 * np.cos(t))[..., None] * A.astype(np.float64)
        return np.einsum("pmn,pm->pn", J, r), np.einsum("pmn,pmk->pnk", J, J), (r * r).sum(1)
    g64, H64, c64 = f64(x)
    g_o, H_o, c_o, _ = oracle.dense_row_accumulate(A, b, x)
    model = ta.DenseRow.from_arrays(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
    g, H, c, _ = ta.accumulate(model, torch.from_numpy(x).cuda())
    c0 = ta.accumulate(model, torch.from_numpy(x).cuda(), want_grad=False)[2]
    torch.cuda.synchronize()
    for name, dev, orc, ref in (("g", g.cpu().numpy(), g_o, g64), ("H", H.cpu().numpy(), H_o, H64), ("cost", c.cpu().numpy(), c_o, c64),
                                ("cost-only", c0.cpu().numpy(), c_o, c64)):
        e_dev, e_orc = _rel(dev, ref), _rel(orc, ref)
        assert e_dev <= 4 * e_orc + 2e-4, (name, scale, n, e_dev, e_orc)
    # the solve: the planted solution is recovered to the accuracy fp32 allows at this magnitude of x
    opts = ta.Options.benchmark()
    xg = torch.from_numpy(x.copy()).cuda()
    out = ta.Optimize(xg, model, opts)
    torch.cuda.synchronize()
    assert bool((out.stop_reason >= 0).all())
    ref = oracle.dense_row_lm(A, b, x, opts.to_pod())
    err_dev = np.abs(xg.cpu().numpy().astype(np.float64) - xs).max()
    err_orc = np.abs(ref["x"].astype(np.float64) - xs).max()
    assert err_dev <= 4 * err_orc + 1e-6 * scale, (scale, n, err_dev, err_orc)


@pytest.mark.parametrize("dtype,n", [(np.float32, n) for n in range(1, 12)] + [(np.float64, n) for n in range(1, 7)])
def test_narrow_blocks_take_the_item_per_lane_routes(ta, oracle, dtype, n):
    """Round 6: narrow blocks of TOA_MODEL_DENSE_ROW run an item per lane — the Gram in registers (JetModel over the packed rows: fp32
    n <= 10, fp64 n <= 5) or staged into the MFMA Gram (RowModel, fp32 n = 11) — instead of sixteen lanes per row.  Whole trajectories
    against the oracle on ragged row counts (the packed layout pads rows to a multiple of four: the routes must not count them), the
    seam against the oracle, and both against the old route (toa_tuning::narrow_mfma_pass)."""
    P = 70
    f32 = dtype == np.float32
    tol = 3e-5 if f32 else 1e-10
    for m in (4 * n + 202, 333, 1000):   # (ragged: 4 n + 202 is 2 mod 4, 333 is 1 mod 4; fp32 problems of a few dozen rows stop on straddled thresholds)
        A, b, x0, xs = oracle.synth_dense_row(P, n, m, dtype, seed=300 + n + m)
        opts = ta.Options.benchmark()
        ref = oracle.dense_row_lm(A, b, x0, opts.to_pod(), history=True)
        model = ta.DenseRow.from_arrays(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
        x = torch.from_numpy(x0.copy()).cuda()
        out = ta.Optimize(x, model, opts, history=True)
        torch.cuda.synchronize()
        if n >= 4 or not f32:
            st = check_trajectories(gpu_dict(out, x), ref, dtype, opts.to_pod(), label=f"narrow route n = {n}, m = {m}")
            assert st["full"] + st["ties"] == P, st
        else:   # (fp32 with one to three parameters: |g|^2 and |dx|^2 are sums of one to three numbers, a stop threshold straddled by the last
                #  bits of a differently ordered sum ends one run early while the other goes on — on BOTH routes —, and the costs of the two
                #  sum orders differ by more than the trajectory tolerance calibrated on wider blocks: end points instead of the tie analysis)
            assert np.abs(x.cpu().numpy() - ref["x"]).max() < (3e-3 if f32 else 1e-8)
            assert (out.stop_reason.cpu().numpy() > 0).all()
        g_ref, H_ref, c_ref, _ = oracle.dense_row_accumulate(A, b, x0)
        g, H, c, nres = ta.accumulate(model, torch.from_numpy(x0).cuda())
        assert np.abs(g.cpu().numpy() - g_ref).max() <= tol * np.abs(g_ref).max()
        assert np.abs(H.cpu().numpy() - H_ref).max() <= tol * np.abs(H_ref).max()
        assert np.allclose(c.cpu().numpy(), c_ref, rtol=tol) and (nres.cpu().numpy() == m).all()
        _, _, c0, _ = ta.accumulate(model, torch.from_numpy(x0).cuda(), want_grad=False)
        assert np.allclose(c0.cpu().numpy(), c_ref, rtol=tol)
        with ta.api.default_context().tuning(narrow_mfma_pass=1):
            x2 = torch.from_numpy(x0.copy()).cuda()
            out2 = ta.Optimize(x2, model, opts, history=True)
            g2, H2, c2, _ = ta.accumulate(model, torch.from_numpy(x0).cuda())
            torch.cuda.synchronize()
        if n >= 4:   # (below, fp32 runs of the old route part from the oracle at straddled stop thresholds the tie analysis does not cover)
            st2 = check_trajectories(gpu_dict(out2, x2), ref, dtype, opts.to_pod(), label=f"sixteen lanes per row n = {n}, m = {m}")
            assert st2["full"] + st2["ties"] == P, st2
        assert float((x - x2).abs().max()) < (2e-3 if f32 else 1e-8)
        assert np.abs((H - H2).cpu().numpy()).max() <= tol * np.abs(H_ref).max()
