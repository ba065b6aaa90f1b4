"""Row models for wide parameter blocks (csrc/row_model.hpp; VERDICT r05 "next" #1): a user's residual WITH ITS OWN JACOBIAN
(the reference's manual Accumulate callback — docs/API.md:37-57; the published benchmark is exactly such callbacks up to n = 50,
benchmarks/dense.cpp:57-66,90-99) beyond 12 parameters, handed over as TEXT, on the matrix-core Gram / LDL^T / state machine of the
compiled-in DenseRow family; M-estimators on those models; and the AD of the same residual through row-per-lane chunked Jets.
Oracle: the CPU restatement's DenseRow functions (the item of every model below is a DenseRow row, or several)."""
import numpy as np
import pytest
import torch

from parity import check_trajectories, gpu_dict

pytestmark = pytest.mark.gpu


def manual_body(n, fast_sincos=True):
    """r = a.x + 0.1 sin(a.x) - b and its Jacobian row (1 + 0.1 cos(a.x)) a, item = [a_0 .. a_{n-1}, b]."""
    sc = "T sn, cs; sincos_t(t, &sn, &cs);" if fast_sincos else "const T sn = sin(t), cs = cos(t);"
    return (f"T t = x[0] * p[0];\nfor (int j = 1; j < {n}; ++j) t += x[j] * p[j];\n{sc}\nr[0] = t + T(0.1) * sn - p[{n}];\n"
            f"if (want_grad) {{\n  const T sc = T(1) + T(0.1) * cs;\n#pragma unroll\n  for (int j = 0; j < {n}; ++j) J[0][j] = sc * p[j];\n}}")


def ad_body(n):
    return f"S t = x[0] * p[0];\n#pragma unroll 2\nfor (int j = 1; j < {n}; ++j) t = t + x[j] * p[j];\nr[0] = t + T(0.1) * sin(t) - p[{n}];"


def _items(A, b):
    return torch.from_numpy(np.ascontiguousarray(np.concatenate([A, b[..., None]], -1))).cuda()


SHAPES = [(50, 300, np.float32), (50, 777, np.float32), (50, 2000, np.float32), (13, 69, np.float64), (16, 100, np.float64), (19, 64, np.float64),
          (33, 129, np.float64), (47, 250, np.float64), (48, 200, np.float32), (50, 130, np.float64), (63, 200, np.float32), (63, 95, np.float64), (19, 27, np.float64)]


@pytest.mark.parametrize("n,m,dtype", SHAPES)
def test_dense_row_with_its_jacobian_supplied_as_text(ta, oracle, n, m, dtype):
    """kind = "accumulate" beyond 12 parameters: (g, H, cost) against the oracle, the cost-only form, the LM trajectory against the
    oracle's and against the compiled-in DenseRow model on the same rows — b in the main block and in the thin tail, row counts that
    end inside a 64-item super-step, one of fewer rows than lanes."""
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    P = 6
    A, b, x0, xs = oracle.synth_dense_row(P, n, m, dtype, seed=700 + n + m)
    fit = ta.JitResidual(manual_body(n), n=n, item_scalars=n + 1, dtype=tdt, kind="accumulate")
    model = fit.bind(_items(A, b))
    x = torch.from_numpy(x0.copy()).cuda()
    g, H, c, nres = ta.accumulate(model, x)
    g_ref, H_ref, c_ref, _ = oracle.dense_row_accumulate(A, b, x0)
    tol = 1e-10 if dtype == np.float64 else 1e-4
    assert np.abs(g.cpu().numpy() - g_ref).max() <= tol * np.abs(g_ref).max()
    assert np.abs(H.cpu().numpy() - H_ref).max() <= tol * np.abs(H_ref).max()
    assert np.allclose(H.cpu().numpy(), np.swapaxes(H.cpu().numpy(), 1, 2))
    assert np.allclose(c.cpu().numpy(), c_ref, rtol=tol) and (nres.cpu().numpy() == m).all()
    c0 = ta.accumulate(model, x, want_grad=False)[2]
    assert np.allclose(c0.cpu().numpy(), c_ref, rtol=tol)
    opts = ta.Options.benchmark()
    ref = oracle.dense_row_lm(A, b, x0, opts.to_pod(), history=True)
    out = ta.Optimize(x, model, opts, history=True)
    torch.cuda.synchronize()
    st = check_trajectories(gpu_dict(out, x), ref, dtype, opts.to_pod(), label=f"manual Jacobian as text, n = {n}")
    assert st["full"] + st["ties"] == P
    # the compiled-in family on the same rows: the same decisions, the same point to rounding
    xb = torch.from_numpy(x0.copy()).cuda()
    outb = ta.Optimize(xb, ta.DenseRow.from_arrays(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda()), opts, history=True)
    torch.cuda.synchronize()
    assert float((x - xb).abs().max()) <= (1e-9 if dtype == np.float64 else 2e-3)
    if dtype == np.float64:
        assert torch.equal(out.num_iters, outb.num_iters) and torch.equal(out.stop_reason, outb.stop_reason)
    # run to run, and with the memo of the last accepted linearisation switched off: the same bits
    x2 = torch.from_numpy(x0.copy()).cuda()
    with ta.api.default_context().tuning(memo_off=1):
        out2 = ta.Optimize(x2, model, opts, history=True)
    torch.cuda.synchronize()
    assert torch.equal(x2, x) and torch.equal(out2.errs, out.errs) and torch.equal(out2.num_iters, out.num_iters)


@pytest.mark.parametrize("n,m,dtype", [(50, 300, np.float32), (24, 120, np.float64), (13, 90, np.float64), (63, 200, np.float32), (50, 1000, np.float64)])
def test_row_per_lane_ad_equals_the_manual_jacobian(ta, oracle, n, m, dtype):
    """The same residual as r(x) only: chunked Jets, a row per lane (AdRowFunctor).  Its (g, H, cost) equal the manual model's to
    rounding, and both the oracle's."""
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    P = 5
    A, b, x0, xs = oracle.synth_dense_row(P, n, m, dtype, seed=40 + n)
    data = _items(A, b)
    man = ta.JitResidual(manual_body(n, fast_sincos=False), n=n, item_scalars=n + 1, dtype=tdt, kind="accumulate").bind(data)
    ad = ta.JitResidual(ad_body(n), n=n, item_scalars=n + 1, dtype=tdt).bind(data)
    x = torch.from_numpy(x0.copy()).cuda()
    gm, Hm, cm, _ = ta.accumulate(man, x)
    ga, Ha, ca, _ = ta.accumulate(ad, x)
    g_ref, H_ref, c_ref, _ = oracle.dense_row_accumulate(A, b, x0)
    tol = 1e-10 if dtype == np.float64 else 1e-4
    for g, H, c in ((gm, Hm, cm), (ga, Ha, ca)):
        assert np.abs(g.cpu().numpy() - g_ref).max() <= tol * np.abs(g_ref).max()
        assert np.abs(H.cpu().numpy() - H_ref).max() <= tol * np.abs(H_ref).max()
        assert np.allclose(c.cpu().numpy(), c_ref, rtol=tol)
    assert float((gm - ga).abs().max()) <= 10 * tol * float(gm.abs().max())


KINDS = ["truncated", "huber", "tukey", "arctan", "cauchy", "geman_mcclure", "blake_zisserman"]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("n,m,dtype,form", [(50, 402, np.float32, "manual"), (16, 77, np.float64, "manual"), (33, 150, np.float64, "ad"), (50, 130, np.float64, "manual")])
def test_m_estimators_on_row_models(ta, oracle, kind, n, m, dtype, form):
    """toa_set_loss on a model beyond 12 parameters (refused until round 5): every loss kind, accumulate and cost-only, against the
    oracle's DenseRow with the same loss; then the LM trajectory under the loss."""
    from test_gpu_robust_dense import _with_outliers
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    P = 5
    A, b, x0, xs, _ = _with_outliers(oracle, P, n, m, dtype)
    x = (xs + 0.02 * np.random.default_rng(1).uniform(-1, 1, xs.shape)).astype(dtype)
    th = 0.05 if dtype == np.float64 else 0.3
    g_ref, H_ref, c_ref, nres_ref, inl_ref = oracle.dense_row_accumulate(A, b, x, loss=kind, th2=th * th)
    body = manual_body(n, fast_sincos=False) if form == "manual" else ad_body(n)
    fit = ta.JitResidual(body, n=n, item_scalars=n + 1, dtype=tdt, kind="accumulate" if form == "manual" else "residual")
    model = fit.bind(_items(A, b)).with_loss(kind, th)
    g, H, c, nres = ta.accumulate(model, torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    tol = 1e-10 if dtype == np.float64 else 2e-4
    assert np.abs(g.cpu().numpy() - g_ref).max() <= tol * np.abs(g_ref).max()
    assert np.abs(H.cpu().numpy() - H_ref).max() <= tol * np.abs(H_ref).max()
    assert np.allclose(c.cpu().numpy(), c_ref, rtol=tol) and (nres.cpu().numpy() == m).all()
    c0 = ta.accumulate(model, torch.from_numpy(x).cuda(), want_grad=False)[2]
    assert np.allclose(c0.cpu().numpy(), c_ref, rtol=tol)
    if dtype == np.float64 and kind in ("huber", "cauchy", "tukey"):
        # (the fixture of test_dense_row_robust_lm_recovers_planted_solution: a start inside the basin of the redescending losses)
        x0 = (xs + 0.05 * np.random.default_rng(2).uniform(-1, 1, xs.shape)).astype(dtype)
        thl = 0.5 if kind == "tukey" else 0.02
        opts = ta.Options()
        ref = oracle.dense_row_lm(A, b, x0, opts.to_pod(), history=True, loss=kind, th2=thl * thl)
        xg = torch.from_numpy(x0.copy()).cuda()
        out = ta.Optimize(xg, model.with_loss(kind, thl), opts, history=True)
        torch.cuda.synchronize()
        refd = dict(errs=ref["errs"], succ=ref["succ"], iters=ref["iters"], stop=ref["stop"], x=ref["x"], cost=ref["cost"],
                    fails=ref["fails"], deltas2=ref["deltas2"])
        st = check_trajectories(gpu_dict(out, xg), refd, dtype, opts.to_pod(), label=f"row model + {kind}")
        assert st["full"] + st["ties"] == P
        assert np.abs(out.final_inlier_ratio.cpu().numpy() - ref["inlier_ratio"]).max() <= 2.0 / m
        assert np.abs(xg.cpu().numpy() - ref["x"]).max() < 1e-7


def _two_rows_manual(n):
    """An item = (a [n], c [n], b0, b1): two DenseRow rows with their Jacobian rows."""
    return (f"T t = 0, u = 0;\nfor (int j = 0; j < {n}; ++j) {{ t += x[j] * p[j]; u += x[j] * p[{n} + j]; }}\n"
            f"r[0] = t + T(0.1) * sin(t) - p[{2 * n}];\nr[1] = u + T(0.1) * sin(u) - p[{2 * n + 1}];\n"
            f"if (want_grad) {{\n  const T s0 = T(1) + T(0.1) * cos(t), s1 = T(1) + T(0.1) * cos(u);\n#pragma unroll\n"
            f"  for (int j = 0; j < {n}; ++j) {{ J[0][j] = s0 * p[j]; J[1][j] = s1 * p[{n} + j]; }}\n}}")


@pytest.mark.parametrize("n,items,dtype", [(20, 160, np.float64), (50, 203, np.float32), (14, 37, np.float64)])
def test_vector_residual_items_with_manual_jacobians_every_form(ta, oracle, n, items, dtype):
    """Two residuals per item with their Jacobian rows: (g, H, cost), the whole solve, the row-split form (chunks on item
    boundaries) and the stepping form."""
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    P, m = 4, 2 * items
    A, b, x0, xs = oracle.synth_dense_row(P, n, m, dtype, seed=300 + n)
    fit = ta.JitResidual(_two_rows_manual(n), n=n, item_scalars=2 * n + 2, residuals_per_item=2, dtype=tdt, kind="accumulate")
    item = np.concatenate([A[:, 0::2], A[:, 1::2], b[:, 0::2, None], b[:, 1::2, None]], -1)       # rows 2 i, 2 i + 1 -> item i
    model = fit.bind(torch.from_numpy(np.ascontiguousarray(item)).cuda())
    x = torch.from_numpy(x0.copy()).cuda()
    g, H, c, nres = ta.accumulate(model, x)
    g_ref, H_ref, c_ref, _ = oracle.dense_row_accumulate(A, b, x0)
    tol = 1e-10 if dtype == np.float64 else 1e-4
    assert np.abs(g.cpu().numpy() - g_ref).max() <= tol * np.abs(g_ref).max()
    assert np.abs(H.cpu().numpy() - H_ref).max() <= tol * np.abs(H_ref).max()
    assert np.allclose(c.cpu().numpy(), c_ref, rtol=tol) and (nres.cpu().numpy() == m).all()
    opts = ta.Options.benchmark()
    ref = oracle.dense_row_lm(A, b, x0, opts.to_pod(), history=True)
    out = ta.Optimize(x, model, opts, history=True)
    torch.cuda.synchronize()
    st = check_trajectories(gpu_dict(out, x), ref, dtype, opts.to_pod(), label=f"two manual rows per item, n = {n}")
    assert st["full"] + st["ties"] == P
    for splits in (1, 3):
        x2 = torch.from_numpy(x0.copy()).cuda()
        o2 = ta.Optimize(x2, model, opts, history=True, splits=splits)
        torch.cuda.synchronize()
        st2 = check_trajectories(gpu_dict(o2, x2), ref, dtype, opts.to_pod(), label=f"two manual rows per item, splits = {splits}")
        assert st2["full"] + st2["ties"] == P
    x3 = torch.from_numpy(x0.copy()).cuda()
    o3 = ta.Optimize(x3, model, opts, history=True, splits=1)
    x4 = torch.from_numpy(x0.copy()).cuda()
    opt = ta.Optimizer(x4, model, opts, history=True)
    for _ in range(opts.max_iters + 3):
        if opt.Step() == 0:
            break
    torch.cuda.synchronize()
    assert torch.equal(x4, x3) and torch.equal(opt.out.num_iters, o3.num_iters) and torch.equal(opt.out.errs, o3.errs)


def test_headline_shape_as_text_follows_the_compiled_in_model(ta, oracle):
    """BASELINE C4's shape (n = 50, m = 2000, fp32) with the residual and its Jacobian supplied as text: the same iteration counts
    as the compiled-in family within the fp32 tie tolerance of check_trajectories, the same point to 2e-3."""
    P, n, m = 64, 50, 2000
    A, b, x0, xs = oracle.synth_dense_row(P, n, m, np.float32)
    opts = ta.Options.benchmark()
    ref = oracle.dense_row_lm(A, b, x0, opts.to_pod(), history=True)
    fit = ta.JitResidual(manual_body(n), n=n, item_scalars=n + 1, dtype=torch.float32, kind="accumulate")
    x = torch.from_numpy(x0.copy()).cuda()
    out = ta.Optimize(x, fit.bind(_items(A, b)), opts, history=True)
    torch.cuda.synchronize()
    st = check_trajectories(gpu_dict(out, x), ref, np.float32, opts.to_pod(), label="C4 as text")
    assert st["full"] + st["ties"] == P
    assert np.abs(x.cpu().numpy() - ref["x"]).max() < 2e-3


@pytest.mark.parametrize("n", [16, 32, 48])
@pytest.mark.parametrize("kind", ["accumulate", "residual"])
def test_row_models_in_the_row_split_and_stepping_forms_at_block_boundaries(ta, oracle, n, kind):
    """a run-time row model with P * 4 <= #CUs and >= 512 rows is routed to the row-split form by toa_jit_lm_run
    itself; at n = 16, 32, 48 the step kernels' register LDL^T is exactly as wide as the parameter block (npad = n), narrower than
    the compiled-in families' width.  One huge problem per shape: the automatic route, explicit chunk counts and the stepping form
    against the oracle's trajectory, and the one-wavefront form (toa_tuning::wide_no_autosplit) on the same data."""
    P, m = 1, 1600
    A, b, x0, xs = oracle.synth_dense_row(P, n, m, np.float64, seed=910 + n)
    body = manual_body(n, fast_sincos=False) if kind == "accumulate" else ad_body(n)
    model = ta.JitResidual(body, n=n, item_scalars=n + 1, dtype=torch.float64, kind=kind).bind(_items(A, b))
    opts = ta.Options.benchmark()
    ref = oracle.dense_row_lm(A, b, x0, opts.to_pod(), history=True)
    for splits in (None, 1, 5, 64):
        x = torch.from_numpy(x0.copy()).cuda()
        out = ta.Optimize(x, model, opts, history=True) if splits is None else ta.Optimize(x, model, opts, history=True, splits=splits)
        torch.cuda.synchronize()
        st = check_trajectories(gpu_dict(out, x), ref, np.float64, opts.to_pod(), label=f"row model n = {n}, splits = {splits}")
        assert st["full"] + st["ties"] == P
    with ta.api.default_context().tuning(wide_no_autosplit=1):
        x1 = torch.from_numpy(x0.copy()).cuda()
        o1 = ta.Optimize(x1, model, opts, history=True)
        torch.cuda.synchronize()
    st = check_trajectories(gpu_dict(o1, x1), ref, np.float64, opts.to_pod(), label=f"row model n = {n}, one wavefront")
    assert st["full"] + st["ties"] == P
    xs_ = torch.from_numpy(x0.copy()).cuda()
    opt = ta.Optimizer(xs_, model, opts, history=True)
    for _ in range(opts.max_iters + 3):
        if opt.Step() == 0:
            break
    torch.cuda.synchronize()
    st = check_trajectories(gpu_dict(opt.out, xs_), ref, np.float64, opts.to_pod(), label=f"row model n = {n}, stepping")
    assert st["full"] + st["ties"] == P


# ---------------------------------------------------------------------------------------------------------------------------
# A USER manifold beyond 12 tangent dimensions (VERDICT r05 "missing" #2: "a user's own Jacobian (or manifold, or loss) at
# 13 <= n <= 63"): K planar rotations stored as (cos, sin) pairs — x has 2 K scalars, the tangent K angles, x (+) d rotates pair k
# by d[k] (traits::params_trait<T>::PlusEq, traits.h:103-359, as text).  One residual per item, dense in every rotation.
# ---------------------------------------------------------------------------------------------------------------------------
def _rot_plus(K):
    return (f"for (int k = 0; k < {K}; ++k) {{\n  const S c = cos(d[k]), s = sin(d[k]);\n"
            "  xp[2 * k] = x[2 * k] * c - x[2 * k + 1] * s;\n  xp[2 * k + 1] = x[2 * k + 1] * c + x[2 * k] * s;\n}")


def _rot_residual(K):      # over the stored scalars (cos, sin): linear in them
    return (f"S t = x[0] * p[0] - x[1] * p[1];\nfor (int k = 1; k < {K}; ++k) t = t + x[2 * k] * p[2 * k] - x[2 * k + 1] * p[2 * k + 1];\n"
            f"r[0] = t - p[{2 * K}];")


def _rot_manual(K):        # the same with its own Jacobian over the TANGENT: d r / d d_k at d = 0
    return (f"T t = T(0);\nfor (int k = 0; k < {K}; ++k) t += x[2 * k] * p[2 * k] - x[2 * k + 1] * p[2 * k + 1];\nr[0] = t - p[{2 * K}];\n"
            f"if (want_grad) for (int k = 0; k < {K}; ++k) J[0][k] = -x[2 * k + 1] * p[2 * k] - x[2 * k] * p[2 * k + 1];")


def _angle_residual(K):    # the Euclidean twin: the angles themselves are the parameters
    return (f"S t = cos(x[0]) * p[0] - sin(x[0]) * p[1];\nfor (int k = 1; k < {K}; ++k) t = t + cos(x[k]) * p[2 * k] - sin(x[k]) * p[2 * k + 1];\n"
            f"r[0] = t - p[{2 * K}];")


@pytest.mark.parametrize("K,tdt", [(16, torch.float64), (20, torch.float64), (27, torch.float64), (16, torch.float32), (20, torch.float32)])
def test_user_manifold_beyond_twelve_tangent_dimensions(ta, oracle, K, tdt):
    """x (+) d is t + d exactly, so the solve on the manifold — derivative by chunked Jets THROUGH x (+) d (optimize_autodiff.h:48-77),
    or the user's own Jacobian over the tangent — takes the steps of the Euclidean solve over the angles: (g, H, cost) against numpy,
    cost / accept histories, iteration counts and StopReasons against the twin (fp64; fp32: end points), every execution form, and x
    stays on the manifold.  K = 20: two chunks of ten partials; K = 27: three of nine."""
    rng = np.random.default_rng(40 + K)
    P, items = 5, 600
    t_true = rng.uniform(-1.0, 1.0, (P, K))
    ab = rng.uniform(-1, 1, (P, items, 2 * K))
    a, b = ab[..., 0::2], ab[..., 1::2]
    y = (np.cos(t_true)[:, None, :] * a - np.sin(t_true)[:, None, :] * b).sum(-1) + 1e-3 * rng.uniform(-1, 1, (P, items))
    data = torch.from_numpy(np.concatenate([ab, y[..., None]], -1)).to(tdt).cuda()
    t0 = t_true + rng.uniform(-0.3, 0.3, (P, K))
    x0m = np.stack([np.cos(t0), np.sin(t0)], -1).reshape(P, 2 * K)
    kw = dict(n=K, item_scalars=2 * K + 1, dtype=tdt)
    angle = ta.JitResidual(_angle_residual(K), **kw).bind(data)
    man_ad = ta.JitResidual(_rot_residual(K), manifold="user", plus_body=_rot_plus(K), x_scalars=2 * K, **kw)
    man_j = ta.JitResidual(_rot_manual(K), manifold="user", plus_body=_rot_plus(K), x_scalars=2 * K, kind="accumulate", **kw)
    assert man_ad.xdim == 2 * K and man_j.xdim == 2 * K
    # the seam against numpy: J_k = -sin(t_k) a_k - cos(t_k) b_k
    Jn = -np.sin(t0)[:, None, :] * a - np.cos(t0)[:, None, :] * b
    rn = (np.cos(t0)[:, None, :] * a - np.sin(t0)[:, None, :] * b).sum(-1) - y
    Hn, gn, cn = np.einsum("pik,pil->pkl", Jn, Jn), np.einsum("pik,pi->pk", Jn, rn), (rn * rn).sum(-1)
    tol = 1e-10 if tdt == torch.float64 else 2e-4
    for jit in (man_ad, man_j):
        g, H, c, nres = ta.accumulate(jit.bind(data), torch.from_numpy(x0m).to(tdt).cuda())
        torch.cuda.synchronize()
        assert np.abs(H.double().cpu().numpy() - Hn).max() <= tol * np.abs(Hn).max()
        assert np.abs(g.double().cpu().numpy() - gn).max() <= tol * np.abs(Hn).max()
        assert np.allclose(c.cpu().numpy(), cn, rtol=tol * 10) and int(nres[0]) == items
        _, _, c2, _ = ta.accumulate(jit.bind(data), torch.from_numpy(x0m).to(tdt).cuda(), want_grad=False)
        assert np.allclose(c2.cpu().numpy(), cn, rtol=tol * 10)
    opts = ta.Options()
    xa = torch.from_numpy(t0.copy()).to(tdt).cuda()
    oa = ta.Optimize(xa, angle, opts, history=True)
    torch.cuda.synchronize()
    assert bool((oa.stop_reason > 0).all())
    for name, jit in (("AD", man_ad), ("own Jacobian", man_j)):
        for form in ("launch", "split", "step"):
            xm = torch.from_numpy(x0m.copy()).to(tdt).cuda()
            if form == "launch":
                om = ta.Optimize(xm, jit.bind(data), opts, history=True)
            elif form == "split":
                om = ta.Optimize(xm, jit.bind(data), opts, history=True, splits=3)
            else:
                om = ta.Optimizer(xm, jit.bind(data), opts, history=True)()
            torch.cuda.synchronize()
            label = f"K = {K}, {name}, {form}"
            assert bool((om.stop_reason > 0).all()), label
            if tdt == torch.float64 and form != "split":   # (the split form sums the rows in chunks: same steps to round-off, not to the bit)
                assert torch.equal(om.num_iters, oa.num_iters) and torch.equal(om.stop_reason, oa.stop_reason), label
                k = int(oa.num_iters.min())
                assert np.allclose(om.errs.cpu().numpy()[:, :k], oa.errs.cpu().numpy()[:, :k], rtol=1e-7, atol=1e-14), label
            xm64 = xm.double().cpu().numpy().reshape(P, K, 2)
            ang = np.arctan2(xm64[..., 1], xm64[..., 0])
            assert np.abs(ang - xa.double().cpu().numpy()).max() < (1e-7 if tdt == torch.float64 else 2e-3), label
            assert np.abs(ang - t_true).max() < 1e-2, label
            assert np.abs((xm64 ** 2).sum(-1) - 1).max() < (1e-12 if tdt == torch.float64 else 1e-5), label   # stays on the manifold
    with pytest.raises(Exception):
        ta.JitResidual(_rot_residual(K), manifold="user", plus_body=_rot_plus(K), x_scalars=65, **kw)   # one stored scalar per lane at most
