"""Run-time user functors (csrc/jit.hip, `toa_model_compile`): the device-side form of tinyopt's "pass any callable"
(optimize.h:16-33, optimizer.h:145-160, docs/API.md:21-35).  The residual arrives as C++ source text at RUN time, hiprtc builds
the fused LM kernel for it and the code object is loaded — libtinyopt_amd.so is not rebuilt.

Done-criterion of the round-2 verdict: the circle fit of tests/circle.cpp:32-68, supplied as source, reproduces
test_circle_fit_reference_known_answer and the oracle's trajectory."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CIRCLE = "const S dx = p[0] - x[0];\nconst S dy = p[1] - x[1];\nr[0] = dx * dx + dy * dy - x[2] * x[2];"


def _circle_obs(P, n, dtype, seed=0):
    """tests/circle.cpp:20-30: n points on a circle of radius 2 centred at (2, 7) + 1e-5 noise."""
    rng = np.random.default_rng(seed)
    ang = np.linspace(0, 2 * np.pi, n)[None, :] + rng.uniform(0, 1, (P, 1))
    obs = np.stack([2 + 2 * np.cos(ang), 7 + 2 * np.sin(ang)], -1) + 1e-5 * rng.uniform(-1, 1, (P, n, 2))
    return obs.astype(dtype)


@pytest.mark.parametrize("dtype,tdt", [(np.float64, torch.float64), (np.float32, torch.float32)])
def test_circle_fit_supplied_as_source_at_run_time(ta, oracle, dtype, tdt):
    """tests/circle.cpp:32-68: x0 = (0, 0, 1), lm.damping_init = 10 -> (2, 7, 2) +- 1e-5, Succeeded — with the residual
    handed over as text; StopReason / iterations / cost / x equal to the oracle's and BIT-equal to the built-in CircleFit."""
    P, npts = 7, 10
    obs = _circle_obs(P, npts, dtype)
    x0 = np.tile(np.array([0, 0, 1], dtype), (P, 1))
    o = ta.Options()
    o.lm.damping_init = 1e1
    ref = oracle.circle_fit_lm(obs, x0, o.to_pod())
    fit = ta.JitResidual(CIRCLE, n=3, item_scalars=2, dtype=tdt)
    assert fit.compile_log == "" or "error" not in fit.compile_log
    x = torch.from_numpy(x0.copy()).cuda()
    out = ta.Optimize(x, fit.bind(torch.from_numpy(obs).cuda()), o, history=True)
    torch.cuda.synchronize()
    xg, stop = x.cpu().numpy(), out.stop_reason.cpu().numpy()
    assert (stop >= 0).all()
    tol = 1e-5 if dtype == np.float64 else 2e-4
    assert np.abs(xg[:, 0] - 2).max() < tol and np.abs(xg[:, 1] - 7).max() < tol and np.abs(np.abs(xg[:, 2]) - 2).max() < tol
    if dtype == np.float64:
        assert np.abs(xg - ref["x"]).max() < 1e-8
        assert np.array_equal(stop, ref["stop"]) and np.array_equal(out.num_iters.cpu().numpy(), ref["iters"])
        assert np.allclose(out.final_cost.cpu().numpy(), ref["cost"], rtol=1e-6, atol=1e-18)
    xb = torch.from_numpy(x0.copy()).cuda()
    outb = ta.Optimize(xb, ta.CircleFit(torch.from_numpy(obs).cuda()), o, history=True)
    torch.cuda.synchronize()
    assert torch.equal(x, xb) and torch.equal(out.errs, outb.errs) and torch.equal(out.num_iters, outb.num_iters)


def test_functor_with_header_two_residuals_and_transcendentals(ta):
    """A residual the library has no built-in for: per problem a header (w, phi), per item (t, y0, y1); two residuals per
    item through exp / sin / cos / atan2 / pow of the parameters.  (g, H, cost) of the device AD against torch.autograd of
    the same formula in float64, then a solve that recovers the planted parameters."""
    body = """
    const S a = x[0], k = x[1], f = x[2], c = x[3];
    const S e = exp(-k * p[0]);
    r[0] = a * e * sin(f * p[0] + h[1]) + c - p[1];
    r[1] = h[0] * (atan2(a * e * cos(f * p[0] + h[1]), S(1.0) + pow(c, 2)) - p[2]);
    """
    P, items = 11, 160
    rng = np.random.default_rng(5)
    xs = np.stack([rng.uniform(1.5, 2.5, P), rng.uniform(0.2, 0.6, P), rng.uniform(2.0, 3.0, P), rng.uniform(-0.5, 0.5, P)], axis=1)
    hdr = np.stack([rng.uniform(0.5, 1.5, P), rng.uniform(-1, 1, P)], axis=1)
    t = np.tile(np.linspace(0.0, 3.0, items), (P, 1))

    def model_np(xv):
        a, k, f, c = (xv[:, i:i + 1] for i in range(4))
        e = np.exp(-k * t)
        y0 = a * e * np.sin(f * t + hdr[:, 1:2]) + c
        y1 = np.arctan2(a * e * np.cos(f * t + hdr[:, 1:2]), 1.0 + c ** 2)
        return y0, y1
    y0, y1 = model_np(xs)
    y0 = y0 + 1e-3 * rng.uniform(-1, 1, y0.shape)
    y1 = y1 + 1e-3 * rng.uniform(-1, 1, y1.shape)
    data = np.stack([t, y0, y1], axis=2)
    res = ta.JitResidual(body, n=4, item_scalars=3, residuals_per_item=2, header_scalars=2, dtype=torch.float64)
    model = res.bind(torch.from_numpy(data).cuda(), torch.from_numpy(hdr).cuda())
    x0 = xs + 0.05 * rng.uniform(-1, 1, xs.shape)
    # ---- Accumulate against autograd
    xt = torch.tensor(x0, dtype=torch.float64, requires_grad=True)
    tt, h0, h1 = torch.from_numpy(t), torch.from_numpy(hdr[:, 0:1]), torch.from_numpy(hdr[:, 1:2])

    def residuals(xv):
        a, k, f, c = (xv[:, i:i + 1] for i in range(4))
        e = torch.exp(-k * tt)
        r0 = a * e * torch.sin(f * tt + h1) + c - torch.from_numpy(y0)
        r1 = h0 * (torch.atan2(a * e * torch.cos(f * tt + h1), 1.0 + c ** 2) - torch.from_numpy(y1))
        return torch.stack([r0, r1], dim=2).reshape(P, -1)
    rr = residuals(xt)
    J = torch.stack([torch.autograd.grad(rr[:, i].sum(), xt, retain_graph=True)[0] for i in range(rr.shape[1])], dim=1)  # [P, m, n]
    g_ref = torch.einsum("pmn,pm->pn", J, rr.detach()).numpy()
    H_ref = torch.einsum("pmn,pmk->pnk", J, J).numpy()
    c_ref = (rr.detach() ** 2).sum(dim=1).numpy()
    g, H, c, nres = ta.accumulate(model, torch.from_numpy(x0).cuda())
    torch.cuda.synchronize()
    assert (nres.cpu().numpy() == 2 * items).all()
    assert np.allclose(g.cpu().numpy(), g_ref, rtol=1e-10, atol=1e-10 * np.abs(g_ref).max())
    assert np.allclose(H.cpu().numpy(), H_ref, rtol=1e-10, atol=1e-10 * np.abs(H_ref).max())
    assert np.allclose(c.cpu().numpy(), c_ref, rtol=1e-12)
    c0 = ta.accumulate(model, torch.from_numpy(x0).cuda(), want_grad=False)[2]
    assert np.allclose(c0.cpu().numpy(), c_ref, rtol=1e-12)
    # ---- and the solve
    x = torch.from_numpy(x0.copy()).cuda()
    out = ta.Optimize(x, model, ta.Options())
    torch.cuda.synchronize()
    assert bool((out.stop_reason >= 0).all())
    assert np.abs(x.cpu().numpy() - xs).max() < 5e-3
    # a Huber loss on each item's squared norm (toa_set_loss applies to run-time models like to the built-in Jet families)
    xl = torch.from_numpy(x0.copy()).cuda()
    outl = ta.Optimize(xl, model.with_loss("huber", 0.05), ta.Options())
    torch.cuda.synchronize()
    assert bool((outl.stop_reason >= 0).all()) and np.abs(xl.cpu().numpy() - xs).max() < 5e-3
    assert float(outl.final_inlier_ratio.min()) > 0.9


def test_compile_errors_and_limits(ta):
    with pytest.raises(ta.ToaError) as e:
        ta.JitResidual("r[0] = undefined_symbol(x[0]);", n=1, item_scalars=1)
    assert "undefined_symbol" in str(e.value)                       # the compiler's diagnostic reaches the caller
    with pytest.raises(ta.ToaError):
        ta.JitResidual("r[0] = x[0];", n=64, item_scalars=1)           # one wavefront per problem: 63 parameters at most
    with pytest.raises(ta.ToaError):
        ta.JitResidual("r[0] = x[0];", n=7, item_scalars=1, manifold="se3")   # an SE3 pose has a 6-dimensional tangent
    res = ta.JitResidual("r[0] = x[0] * x[0] - p[0];", n=1, item_scalars=1)     # sqrt: the smallest possible model
    data = torch.full((3, 1, 1), 2.0, dtype=torch.float64, device="cuda")
    x = torch.tensor([[1.0], [-0.3], [3.2]], dtype=torch.float64, device="cuda")
    o = ta.Options()
    o.max_iters, o.max_consec_failures = 20, 0                                  # tests/sqrt2.cpp:106-112
    out = ta.Optimize(x, res.bind(data), o)
    torch.cuda.synchronize()
    assert bool((out.stop_reason >= 1).all()) and float((x.abs() - 2 ** 0.5).abs().max()) < 1e-5       # tests/sqrt2.cpp:55


# ---------------------------------------------------------------------------------------------------------------------------
# Round 4 (VERDICT r03 "missing #2"): wide blocks, a manifold, manual Accumulate bodies, the on-disk cache
# ---------------------------------------------------------------------------------------------------------------------------
SE3_PRIOR = """
// tests/sophus.cpp:26-44: residual(x) = log(prior_inv * x); h = prior_inv (R row-major, t), x = the pose (R row-major, t)
S RA[9], tA[3];
for (int i = 0; i < 3; ++i) {
  for (int j = 0; j < 3; ++j) RA[3 * i + j] = x[j] * h[3 * i] + x[3 + j] * h[3 * i + 1] + x[6 + j] * h[3 * i + 2];
  tA[i] = x[9] * h[3 * i] + x[10] * h[3 * i + 1] + x[11] * h[3 * i + 2] + h[9 + i];
}
se3_log<S, T>(RA, tA, r);
"""


def _dense_row_body(n):
    return f"S t = x[0] * p[0];\n#pragma unroll 2\nfor (int j = 1; j < {n}; ++j) t = t + x[j] * p[j];\nr[0] = t + T(0.1) * sin(t) - p[{n}];"


@pytest.mark.parametrize("n,m,dtype,tdt", [(50, 300, np.float32, torch.float32), (24, 120, np.float64, torch.float64), (13, 90, np.float64, torch.float64),
                                           (63, 200, np.float32, torch.float32)])
def test_wide_parameter_blocks_at_run_time(ta, oracle, n, m, dtype, tdt):
    """num_params up to 63: the DenseRow residual written as r(x) only and handed over as TEXT — (g, H, cost) equal the analytic
    MFMA path's to rounding and the LM trajectory equals the oracle's (the n = 50 case is what `DenseRowAD` compiles in)."""
    from parity import check_trajectories, gpu_dict
    P = 6
    A, b, x0, xs = oracle.synth_dense_row(P, n, m, dtype, seed=90 + n)
    fit = ta.JitResidual(_dense_row_body(n), n=n, item_scalars=n + 1, dtype=tdt)
    data = torch.from_numpy(np.concatenate([A, b[..., None]], -1)).cuda()
    model = fit.bind(data)
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
    st = check_trajectories(gpu_dict(out, x), ref, dtype, opts.to_pod(), label=f"JIT n = {n}")
    assert st["full"] + st["ties"] == P
    # (round 6: a loss on a model beyond 12 parameters is honoured — tests/test_gpu_row_models.py — no longer refused)


@pytest.mark.parametrize("dtype,tdt", [(np.float64, torch.float64), (np.float32, torch.float32)])
def test_se3_pose_prior_supplied_as_source(ta, oracle, dtype, tdt):
    """tests/sophus.cpp:26-44 — `Optimize(pose, [&](const auto& x) { return (prior_inv * x).log(); })` — with the lambda as a
    string and the manifold as a tag: same verdict as the reference test (||log(pose * prior_inv)|| < 1e-5, Succeeded) and the
    same trajectory as the compiled-in SE3Prior model."""
    P = 16
    rng = np.random.default_rng(5)
    ident64 = np.tile(np.concatenate([np.eye(3).ravel(), np.zeros(3)]), (P, 1))
    prior_inv = oracle.se3_plus(ident64, 0.6 * rng.uniform(-1, 1, (P, 6))).astype(dtype)   # prior_inv = exp(random), as the reference test
    ident = ident64.astype(dtype)
    fit = ta.JitResidual(SE3_PRIOR, n=6, item_scalars=0, residuals_per_item=6, header_scalars=12, dtype=tdt, manifold="se3")
    hdr = torch.from_numpy(prior_inv).cuda()
    model = fit.bind(None, hdr)
    o = ta.Options()
    x = torch.from_numpy(ident.copy()).cuda()
    out = ta.Optimize(x, model, o, history=True)
    xb = torch.from_numpy(ident.copy()).cuda()
    outb = ta.Optimize(xb, ta.SE3Prior(hdr), o, history=True)
    torch.cuda.synchronize()
    stop = out.stop_reason.cpu().numpy()
    assert ((stop >= 1) & (stop < 5)).all()                      # Succeeded && Converged
    resid = oracle.se3_log(oracle.se3_compose(x.cpu().numpy().astype(np.float64), prior_inv.astype(np.float64)))
    assert np.linalg.norm(resid, axis=1).max() < (1e-5 if dtype == np.float64 else 2e-3)
    assert torch.equal(out.stop_reason, outb.stop_reason) and torch.equal(out.num_iters, outb.num_iters)
    k = int(out.num_iters.min())
    # (fp32: the last costs are round-off of a residual that is zero at the solution — compared on the scale of the first cost)
    assert np.allclose(out.errs.cpu().numpy()[:, :k], outb.errs.cpu().numpy()[:, :k], rtol=1e-9 if dtype == np.float64 else 2e-3,
                       atol=1e-12 if dtype == np.float64 else 1e-6)
    R = x.cpu().numpy()[:, :9].reshape(P, 3, 3)
    assert np.abs(np.einsum("pij,pkj->pik", R, R) - np.eye(3)).max() < (1e-12 if dtype == np.float64 else 1e-5)   # stays on the manifold


ROSENBROCK_ACC = """
// tests/optimize_easy.cpp:35-79: a manual Accumulate callback written as least squares r = (1 - x0, 10 (x1 - x0^2)) with ITS OWN Jacobian
r[0] = T(1) - x[0];
r[1] = T(10) * (x[1] - x[0] * x[0]);
if (want_grad) {
  J[0][0] = T(-1);             J[0][1] = T(0);
  J[1][0] = T(-20) * x[0];     J[1][1] = T(10);
}
"""
ROSENBROCK_AD = "r[0] = T(1) - x[0];\nr[1] = T(10) * (x[1] - x[0] * x[0]);"


def test_manual_accumulate_body_equals_the_ad_of_the_same_residual(ta):
    """kind="accumulate" (docs/API.md:37-57): the user's Jacobian rows instead of Jets.  The same residual through device AD must
    give the same (g, H, cost) and the same solve; from the reference's start (-1.2, 1) both reach (1, 1) +- 1e-5."""
    starts = torch.tensor([[-1.2, 1.0], [-0.9, 1.3], [0.3, -0.4], [2.0, 2.0]], dtype=torch.float64, device="cuda")
    P = starts.shape[0]
    man = ta.JitResidual(ROSENBROCK_ACC, n=2, item_scalars=0, residuals_per_item=2, header_scalars=1, kind="accumulate")
    ad = ta.JitResidual(ROSENBROCK_AD, n=2, item_scalars=0, residuals_per_item=2, header_scalars=1)
    hdr = torch.zeros(P, 1, dtype=torch.float64, device="cuda")
    gm, Hm, cm, nm = ta.accumulate(man.bind(None, hdr), starts)
    ga, Ha, ca, na = ta.accumulate(ad.bind(None, hdr), starts)
    assert torch.allclose(gm, ga, rtol=1e-13, atol=0) and torch.allclose(Hm, Ha, rtol=1e-13, atol=0) and torch.equal(cm, ca)
    assert (nm.cpu().numpy() == 2).all()
    _, _, c0, _ = ta.accumulate(man.bind(None, hdr), starts, want_grad=False)       # the cost-only form never touches J
    assert torch.equal(c0, cm)
    o = ta.Options()
    o.max_iters = 200
    o.min_rerr_dec = 0
    o.max_consec_failures = 20
    xm, xa = starts.clone(), starts.clone()
    om = ta.Optimize(xm, man.bind(None, hdr), o, history=True)
    oa = ta.Optimize(xa, ad.bind(None, hdr), o, history=True)
    torch.cuda.synchronize()
    assert torch.equal(om.num_iters, oa.num_iters) and torch.equal(om.stop_reason, oa.stop_reason)
    assert (om.stop_reason.cpu().numpy() >= 1).all()
    assert np.abs(xm.cpu().numpy() - 1.0).max() < 1e-5 and np.abs(xm.cpu().numpy() - xa.cpu().numpy()).max() < 1e-9


def test_code_objects_are_cached_on_disk(ta, tmp_path):
    """A second construction of the same residual loads the code object from the cache: milliseconds, not seconds."""
    import time
    body = CIRCLE + "\n// cache test " + str(time.time_ns())      # a residual no earlier run can have cached
    ta.JitResidual.set_cache_dir(str(tmp_path))
    try:
        t0 = time.perf_counter()
        a = ta.JitResidual(body, n=3, item_scalars=2)
        t1 = time.perf_counter()
        b = ta.JitResidual(body, n=3, item_scalars=2)
        t2 = time.perf_counter()
        assert not a.from_cache and b.from_cache
        assert t2 - t1 < 0.05 < t1 - t0, (t1 - t0, t2 - t1)
        assert len(list(tmp_path.glob("*.toajit"))) == 1
        obs = torch.from_numpy(_circle_obs(3, 10, np.float64)).cuda()
        o = ta.Options()
        o.lm.damping_init = 1e1
        xa, xb = torch.tensor([[0.0, 0.0, 1.0]] * 3, device="cuda", dtype=torch.float64), torch.tensor([[0.0, 0.0, 1.0]] * 3, device="cuda", dtype=torch.float64)
        oa, ob = ta.Optimize(xa, a.bind(obs), o), ta.Optimize(xb, b.bind(obs), o)
        torch.cuda.synchronize()
        assert torch.equal(xa, xb) and torch.equal(oa.num_iters, ob.num_iters)
        c = ta.JitResidual(body, n=3, item_scalars=2, dtype=torch.float32)          # another instantiation: its own entry
        assert not c.from_cache
        ta.JitResidual.set_cache_dir("")
        d = ta.JitResidual(body, n=3, item_scalars=2)
        assert not d.from_cache                                                      # "" = no cache
    finally:
        ta.JitResidual.set_cache_dir(None)


REPROJ = """
const S X = x[0] * p[0] + x[1] * p[1] + x[2] * p[2] + x[9];
const S Y = x[3] * p[0] + x[4] * p[1] + x[5] * p[2] + x[10];
const S Z = x[6] * p[0] + x[7] * p[1] + x[8] * p[2] + x[11];
r[0] = h[0] * X / Z + h[1] - p[3];
r[1] = h[0] * Y / Z + h[2] - p[4];
"""


@pytest.mark.parametrize("dtype,tdt,npts,P", [(np.float64, torch.float64, 25000, 1), (np.float64, torch.float64, 4096, 3),
                                               (np.float32, torch.float32, 3000, 2)])
def test_single_problem_wide_m_reprojection_supplied_as_source(ta, oracle, dtype, tdt, npts, P):
    """BASELINE config 5 — ONE pose, 25 000 points = 50 000 residuals (benchmarks/…; tests/sophus.cpp's `Optimize(pose, lambda)`
    shape) — with the reprojection residual handed over as TEXT on the SE3 manifold: the run-time model takes the row-split form
    (a wavefront per chunk of items, one persistent launch) by itself, and lands on the oracle's trajectory like the compiled-in
    SE3Reproj model does; every execution form (automatic, explicit chunk counts, launch-per-iteration, one wavefront) agrees."""
    data, p0, pstar = oracle.synth_se3_reproj(P, npts, dtype, seed=4)
    o = ta.Options()
    ref = oracle.se3_reproj_lm(data, p0, npts, o.to_pod())
    fit = ta.JitResidual(REPROJ, n=6, item_scalars=5, residuals_per_item=2, header_scalars=8, dtype=tdt, manifold="se3")
    d = torch.from_numpy(data).cuda()
    model = fit.bind(d[:, 8:].reshape(P, npts, 5).contiguous(), header=d[:, :8].contiguous())
    ctx = ta.api.default_context()
    tol_x = 1e-9 if dtype == np.float64 else 5e-4
    results = []
    for splits, tune in ((None, {}), (0, {}), (7, {}), (16, dict(wide_multilaunch=1)), (None, dict(wide_no_autosplit=1))):
        x = torch.from_numpy(p0.copy()).cuda()
        with ctx.tuning(**tune):
            out = ta.Optimize(x, model, o, splits=splits)
        torch.cuda.synchronize()
        xg, stop = x.cpu().numpy(), out.stop_reason.cpu().numpy()
        assert np.abs(xg - ref["x"]).max() < tol_x, (splits, tune)
        if dtype == np.float64:
            assert np.array_equal(stop, ref["stop"]) and np.array_equal(out.num_iters.cpu().numpy(), ref["iters"]), (splits, tune)
            assert np.allclose(out.final_cost.cpu().numpy(), ref["cost"], rtol=1e-10)
            assert np.allclose(out.final_hessian.cpu().numpy(), ref["H"], rtol=1e-9, atol=1e-6 * np.abs(ref["H"]).max())
        else:
            assert (stop >= 0).all()
        results.append(xg)
    # the compiled-in model on the same bytes
    x = torch.from_numpy(p0.copy()).cuda()
    ta.Optimize(x, ta.SE3Reproj(d, npts), o)
    assert np.abs(x.cpu().numpy() - results[0]).max() < tol_x
    # run to run: the row-split form folds its partials in fixed order
    x = torch.from_numpy(p0.copy()).cuda()
    ta.Optimize(x, model, o)
    assert np.array_equal(x.cpu().numpy(), results[0])


def _two_residual_body(n):
    """An item = (a [n], c [n], b0, b1): r0 = a.x + 0.1 sin(a.x) - b0, r1 = c.x + 0.1 sin(c.x) - b1 — two DenseRow rows per item."""
    return (f"S t = x[0] * p[0]; S u = x[0] * p[{n}];\n#pragma unroll 2\nfor (int j = 1; j < {n}; ++j) {{ t = t + x[j] * p[j]; u = u + x[j] * p[{n} + j]; }}\n"
            f"r[0] = t + T(0.1) * sin(t) - p[{2 * n}];\nr[1] = u + T(0.1) * sin(u) - p[{2 * n + 1}];")


@pytest.mark.parametrize("n,items,dtype,tdt", [(20, 160, np.float64, torch.float64), (50, 200, np.float32, torch.float32)])
def test_wide_parameter_blocks_with_vector_residuals_row_split_and_stepping(ta, oracle, n, items, dtype, tdt):
    """Round 5 (VERDICT r04 "missing" #3): run-time models beyond 12 parameters were one residual per item, whole-solve only.
    Now an item may carry several residuals (optimize_autodiff.h:123-164 takes vector residuals of any width), and the model runs
    in the row-split form and in the stepping form (`stop_callback*`, `max_duration_ms`).  The two-residual item below is two
    DenseRow rows, so the oracle for m = 2 * items rows is the reference: (g, H, cost), the LM trajectory, every execution form."""
    from parity import check_trajectories, gpu_dict
    P, m = 4, 2 * items
    A, b, x0, xs = oracle.synth_dense_row(P, n, m, dtype, seed=300 + n)
    fit = ta.JitResidual(_two_residual_body(n), n=n, item_scalars=2 * n + 2, residuals_per_item=2, dtype=tdt)
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
    st = check_trajectories(gpu_dict(out, x), ref, dtype, opts.to_pod(), label=f"JIT n = {n}, two residuals per item")
    assert st["full"] + st["ties"] == P
    # the row-split form: chunks on item boundaries (multiples of lcm(16, 2) rows), every chunk count the one-wavefront result
    for splits in (1, 3, 7):
        x2 = torch.from_numpy(x0.copy()).cuda()
        o2 = ta.Optimize(x2, model, opts, history=True, splits=splits)
        torch.cuda.synchronize()
        st2 = check_trajectories(gpu_dict(o2, x2), ref, dtype, opts.to_pod(), label=f"JIT n = {n} splits = {splits}")
        assert st2["full"] + st2["ties"] == P
    # the stepping form: stepping to the end is the row-split form with one chunk, bit for bit
    x3 = torch.from_numpy(x0.copy()).cuda()
    o3 = ta.Optimize(x3, model, opts, history=True, splits=1)
    x4 = torch.from_numpy(x0.copy()).cuda()
    opt = ta.Optimizer(x4, model, opts, history=True)
    for _ in range(opts.max_iters + 3):
        if opt.Step() == 0:
            break
    torch.cuda.synchronize()
    assert torch.equal(x4, x3) and torch.equal(opt.out.num_iters, o3.num_iters) and torch.equal(opt.out.errs, o3.errs)
    assert torch.equal(opt.out.stop_reason, o3.stop_reason)


def test_row_split_chunks_fall_on_item_boundaries(ta, oracle):
    """Three residuals per item: chunks are multiples of lcm(16, 3) = 48 rows, so no item straddles two chunks.  Every chunk
    count — more chunks than the rows allow included — gives the one-wavefront result up to the fold order; m = 0 (mod 48) or not."""
    body = "r[0] = x[0] * p[0] + x[1] - p[1];\nr[1] = x[2] * p[0] * p[0] - p[2] + x[1];\nr[2] = sin(x[0]) * p[0] + x[3] - p[3];"
    fit = ta.JitResidual(body, n=4, item_scalars=4, residuals_per_item=3)
    rng = np.random.default_rng(5)
    for items in (700, 768, 17):
        P = 2
        xs = rng.uniform(-1, 1, (P, 4))
        t = rng.uniform(-2, 2, (P, items))
        d = np.stack([t, xs[:, :1] * t + xs[:, 1:2], xs[:, 2:3] * t * t + xs[:, 1:2], np.sin(xs[:, :1]) * t + xs[:, 3:4]], -1)
        d[..., 1:] += 1e-3 * rng.standard_normal(d[..., 1:].shape)
        model = fit.bind(torch.from_numpy(d).cuda())
        x0 = xs + 0.2 * rng.uniform(-1, 1, xs.shape)
        ctx = ta.api.default_context()
        with ctx.tuning(wide_no_autosplit=1):
            xr = torch.from_numpy(x0.copy()).cuda()
            ref = ta.Optimize(xr, model, ta.Options())
        for splits in (0, 1, 3, 5, 64, 1000):
            x = torch.from_numpy(x0.copy()).cuda()
            out = ta.Optimize(x, model, ta.Options(), splits=splits)
            torch.cuda.synchronize()
            assert float((x - xr).abs().max()) < 1e-9, (items, splits)
            assert torch.equal(out.stop_reason, ref.stop_reason) and torch.equal(out.num_iters, ref.num_iters), (items, splits)
            assert np.allclose(out.final_cost.cpu().numpy(), ref.final_cost.cpu().numpy(), rtol=1e-10)
    # an empty batch is a no-op
    x = torch.zeros(0, 4, dtype=torch.float64, device="cuda")
    ta.Optimize(x, fit.bind(torch.zeros(0, 5, 4, dtype=torch.float64, device="cuda")), ta.Options(), splits=2)


def test_stepping_form_and_stop_controls_for_a_residual_supplied_as_text(ta, oracle):
    """`optimizer.Step(x, acc, out)` and Options::stop_callback / stop_callback2 / max_duration_ms (optimizer.h:302-305,331-539,
    options.h:96-106) for a run-time model: stepping to the end reproduces Optimize; a callback stops the named problems after
    that iteration with kUserStopped and a finalised Output row; a time limit gives kTimedOut."""
    P, npts = 9, 40
    obs = _circle_obs(P, npts, np.float64, seed=3)
    x0 = np.tile(np.array([0, 0, 1], np.float64), (P, 1)) + 0.1 * np.random.default_rng(1).uniform(-1, 1, (P, 3))
    o = ta.Options(); o.lm.damping_init = 1e1
    fit = ta.JitResidual(CIRCLE, n=3, item_scalars=2)
    model = fit.bind(torch.from_numpy(obs).cuda())
    xr = torch.from_numpy(x0.copy()).cuda()
    ref = ta.Optimize(xr, model, o, history=True)
    x = torch.from_numpy(x0.copy()).cuda()
    opt = ta.Optimizer(x, model, o, history=True)
    steps = 0
    while opt.Step() > 0:
        steps += 1
        assert steps < o.max_iters + 3
    torch.cuda.synchronize()
    assert torch.equal(opt.out.stop_reason, ref.stop_reason) and torch.equal(opt.out.num_iters, ref.num_iters)
    assert float((x - xr).abs().max()) < 1e-10 and np.allclose(opt.out.final_cost.cpu().numpy(), ref.final_cost.cpu().numpy(), rtol=1e-9)
    # callbacks: from the second pass on, stop the problems whose cost is above the median
    seen = []
    e1 = ref.errs.cpu().numpy()[:, 1]
    thr = float(np.median(e1))

    def cb(err, dx2, g2):
        seen.append((err, dx2, g2))
        return len(seen) > P and err > thr

    o2 = ta.Options(); o2.lm.damping_init = 1e1; o2.stop_callback = cb
    x = torch.from_numpy(x0.copy()).cuda()
    out = ta.Optimize(x, model, o2, history=True)
    torch.cuda.synchronize()
    stop, iters = out.stop_reason.cpu().numpy(), out.num_iters.cpu().numpy()
    user = e1 > thr
    assert user.sum() == P // 2
    assert (stop[user] == int(ta.StopReason.kUserStopped)).all() and (iters[user] == 2).all()
    assert np.array_equal(stop[~user], ref.stop_reason.cpu().numpy()[~user]) and np.array_equal(iters[~user], ref.num_iters.cpu().numpy()[~user])
    assert np.allclose(sorted(c[0] for c in seen[:P]), sorted(ref.errs.cpu().numpy()[:, 0]), rtol=1e-9)
    assert (out.final_cost.cpu().numpy()[user] > 0).all() and float(out.final_hessian[torch.from_numpy(user).cuda()].abs().sum()) > 0
    got = []
    o3 = ta.Options(); o3.stop_callback2 = lambda err, dx, g: (got.append((dx.shape, g.shape)) or True)
    x = torch.from_numpy(x0.copy()).cuda()
    out = ta.Optimize(x, model, o3)
    assert (out.stop_reason.cpu().numpy() == int(ta.StopReason.kUserStopped)).all() and got == [((3,), (3,))] * P
    o4 = ta.Options(); o4.max_duration_ms = 1e-6
    x = torch.from_numpy(x0.copy()).cuda()
    out = ta.Optimize(x, model, o4)
    assert (out.stop_reason.cpu().numpy() == int(ta.StopReason.kTimedOut)).all() and (out.num_iters.cpu().numpy() == 1).all()
    # the SE3 pose prior as text, stepped: lands where the one-launch solve lands
    rng = np.random.default_rng(7)
    ident = np.tile(np.concatenate([np.eye(3).reshape(-1), np.zeros(3)]), (4, 1))
    hdr = torch.from_numpy(oracle.se3_plus(ident, 0.6 * rng.uniform(-1, 1, (4, 6)))).cuda()
    prior = ta.JitResidual(SE3_PRIOR, n=6, item_scalars=0, residuals_per_item=6, header_scalars=12, manifold="se3").bind(None, header=hdr)
    xa = torch.from_numpy(ident.copy()).cuda(); xb = torch.from_numpy(ident.copy()).cuda()
    ra = ta.Optimize(xa, prior, ta.Options())
    rb = ta.Optimizer(xb, prior, ta.Options())()
    assert torch.equal(ra.stop_reason, rb.stop_reason) and torch.equal(ra.num_iters, rb.num_iters) and float((xa - xb).abs().max()) < 1e-10
    # (round 5: models beyond 12 parameters step too — test_wide_parameter_blocks_with_vector_residuals_row_split_and_stepping)


# ---------------------------------------------------------------------------------------------------------------------------
# Round 5 (VERDICT r04 "missing" #4): a USER manifold as text — the reference's traits::params_trait<T> extension point
# ---------------------------------------------------------------------------------------------------------------------------
SO2_PLUS = """
// x = (cos t, sin t) on the unit circle, d[0] = the angle increment: x (+) d = the rotation of x by d
const S c = cos(d[0]), s = sin(d[0]);
xp[0] = x[0] * c - x[1] * s;
xp[1] = x[1] * c + x[0] * s;
"""
SO2_RESIDUAL = "r[0] = x[0] * p[0] - x[1] * p[1] - p[2];\nr[1] = x[1] * p[0] + x[0] * p[1] - p[3];"        # R(x) a - b
ANGLE_RESIDUAL = "const S c = cos(x[0]), s = sin(x[0]);\nr[0] = c * p[0] - s * p[1] - p[2];\nr[1] = s * p[0] + c * p[1] - p[3];"

SE3_PLUS = """
// pose * exp(d), d = (upsilon, omega) in Sophus order; x = R row-major (9) + t (3)
const S wx = d[3], wy = d[4], wz = d[5];
const S t2 = wx * wx + wy * wy + wz * wz;
S A, B, Cc;
if (t2 < T(1e-10)) { A = T(1) - t2 / T(6); B = T(0.5) - t2 / T(24); Cc = T(1) / T(6) - t2 / T(120); }
else { const S th = sqrt(t2); A = sin(th) / th; B = (T(1) - cos(th)) / t2; Cc = (th - sin(th)) / (t2 * th); }
S Rd[9];
Rd[0] = T(1) - B * (wy * wy + wz * wz); Rd[1] = B * wx * wy - A * wz;          Rd[2] = A * wy + B * wx * wz;
Rd[3] = A * wz + B * wx * wy;          Rd[4] = T(1) - B * (wx * wx + wz * wz); Rd[5] = B * wy * wz - A * wx;
Rd[6] = B * wx * wz - A * wy;          Rd[7] = A * wx + B * wy * wz;          Rd[8] = T(1) - B * (wx * wx + wy * wy);
S c1[3], c2[3], td[3];
c1[0] = wy * d[2] - wz * d[1]; c1[1] = wz * d[0] - wx * d[2]; c1[2] = wx * d[1] - wy * d[0];
c2[0] = wy * c1[2] - wz * c1[1]; c2[1] = wz * c1[0] - wx * c1[2]; c2[2] = wx * c1[1] - wy * c1[0];
for (int i = 0; i < 3; ++i) td[i] = d[i] + B * c1[i] + Cc * c2[i];
for (int i = 0; i < 3; ++i) {
  for (int j = 0; j < 3; ++j) xp[3 * i + j] = x[3 * i] * Rd[j] + x[3 * i + 1] * Rd[3 + j] + x[3 * i + 2] * Rd[6 + j];
  xp[9 + i] = x[3 * i] * td[0] + x[3 * i + 1] * td[1] + x[3 * i + 2] * td[2] + x[9 + i];
}
"""


@pytest.mark.parametrize("tdt", [torch.float64, torch.float32])
def test_user_manifold_supplied_as_text(ta, oracle, tdt):
    """(i) The unit circle with x (+) d = rotation by d, against the same fit parametrised by the angle itself (Euclidean):
    x (+) d is t + d exactly, so the two solves take the same steps — cost / accept histories, iteration counts and StopReasons
    equal, the final points the same angle.  (ii) The SE3 pose prior of tests/sophus.cpp:26-44 with pose * exp(d) written out as
    TEXT, against the built-in TOA_MANIFOLD_SE3: same trajectories.  Every execution form (one launch, row-split, stepping)."""
    rng = np.random.default_rng(12)
    P, items = 6, 700
    th_true = rng.uniform(-1.0, 1.0, P)
    a = rng.uniform(-1, 1, (P, items, 2))
    c, s = np.cos(th_true)[:, None], np.sin(th_true)[:, None]
    b = np.stack([c * a[..., 0] - s * a[..., 1], s * a[..., 0] + c * a[..., 1]], -1) + 1e-3 * rng.uniform(-1, 1, (P, items, 2))
    data = torch.from_numpy(np.concatenate([a, b], -1)).to(tdt).cuda()
    th0 = th_true + rng.uniform(-0.6, 0.6, P)
    circle = ta.JitResidual(SO2_RESIDUAL, n=1, item_scalars=4, residuals_per_item=2, dtype=tdt, manifold="user", plus_body=SO2_PLUS, x_scalars=2)
    angle = ta.JitResidual(ANGLE_RESIDUAL, n=1, item_scalars=4, residuals_per_item=2, dtype=tdt)
    assert circle.xdim == 2
    opts = ta.Options()
    xa = torch.from_numpy(th0[:, None].copy()).to(tdt).cuda()
    oa = ta.Optimize(xa, angle.bind(data), opts, history=True)
    tol = 1e-9 if tdt == torch.float64 else 2e-4
    for form in ("launch", "split", "step"):
        xc = torch.from_numpy(np.stack([np.cos(th0), np.sin(th0)], -1)).to(tdt).cuda()
        if form == "launch":
            oc = ta.Optimize(xc, circle.bind(data), opts, history=True)
        elif form == "split":
            oc = ta.Optimize(xc, circle.bind(data), opts, history=True, splits=3)
        else:
            oc = ta.Optimizer(xc, circle.bind(data), opts, history=True)()
        torch.cuda.synchronize()
        assert bool((oc.stop_reason > 0).all())
        if tdt == torch.float64:      # (fp32: the two parametrisations round differently at the noise floor: end points only)
            assert torch.equal(oc.num_iters, oa.num_iters) and torch.equal(oc.stop_reason, oa.stop_reason), form
            k = int(oa.num_iters.min())
            assert np.allclose(oc.errs.cpu().numpy()[:, :k], oa.errs.cpu().numpy()[:, :k], rtol=1e-8), form
        ang = np.arctan2(xc[:, 1].double().cpu().numpy(), xc[:, 0].double().cpu().numpy())
        assert np.abs(ang - xa[:, 0].double().cpu().numpy()).max() < tol * 10, form
        assert np.abs(ang - th_true).max() < 1e-3
        assert np.abs((xc.double() ** 2).sum(1).cpu().numpy() - 1).max() < (1e-12 if tdt == torch.float64 else 1e-5)   # stays on the manifold
    # (ii) SE3: the pose prior with the built-in manifold and with pose * exp(d) as text
    ident = np.tile(np.concatenate([np.eye(3).reshape(-1), np.zeros(3)]), (8, 1))
    hdr = torch.from_numpy(oracle.se3_plus(ident, 0.6 * rng.uniform(-1, 1, (8, 6)))).to(tdt).cuda()
    builtin = ta.JitResidual(SE3_PRIOR, n=6, item_scalars=0, residuals_per_item=6, header_scalars=12, manifold="se3", dtype=tdt).bind(None, header=hdr)
    user = ta.JitResidual(SE3_PRIOR, n=6, item_scalars=0, residuals_per_item=6, header_scalars=12, manifold="user", plus_body=SE3_PLUS, x_scalars=12,
                          dtype=tdt).bind(None, header=hdr)
    xb = torch.from_numpy(ident.copy()).to(tdt).cuda()
    xu = torch.from_numpy(ident.copy()).to(tdt).cuda()
    ob = ta.Optimize(xb, builtin, ta.Options(), history=True)
    ou = ta.Optimize(xu, user, ta.Options(), history=True)
    torch.cuda.synchronize()
    assert bool((ou.stop_reason > 0).all()) and torch.equal(ou.stop_reason, ob.stop_reason) and torch.equal(ou.num_iters, ob.num_iters)
    assert float((xu - xb).abs().max()) < (1e-9 if tdt == torch.float64 else 1e-4)
    k = int(ob.num_iters.min())
    assert np.allclose(ou.errs.cpu().numpy()[:, :k], ob.errs.cpu().numpy()[:, :k], rtol=1e-7 if tdt == torch.float64 else 1e-2, atol=1e-12)
    with pytest.raises(Exception):
        ta.JitResidual(SO2_RESIDUAL, n=1, item_scalars=4, residuals_per_item=2, manifold="user")      # no plus body
