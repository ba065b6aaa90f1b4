"""The device dual numbers (csrc/jet.hpp = ceres::Jet<T, N>, include/tinyopt/3rdparty/ceres/jet.h:216-1400) function
by function: value and both partial derivatives through Jet<T, 2> against closed forms (numpy / scipy), like the
reference's own derivative checks; and the wide-block AD model (JetRowModel, "chunked Jets") against the hand-derived
MFMA path and the oracle at the BASELINE shapes."""
import ctypes as C

import numpy as np
import pytest
import torch
from scipy import special

from parity import check_trajectories, gpu_dict

pytestmark = pytest.mark.gpu

LN2, LN10 = np.log(2.0), np.log(10.0)
# fn id -> (domain of a, domain of b, value, d/da, d/db)
CASES = {
    0: ((-2, 2), (-2, 2), lambda a, b: a + b, lambda a, b: 1 + 0 * a, lambda a, b: 1 + 0 * a),
    1: ((-2, 2), (-2, 2), lambda a, b: a - b, lambda a, b: 1 + 0 * a, lambda a, b: -1 + 0 * a),
    2: ((-2, 2), (-2, 2), lambda a, b: a * b, lambda a, b: b, lambda a, b: a),
    3: ((-2, 2), (0.5, 2), lambda a, b: a / b, lambda a, b: 1 / b, lambda a, b: -a / b ** 2),
    4: ((-2, 2), (-2, 2), lambda a, b: np.abs(a), lambda a, b: np.sign(a), lambda a, b: 0 * a),
    5: ((0.1, 3), (-1, 1), lambda a, b: np.log(a), lambda a, b: 1 / a, lambda a, b: 0 * a),
    6: ((-2, 2), (-1, 1), lambda a, b: np.exp(a), lambda a, b: np.exp(a), lambda a, b: 0 * a),
    7: ((0.1, 3), (-1, 1), lambda a, b: np.sqrt(a), lambda a, b: 0.5 / np.sqrt(a), lambda a, b: 0 * a),
    8: ((-3, 3), (-1, 1), lambda a, b: np.cos(a), lambda a, b: -np.sin(a), lambda a, b: 0 * a),
    9: ((-3, 3), (-1, 1), lambda a, b: np.sin(a), lambda a, b: np.cos(a), lambda a, b: 0 * a),
    10: ((-1.2, 1.2), (-1, 1), lambda a, b: np.tan(a), lambda a, b: 1 + np.tan(a) ** 2, lambda a, b: 0 * a),
    11: ((-3, 3), (-1, 1), lambda a, b: np.arctan(a), lambda a, b: 1 / (1 + a * a), lambda a, b: 0 * a),
    12: ((-2, 2), (-1, 1), lambda a, b: np.tanh(a), lambda a, b: 1 - np.tanh(a) ** 2, lambda a, b: 0 * a),
    13: ((-2, 2), (0.3, 2), lambda a, b: np.arctan2(a, b), lambda a, b: b / (a * a + b * b), lambda a, b: -a / (a * a + b * b)),
    14: ((0.2, 3), (-1, 1), lambda a, b: a ** 2.5, lambda a, b: 2.5 * a ** 1.5, lambda a, b: 0 * a),
    15: ((-0.9, 0.9), (-1, 1), lambda a, b: np.arccos(a), lambda a, b: -1 / np.sqrt(1 - a * a), lambda a, b: 0 * a),
    16: ((-0.9, 0.9), (-1, 1), lambda a, b: np.arcsin(a), lambda a, b: 1 / np.sqrt(1 - a * a), lambda a, b: 0 * a),
    17: ((-2, 2), (-1, 1), lambda a, b: np.sinh(a), lambda a, b: np.cosh(a), lambda a, b: 0 * a),
    18: ((-2, 2), (-1, 1), lambda a, b: np.cosh(a), lambda a, b: np.sinh(a), lambda a, b: 0 * a),
    19: ((0.2, 3), (-1, 1), lambda a, b: np.cbrt(a), lambda a, b: 1 / (3 * np.cbrt(a * a)), lambda a, b: 0 * a),
    20: ((-2, 2), (-1, 1), lambda a, b: np.exp2(a), lambda a, b: np.exp2(a) * LN2, lambda a, b: 0 * a),
    21: ((0.2, 3), (-1, 1), lambda a, b: np.log2(a), lambda a, b: 1 / (a * LN2), lambda a, b: 0 * a),
    22: ((0.2, 3), (-1, 1), lambda a, b: np.log10(a), lambda a, b: 1 / (a * LN10), lambda a, b: 0 * a),
    23: ((-0.5, 3), (-1, 1), lambda a, b: np.log1p(a), lambda a, b: 1 / (1 + a), lambda a, b: 0 * a),
    24: ((-2, 2), (-1, 1), lambda a, b: np.expm1(a), lambda a, b: np.exp(a), lambda a, b: 0 * a),
    25: ((0.2, 2), (0.2, 2), lambda a, b: np.hypot(a, b), lambda a, b: a / np.hypot(a, b), lambda a, b: b / np.hypot(a, b)),
    26: ((-2, 2), (-2, 2), lambda a, b: np.fmax(a, b), lambda a, b: 1.0 * (a > b), lambda a, b: 1.0 * (b > a)),
    27: ((-2, 2), (-2, 2), lambda a, b: np.fmin(a, b), lambda a, b: 1.0 * (a < b), lambda a, b: 1.0 * (b < a)),
    28: ((-2, 2), (-1, 1), lambda a, b: special.erf(a), lambda a, b: 2 / np.sqrt(np.pi) * np.exp(-a * a), lambda a, b: 0 * a),
    29: ((-2, 2), (-1, 1), lambda a, b: special.erfc(a), lambda a, b: -2 / np.sqrt(np.pi) * np.exp(-a * a), lambda a, b: 0 * a),
    30: ((0.3, 2), (-2, 2), lambda a, b: a ** b, lambda a, b: b * a ** (b - 1), lambda a, b: a ** b * np.log(a)),
    31: ((0.3, 2), (-2, 2), lambda a, b: a ** b, lambda a, b: 0 * a, lambda a, b: a ** b * np.log(a)),
    32: ((-2, 2), (-2, 2), lambda a, b: a * b + a, lambda a, b: b + 1, lambda a, b: a),
    33: ((-2, 2), (-2, 2), lambda a, b: np.maximum(a - b, 0), lambda a, b: 1.0 * (a > b), lambda a, b: -1.0 * (a > b)),
    34: ((-2.4, 2.4), (-1, 1), lambda a, b: np.floor(a), lambda a, b: 0 * a, lambda a, b: 0 * a),
    35: ((-2.4, 2.4), (-1, 1), lambda a, b: np.ceil(a), lambda a, b: 0 * a, lambda a, b: 0 * a),
    36: ((-2, 2), (-1, 1), lambda a, b: a * a, lambda a, b: 2 * a, lambda a, b: 0 * a),
    37: ((0.2, 2), (-2, 2), lambda a, b: np.copysign(a, b), lambda a, b: np.sign(a) * np.sign(b), lambda a, b: 0 * a),
    38: ((0.3, 2), (-2, 2), lambda a, b: 2 / a + a / 4 - 3 * b, lambda a, b: -2 / a ** 2 + 0.25, lambda a, b: -3 + 0 * a),
    39: ((0.2, 2), (0.2, 2), lambda a, b: np.sqrt(a * a + b * b + a * a * b * b),
         lambda a, b: (a + a * b * b) / np.sqrt(a * a + b * b + a * a * b * b),
         lambda a, b: (b + a * a * b) / np.sqrt(a * a + b * b + a * a * b * b)),
    # Bessel functions of the first kind (jet.h:919-1009): J0' = -J1, Jn' = (J(n-1) - J(n+1)) / 2
    40: ((-6, 6), (-1, 1), lambda a, b: special.jv(0, a), lambda a, b: -special.jv(1, a), lambda a, b: 0 * a),
    41: ((-6, 6), (-1, 1), lambda a, b: special.jv(1, a), lambda a, b: 0.5 * (special.jv(0, a) - special.jv(2, a)), lambda a, b: 0 * a),
    42: ((-6, 6), (-1, 1), lambda a, b: special.jv(3, a), lambda a, b: 0.5 * (special.jv(2, a) - special.jv(4, a)), lambda a, b: 0 * a),
    43: ((-6, 6), (-6, 6), lambda a, b: special.jv(0, a) + special.jv(2, b), lambda a, b: -special.jv(1, a),
         lambda a, b: 0.5 * (special.jv(1, b) - special.jv(3, b))),
    # lerp(a, b, t = a b) = a + a b (b - a) and midpoint (jet.h:1171-1218)
    44: ((-2, 2), (-2, 2), lambda a, b: a + a * b * (b - a), lambda a, b: 1 + b * b - 2 * a * b, lambda a, b: 2 * a * b - a * a),
    45: ((-2, 2), (-2, 2), lambda a, b: 0.5 * (a + b), lambda a, b: 0.5 + 0 * a, lambda a, b: 0.5 + 0 * a),
}


def test_jet_classification_and_comparison(ta):
    """jet.h:1011-1168: isfinite / isinf / isnan / isnormal / signbit / fpclassify and the is* comparisons look at the scalar
    part only."""
    ctx = ta.api.default_context()
    a = np.array([1.5, -2.0, np.inf, -np.inf, np.nan, 0.0, 1e-310, 3.0])
    b = np.array([2.0, -2.0, 1.0, 1.0, 1.0, np.nan, 1.0, 1.0])
    ad, bd = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    out = torch.zeros(len(a), 3, dtype=torch.float64, device="cuda")
    ta._capi.check(ctx.lib.toa_jet_eval(ctx.h, 46, 1, len(a), ad.data_ptr(), bd.data_ptr(), out.data_ptr()))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert (got[:, 1:] == 0).all()
    cls = {np.nan: 0}
    for i, (x, y) in enumerate(zip(a, b)):
        m = int(got[i, 0])
        normal = np.isfinite(x) and abs(x) >= 2.2250738585072014e-308
        fpc = 0 if np.isnan(x) else 1 if np.isinf(x) else 2 if x == 0 else 4 if normal else 3
        want = ((1 if np.isfinite(x) else 0) | (2 if np.isinf(x) else 0) | (4 if np.isnan(x) else 0) | (8 if normal else 0) |
                (16 if np.signbit(x) else 0) | (32 if x < y else 0) | (64 if x > y else 0) | (128 if x <= y else 0) |
                (256 if x >= y else 0) | (512 if (x < y or x > y) else 0) | (1024 if (np.isnan(x) or np.isnan(y)) else 0) | (fpc << 11))
        assert m == want, (i, x, y, m, want)


@pytest.mark.parametrize("tdt,tol", [(torch.float64, 1e-12), (torch.float32, 2e-5)])
def test_every_jet_function(ta, tdt, tol):
    ctx = ta.api.default_context()
    rng = np.random.default_rng(0)
    n = 512
    for fn, (da, db, f, fa, fb) in CASES.items():
        a = rng.uniform(da[0], da[1], n)
        b = rng.uniform(db[0], db[1], n)
        if fn in (26, 27, 33):
            a[:8] = b[:8]                     # equality: fmax / fmin average the two Jets (jet.h:800-880)
        ad, bd = torch.from_numpy(a).to(tdt).cuda(), torch.from_numpy(b).to(tdt).cuda()
        out = torch.zeros(n, 3, dtype=tdt, device="cuda")
        ta._capi.check(ctx.lib.toa_jet_eval(ctx.h, fn, 0 if tdt == torch.float32 else 1, n, ad.data_ptr(), bd.data_ptr(), out.data_ptr()))
        torch.cuda.synchronize()
        got = out.cpu().numpy().astype(np.float64)
        a64, b64 = ad.cpu().numpy().astype(np.float64), bd.cpu().numpy().astype(np.float64)
        want = np.stack([f(a64, b64), fa(a64, b64), fb(a64, b64)], 1)
        if fn in (26, 27):
            want[:8, 1:] = 0.5                # averaged on equality
        scale = np.maximum(np.abs(want), 1.0)
        t = max(tol, 1e-10) if 40 <= fn <= 43 else tol     # the device library's jn comes from a recurrence
        assert (np.abs(got - want) <= t * scale).all(), (fn, np.abs(got - want).max())
    assert ctx.lib.toa_jet_eval(ctx.h, 99, 1, 1, ad.data_ptr(), bd.data_ptr(), out.data_ptr()) != 0


def test_jet_special_cases(ta):
    """pow's special cases (jet.h:1275-1400): 0^g for g > 1 has a zero derivative, 0^1 = df; a negative base with an integer
    exponent differentiates in the base only and is NaN in the exponent's direction; fdim / fmax treat NaN as the reference does."""
    ctx = ta.api.default_context()

    def run(fn, a, b):
        ad = torch.tensor(a, dtype=torch.float64, device="cuda")
        bd = torch.tensor(b, dtype=torch.float64, device="cuda")
        out = torch.zeros(len(a), 3, dtype=torch.float64, device="cuda")
        ta._capi.check(ctx.lib.toa_jet_eval(ctx.h, fn, 1, len(a), ad.data_ptr(), bd.data_ptr(), out.data_ptr()))
        return out.cpu().numpy()

    o = run(30, [0.0, 0.0, -2.0, -2.0], [3.0, 1.0, 3.0, 2.0])
    assert np.array_equal(o[0], [0.0, 0.0, 0.0])                       # case 2
    assert np.array_equal(o[1], [0.0, 1.0, 0.0])                       # case 3: 0^1 -> df
    assert o[2][0] == -8.0 and o[2][1] == 12.0 and np.isnan(o[2][2])   # cases 7 / 8
    assert o[3][0] == 4.0 and o[3][1] == -4.0 and np.isnan(o[3][2])
    o = run(31, [0.0, -2.0], [2.0, 3.0])
    assert np.array_equal(o[0], [0.0, 0.0, 0.0]) and o[1][0] == -8.0 and np.isnan(o[1][2])
    o = run(26, [np.nan, 1.0], [2.0, np.nan])
    assert np.array_equal(o[0], [2.0, 0.0, 1.0]) and np.array_equal(o[1], [1.0, 1.0, 0.0])   # NaN = missing data
    o = run(33, [np.nan], [1.0])
    assert np.isnan(o[0][0])


@pytest.mark.parametrize("dtype,n,m,P", [(np.float64, 12, 500, 24), (np.float64, 50, 402, 6), (np.float32, 50, 2000, 6),
                                         (np.float32, 12, 203, 9)])
def test_wide_block_ad_equals_analytic_path(ta, oracle, dtype, n, m, P):
    """TOA_MODEL_DENSE_ROW_AD: the DenseRow residual written as r(x) only at the C3 (n = 12) and C4 (n = 50) shapes,
    differentiated on the device by chunked Jets in the matrix cores' operand layout: (g, H, cost) equal to the hand-derived
    MFMA path and to the oracle, the cost-only pass too, and the whole LM trajectory against the oracle (tie-aware)."""
    A, b, x0, xs = oracle.synth_dense_row(P, n, m, dtype, seed=31)
    Ad, bd, xd = torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda(), torch.from_numpy(x0).cuda()
    ad = ta.DenseRowAD(Ad, bd)
    an = ta.DenseRow.from_arrays(Ad, bd)
    g1, H1, c1, n1 = ta.accumulate(ad, xd)
    g2, H2, c2, _ = ta.accumulate(an, xd)
    g_ref, H_ref, c_ref, _ = oracle.dense_row_accumulate(A, b, x0)
    tol = 1e-10 if dtype == np.float64 else 2e-4
    for got, want in ((g1, g_ref), (H1, H_ref)):
        assert np.abs(got.cpu().numpy() - want).max() <= tol * np.abs(want).max()
    assert np.allclose(c1.cpu().numpy(), c_ref, rtol=tol) and (n1.cpu().numpy() == m).all()
    assert np.abs((g1 - g2).cpu().numpy()).max() <= tol * np.abs(g_ref).max()
    assert np.abs((H1 - H2).cpu().numpy()).max() <= tol * np.abs(H_ref).max()
    Hn = H1.cpu().numpy()
    assert np.array_equal(Hn, np.transpose(Hn, (0, 2, 1)))
    c1e = ta.accumulate(ad, xd, want_grad=False)[2]
    assert np.allclose(c1e.cpu().numpy(), c_ref, rtol=tol)
    for opts in (ta.Options.benchmark(), ta.Options()):
        ref = oracle.dense_row_lm(A, b, x0, opts.to_pod(), history=True)
        x = xd.clone()
        out = ta.Optimize(x, ad, opts, history=True)
        torch.cuda.synchronize()
        refd = dict(errs=ref["errs"], succ=ref["succ"], iters=ref["iters"], stop=ref["stop"], x=ref["x"], cost=ref["cost"],
                    fails=ref["fails"], deltas2=ref["deltas2"])
        st = check_trajectories(gpu_dict(out, x), refd, dtype, opts.to_pod(), label=f"AD n={n}")
        assert st["full"] + st["ties"] == P
        assert np.abs(x.cpu().numpy() - xs).max() < 5e-3
    # the stepping form runs the same model
    x = xd.clone()
    opt = ta.Optimizer(x, ad, ta.Options())
    opt()
    assert (opt.out.stop_reason.cpu().numpy() > 0).all() and np.abs(x.cpu().numpy() - xs).max() < 5e-3
    with pytest.raises(AssertionError):
        ta.DenseRowAD(Ad[:, :, :7].contiguous(), bd)           # only the instantiated parameter counts
