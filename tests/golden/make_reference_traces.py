#!/usr/bin/env python3
"""SECOND READING of tinyopt's Levenberg-Marquardt state machine, as committed fixtures.

Why this file exists ("weak" #1): the reference's own tests only assert end points (Rosenbrock reaches (1, 1),
...), so the arithmetic THROUGH rejected steps — roll-back, re-accumulation at the same point, eval-only iterations with the
stale linearisation re-damped, failed solves re-entering Build inside one iteration — was pinned by ONE reading of
optimizer.h:271-299, 358-393: the one in oracle/lm_oracle.hpp, which sits on both sides of every GPU-vs-oracle comparison.

This is an independent restatement in pure Python (IEEE doubles = the reference's `double` Scalar), written from SURVEY.md
Appendix A and the reference headers directly — NOT from anything under oracle/ (it imports nothing of this repo):

    Optimizer_::OptimizeAcc   include/tinyopt/optimizers/optimizer.h:242-327
    Optimizer_::Step          include/tinyopt/optimizers/optimizer.h:331-539
    SolverLM                  include/tinyopt/solvers/lm.h:46-171
    SolverGN::Build / Solve   include/tinyopt/solvers/gn.h:117-171
    SolverBase                include/tinyopt/solvers/base.h:29-45
    Cost                      include/tinyopt/cost.h:18-97
    Options defaults          include/tinyopt/optimizers/options.h:43-139
    SolveLDLT                 include/tinyopt/math.h:232-240  (+ Eigen 3.4 LDLT, restated from its published algorithm)
    the cost callbacks        tests/optimize_easy.cpp:35-221 (Rosenbrock, plateau, Powell — manual Accumulate callbacks)
                              tests/optimize_hard.cpp:34-102 (Beale, Himmelblau — residual VECTORS: the AD bridge folds them,
                              diff/optimize_autodiff.h:151-164: grad = J^T r, H = J^T J, Cost(||r||^2, r.size()))

Round 4 (VERDICT r03 "weak" #1 ii): the branches the first 30 cases left to one reading — use_step_quality_approx (lm.h:126-129),
grad_clipping (base.h:29-38 through lm.h:79), check_min_H_diag (lm.h:82-86), use_ldlt = false (gn.h:157-162), the three
NormalizeCost flags (base.h:41-45), vector-residual costs with num_residuals > 1 — and "float32" cases: the same reading in
doubles, held against the fp32 oracle / device with fp32 tolerances, emitted only when every decision is further than 1e-3
from flipping and the trace is short enough to stay clear of the fp32 noise floor.

It emits tests/golden/reference_traces.json: per case the options, the start, and per loop pass the cost, |dx|^2, the
accept flag, lambda AFTER the pass, x AFTER the pass, whether the pass rebuilt the linear system, plus the final Output.
tests/test_cpu_oracle.py holds the C++ oracle to these traces; tests/test_gpu_traces.py replays them on the device TestFn
models.  A case is only emitted if its accept / reject sequence is ROBUST: every decision `err < final_cost` is either
the exact zero of a roll-back that restored x bit for bit, or further than 1e-9 (relative) from flipping, and a second run
with every cost perturbed by +-1e-13 relative takes the same decisions.  (Another implementation's roll-back may still miss
the exact zero where this one hit it: the comparators treat a flag that differs AT the last accepted cost as a proven tie.)

Run:  python tests/golden/make_reference_traces.py   (writes the JSON next to this file; needs numpy only)
"""
import json
import math
import os

import numpy as np

DBL_MAX = 1.7976931348623157e308
FLT_MIN_NORMAL_DBL = 2.2250738585072014e-308   # std::numeric_limits<double>::min()


def f32(v):
    """A C++ `float` literal/field, as the double it promotes to."""
    return float(np.float32(v))


# ---- StopReason (include/tinyopt/stop_reasons.h:14-43) -----------------------------------------------------------------
kSolverFailed, kSystemHasNaNOrInf, kSkipped, kNone = -3, -2, -1, 0
kMinError, kMinRelError, kMinDeltaNorm, kMinGradNorm, kMaxIters, kMaxNoDecr, kMaxConsecNoDecr = 1, 2, 3, 4, 5, 6, 7


class Options:
    """options.h:43-139 defaults; float fields are floats (promoted in every comparison)."""

    def __init__(self, **kw):
        self.solver = "lm"
        self.check_final_cost = False
        self.use_step_quality_approx = False
        self.grad_clipping = f32(0)
        self.use_ldlt = True
        self.check_min_H_diag = f32(0)
        self.use_squared_norm = True
        self.downscale_by_2 = False
        self.normalize = False
        self.max_iters = 50
        self.min_error = f32(1e-12)
        self.min_rerr_dec = f32(1e-10)
        self.min_step_norm2 = f32(1e-14)
        self.min_grad_norm2 = f32(1e-18)
        self.max_total_failures = 0
        self.max_consec_failures = 5
        self.damping_init = f32(1e-4)
        self.damping_range = (f32(1e-9), f32(1e9))
        self.good_factor = float(np.float32(1.0) / np.float32(3.0))
        self.bad_factor = f32(2.0)
        for k, v in kw.items():
            assert hasattr(self, k), k
            setattr(self, k, f32(v) if isinstance(getattr(self, k), float) and not isinstance(v, bool) else v)


# ---- Eigen 3.4 LDLT (Eigen/src/Cholesky/LDLT.h: ldlt_inplace<Lower>::unblocked, LDLT::_solve_impl) ---------------------
ZERO, POS, NEG, INDEF = 0, 1, 2, 3


def ldlt_factor(A):
    n = len(A)
    M = [row[:] for row in A]            # full symmetric copy (selfadjointView<Upper> assigned to a dense matrix)
    tr = list(range(n))
    sign = ZERO
    ok = True
    found_zero_pivot = False
    if n == 1:
        sign = POS if M[0][0] > 0 else (NEG if M[0][0] < 0 else ZERO)
        return M, tr, sign, True
    for k in range(n):
        big, idx = -1.0, k
        for i in range(k, n):            # first maximum of |diag| over the trailing block
            if abs(M[i][i]) > big:
                big, idx = abs(M[i][i]), i
        tr[k] = idx
        if idx != k:                     # symmetric swap inside the lower triangle
            for j in range(k):
                M[k][j], M[idx][j] = M[idx][j], M[k][j]
            for i in range(idx + 1, n):
                M[i][k], M[i][idx] = M[i][idx], M[i][k]
            M[k][k], M[idx][idx] = M[idx][idx], M[k][k]
            for i in range(k + 1, idx):
                M[i][k], M[idx][i] = M[idx][i], M[i][k]
        rs = n - k - 1
        if k > 0:
            temp = [M[j][j] * M[k][j] for j in range(k)]
            s = 0.0
            for j in range(k):
                s += M[k][j] * temp[j]
            M[k][k] -= s
            for i in range(k + 1, n):
                s = 0.0
                for j in range(k):
                    s += M[i][j] * temp[j]
                M[i][k] -= s
        akk = M[k][k]
        pivot_valid = abs(akk) > 0.0
        if k == 0 and not pivot_valid:   # the whole diagonal is zero
            sign = ZERO
            tr = list(range(n))
            for j in range(n):
                ok = ok and all(M[i][j] == 0.0 for i in range(j + 1, n))
            return M, tr, sign, ok
        if rs > 0 and pivot_valid:
            for i in range(k + 1, n):
                M[i][k] /= akk
        elif rs > 0:
            ok = ok and all(M[i][k] == 0.0 for i in range(k + 1, n))
        if found_zero_pivot and pivot_valid:
            ok = False
        elif not pivot_valid:
            found_zero_pivot = True
        if sign == POS:
            if akk < 0:
                sign = INDEF
        elif sign == NEG:
            if akk > 0:
                sign = INDEF
        elif sign == ZERO:
            sign = POS if akk > 0 else (NEG if akk < 0 else ZERO)
    return M, tr, sign, ok


def ldlt_solve(M, tr, b):
    n = len(M)
    y = b[:]
    for k in range(n):                   # P b
        y[k], y[tr[k]] = y[tr[k]], y[k]
    for i in range(n):                   # L y = .  (unit lower)
        s = y[i]
        for j in range(i):
            s -= M[i][j] * y[j]
        y[i] = s
    for i in range(n):                   # pseudo-inverse of D
        y[i] = y[i] / M[i][i] if abs(M[i][i]) > FLT_MIN_NORMAL_DBL else 0.0
    for i in range(n - 1, -1, -1):       # L^T x = .
        s = y[i]
        for j in range(i + 1, n):
            s -= M[j][i] * y[j]
        y[i] = s
    for k in range(n - 1, -1, -1):       # P^T
        y[k], y[tr[k]] = y[tr[k]], y[k]
    return y


def solve_ldlt(H, b):
    """math.h:232-240: chol.info() == Success && chol.isPositive() ? chol.solve(b) : nullopt."""
    M, tr, sign, ok = ldlt_factor(H)
    if ok and sign in (POS, ZERO):
        return ldlt_solve(M, tr, b)
    return None


# ---- the cost callbacks of tests/optimize_easy.cpp (scalar return => Cost(v, 1), cost.h:22) ------------------------------
def rosenbrock(v, want):
    x, y = v
    term1 = 1.0 - x
    term2 = y - x * x
    g = H = None
    if want:
        g = [-2.0 * term1 - 400.0 * x * term2, 200.0 * term2]
        H = [[2.0 - 400.0 * y + 1200.0 * x * x, -400.0 * x], [-400.0 * x, 200.0]]
    return term1 * term1 + 100.0 * term2 * term2, g, H, 1


def plateau(v, want):
    PI = math.acos(-1.0)
    dx, dy = v[0] - PI, v[1] - PI
    ex = math.exp(-(dx * dx + dy * dy))
    cx, cy, sx, sy = math.cos(v[0]), math.cos(v[1]), math.sin(v[0]), math.sin(v[1])
    cost = 1.0 - (cx * cy * ex)
    g = H = None
    if want:
        g = [cy * ex * (sx + 2.0 * dx * cx), cx * ex * (sy + 2.0 * dy * cy)]
        h00 = cy * ex * (cx - 4.0 * dx * sx + (2.0 - 4.0 * dx * dx) * cx)
        h11 = cx * ex * (cy - 4.0 * dy * sy + (2.0 - 4.0 * dy * dy) * cy)
        h01 = ex * (sx + 2.0 * dx * cx) * (sy + 2.0 * dy * cy)
        H = [[h00, h01], [h01, h11]]
    return cost, g, H, 1


def powell(v, want):
    x1, x2, x3, x4 = v
    t1, t2, t3, t4 = x1 + 10.0 * x2, x3 - x4, x2 - 2.0 * x3, x1 - x4
    g = H = None
    if want:
        g = [2.0 * t1 + 40.0 * math.pow(t4, 3), 20.0 * t1 + 4.0 * math.pow(t3, 3), 10.0 * t2 - 8.0 * math.pow(t3, 3),
             -10.0 * t2 - 40.0 * math.pow(t4, 3)]
        H = [[0.0] * 4 for _ in range(4)]
        H[0][0] = 2.0; H[0][1] = 20.0; H[1][0] = 20.0; H[1][1] = 200.0
        H[2][2] += 10.0; H[2][3] += -10.0; H[3][2] += -10.0; H[3][3] += 10.0
        d3 = 12.0 * t3 * t3
        H[1][1] += d3; H[1][2] += -2.0 * d3; H[2][1] += -2.0 * d3; H[2][2] += 4.0 * d3
        d4 = 120.0 * t4 * t4
        H[0][0] += d4; H[0][3] += -d4; H[3][0] += -d4; H[3][3] += d4
    return t1 * t1 + 5.0 * t2 * t2 + math.pow(t3, 4) + math.pow(t4, 4) * 10.0, g, H, 1


def _from_residuals(r, J, want):
    """What the AD bridge makes of a residual vector (optimize_autodiff.h:151-164): grad = J^T r, H = J^T J (all n x n
    entries), cost = r.squaredNorm() over r.size() residuals."""
    m, n = len(r), len(J[0])
    cost = 0.0
    for v in r:
        cost += v * v
    g = H = None
    if want:
        g = [0.0] * n
        H = [[0.0] * n for _ in range(n)]
        for a in range(n):
            for i in range(m):
                g[a] += J[i][a] * r[i]
            for b in range(n):
                for i in range(m):
                    H[a][b] += J[i][a] * J[i][b]
    return cost, g, H, m


def beale(v, want):                       # tests/optimize_hard.cpp:34-63
    x, y = v
    r = [1.5 - x + x * y, 2.25 - x + x * y * y, 2.625 - x + x * y * y * y]
    J = [[-1.0 + y, x], [-1.0 + y * y, 2.0 * x * y], [-1.0 + y * y * y, 3.0 * x * y * y]]
    return _from_residuals(r, J, want)


def himmelblau(v, want):                  # tests/optimize_hard.cpp:72-102
    x, y = v
    r = [x * x + y - 11.0, x + y * y - 7.0]
    J = [[2.0 * x, 1.0], [1.0, 2.0 * y]]
    return _from_residuals(r, J, want)


FUNCS = {"rosenbrock": rosenbrock, "plateau": plateau, "powell": powell, "beale": beale, "himmelblau": himmelblau}


def solve_general(H, b):
    """gn.h:162: `-H.inverse() * g` without any checks, here as Gaussian elimination with partial pivoting (a singular H gives
    inf / nan, which Step turns into kSystemHasNaNOrInf, optimizer.h:416-425 — no fixture goes there)."""
    n = len(H)
    A = [row[:] + [b[i]] for i, row in enumerate(H)]
    for k in range(n):
        piv = max(range(k, n), key=lambda i: abs(A[i][k]))
        A[k], A[piv] = A[piv], A[k]
        for i in range(k + 1, n):
            f = A[i][k] / A[k][k]
            for j in range(k, n + 1):
                A[i][j] -= f * A[k][j]
    x = [0.0] * n
    for i in range(n - 1, -1, -1):
        s_ = A[i][n]
        for j in range(i + 1, n):
            s_ -= A[i][j] * x[j]
        x[i] = s_ / A[i][i]
    return x


# ---- the solver (lm.h / gn.h / base.h) -----------------------------------------------------------------------------------
class Solver:
    def __init__(self, opt, n):
        self.o, self.n = opt, n
        self.H = [[0.0] * n for _ in range(n)]
        self.g = [0.0] * n
        self.cost, self.nres = 0.0, 0                       # Cost{} default
        self.lam = opt.damping_init                         # lm.h:46-52 reset()
        self.prev_lam = 0.0
        self.bad_f = opt.bad_factor
        self.rebuild = True

    def normalize(self, c, nres):                           # base.h:41-45
        if not self.o.use_squared_norm:
            c = math.sqrt(c)
        if self.o.downscale_by_2:
            c *= f32(0.5)
        if self.o.normalize and nres > 0:
            c /= nres
        return c

    def valid(self):                                        # cost.h:83
        return self.nres > 0 and self.cost != DBL_MAX

    def build(self, x, fn, perturb):
        n, o = self.n, self.o
        lm = o.solver == "lm"
        if not lm or self.rebuild:                          # lm.h:61-93 / gn.h:117-147
            self.H = [[0.0] * n for _ in range(n)]          # clear()
            self.g = [0.0] * n
            c, g, H, nres = fn(x, True)
            self.H, self.g = [r[:] for r in H], g[:]
            self.cost, self.nres = self.normalize(perturb(c), nres), nres
            if not self.valid():
                return False
            if o.grad_clipping != 0:                        # base.h:29-38
                mm = o.grad_clipping
                self.g = [min(max(v, -mm), mm) for v in self.g]
            if o.check_min_H_diag > 0 and any(abs(self.H[i][i]) < o.check_min_H_diag for i in range(n)):
                return False
        else:                                               # lm.h:94-105: Evaluate(x, acc, save = true)
            c, _, _, nres = fn(x, False)
            self.cost, self.nres = self.normalize(perturb(c), nres), nres
            if not self.valid():
                return False
        if lm and self.lam > 0.0:                           # lm.h:107-117, s is a double
            s = (1.0 + self.lam) if self.rebuild else (1.0 + self.lam) / (1.0 + self.prev_lam)
            for i in range(n):
                self.H[i][i] *= s
        return True

    def solve(self):                                        # gn.h:150-171
        if not self.valid():
            return None
        if self.o.use_ldlt:
            return solve_ldlt(self.H, [-v for v in self.g])
        if self.n == 1:                                     # gn.h:157-161 (Dims == 1)
            return [-(1.0 / self.H[0][0]) * self.g[0] if self.H[0][0] > f32(1e-7) else 0.0]
        return solve_general(self.H, [-v for v in self.g])  # gn.h:162

    def clampl(self, v):
        return min(max(v, self.o.damping_range[0]), self.o.damping_range[1])

    def good_step(self, quality):                           # lm.h:123-137
        if self.o.solver != "lm":
            return
        s = self.o.good_factor
        if quality != 0.0:
            s = max(s, 1.0 - math.pow(2.0 * quality - 1.0, 3.0))
        if self.bad_f != self.o.bad_factor:
            s /= self.bad_f
        self.prev_lam = self.lam
        self.lam = self.clampl(self.lam * s)
        self.bad_f = self.o.bad_factor

    def bad_step(self):                                     # lm.h:140-148 (FailedStep == BadStep)
        if self.o.solver != "lm":
            return
        s = self.bad_f
        self.prev_lam = self.lam
        self.lam = self.clampl(self.lam * s)
        self.bad_f *= self.o.bad_factor

    def hessian(self):                                      # lm.h:157-171
        H = [r[:] for r in self.H]
        if self.o.solver == "lm" and self.prev_lam > 0.0:
            s = 1.0 + self.prev_lam
            for i in range(self.n):
                H[i][i] /= s
        return H


def u8(v):
    return v & 0xFF


def optimize(name, x0, opt, max_iters_arg=-1, perturb=lambda c: c, plus=None, ndim=None):
    """OptimizeAcc (optimizer.h:242-327) with Step (:331-539) inlined as step().
    plus(x, dx, sign) -> x (+) sign * dx and ndim = the tangent's dimension: a parameter object on a manifold (traits::params_trait<T>::
    PlusEq, traits.h:149-191; part 3 of the second reading, make_reference_traces_ba.py); default: Euclidean, len(x0)."""
    fn = FUNCS[name]
    n = len(x0) if ndim is None else ndim
    if plus is None:
        plus = lambda xx, dd, sg: [a + sg * b for a, b in zip(xx, dd)]   # noqa: E731  (x + dx and x + (-dx): the same doubles as before)
    x = list(x0)
    S = Solver(opt, n)
    out = dict(errs=[], deltas2=[], successes=[], final_cost=DBL_MAX, final_nres=0, final_rerr_dec=DBL_MAX, stop=kNone,
               num_iters=0, num_failures=0, num_consec=0, min_margin=float("inf"))
    trace = []

    def step():
        it = out["num_iters"]
        o = opt
        solver_failed = True
        dx = None
        max_tries = max(1, o.max_consec_failures) if o.max_consec_failures > 0 else 255
        spins = 0
        while out["num_consec"] <= max_tries:
            spins += 1
            assert spins < 2000, "the reference itself would never leave this loop (uint8_t counter): not a usable fixture"
            if S.build(x, fn, perturb):
                d = S.solve()
                if d is not None:
                    dx = d
                    solver_failed = False
            cost, nres = S.cost, S.nres
            if solver_failed:
                out["num_consec"] = u8(out["num_consec"] + 1)
                out["num_failures"] = u8(out["num_failures"] + 1)
                if nres == 0:
                    out["stop"] = kSkipped
                    return False, None
                if math.isnan(cost) or math.isinf(cost):
                    out["stop"] = kSystemHasNaNOrInf
                    return False, None
                if o.max_consec_failures > 0 and out["num_consec"] >= o.max_consec_failures:
                    if out["final_cost"] < DBL_MAX:
                        out["stop"] = kMaxConsecNoDecr
                    break
                S.bad_step()
            else:
                break
        if solver_failed:
            out["stop"] = kSolverFailed
            return False, None
        err = S.cost
        if math.isnan(err) or math.isinf(err):
            out["stop"] = kSystemHasNaNOrInf
            return False, None
        dx2 = 0.0
        for v in dx:
            dx2 += v * v
        g2 = 0.0
        if o.min_grad_norm2 > 0.0:
            for v in S.g:
                g2 += v * v
        if math.isnan(dx2) or math.isinf(dx2):
            out["stop"] = kSystemHasNaNOrInf
            return False, None
        derr = err - out["final_cost"]
        good = derr < 0.0
        fc = out["final_cost"]
        if it > 0 and derr != 0.0:       # how far the accept / reject decision is from flipping (0.0 exactly: the roll-back
            out["min_margin"] = min(out["min_margin"], abs(derr) / max(abs(fc), 1e-300))   # restored x bit for bit)
        rel = (fc - err) / fc if (fc > f32(1e-7) and fc < DBL_MAX) else 0.0
        out["errs"].append(err)
        out["deltas2"].append(dx2)
        out["successes"].append(good)
        if good or it == 0:
            if it > 0:
                S.good_step(rel if o.use_step_quality_approx else 0.0)
            out["num_consec"] = 0
            out["final_cost"], out["final_nres"] = err, S.nres
            out["final_rerr_dec"] = rel
        else:
            S.bad_step()
            out["num_failures"] = u8(out["num_failures"] + 1)
            out["num_consec"] = u8(out["num_consec"] + 1)
            if o.max_consec_failures > 0 and out["num_consec"] >= o.max_consec_failures:
                out["stop"] = kMaxConsecNoDecr
                return False, None
            if o.max_total_failures > 0 and out["num_failures"] >= o.max_total_failures:
                out["stop"] = kMaxNoDecr
                return False, None
        if o.min_error > 0 and err < o.min_error:
            out["stop"] = kMinError
        elif o.min_rerr_dec > 0 and rel > 0.0 and rel < o.min_rerr_dec:
            out["stop"] = kMinRelError
        elif o.min_step_norm2 > 0 and dx2 < o.min_step_norm2:
            out["stop"] = kMinDeltaNorm
        elif o.min_grad_norm2 > 0 and g2 < o.min_grad_norm2:
            out["stop"] = kMinGradNorm
        return good, dx

    max_iters = opt.max_iters if max_iters_arg < 0 else max_iters_arg
    max_iters += 1
    if opt.check_final_cost:
        max_iters += 1
    last_dx = None
    last_ok = True
    for it in range(max_iters):
        rebuilt = (opt.solver != "lm") or S.rebuild
        nh = len(out["errs"])
        good, dx = step()
        eval_only = False
        if good:
            x = plus(x, dx, 1.0)
            last_dx = dx
            last_ok = True
            if opt.check_final_cost and it + 1 == max_iters:
                eval_only = True
        else:
            if last_dx is not None:
                x = plus(x, last_dx, -1.0)
                last_dx = None
            elif dx is not None:
                x = plus(x, dx, 1.0)
                last_dx = dx
            eval_only = not last_ok
            last_ok = False
        S.rebuild = not eval_only
        out["num_iters"] += 1
        trace.append(dict(rebuilt=bool(rebuilt), recorded=len(out["errs"]) > nh, lam=S.lam, x=list(x),
                          consec=out["num_consec"], fails=out["num_failures"]))
        if out["stop"] != kNone:
            break
    if out["stop"] == kNone and out["num_iters"] >= max_iters:
        out["stop"] = kMaxIters
    out["x"] = x
    out["final_hessian"] = S.hessian()
    out["trace"] = trace
    return out


CASES = [
    # name, start, options, comment
    ("rosenbrock", [-1.2, 1.0], dict(max_iters=200, min_rerr_dec=0, max_consec_failures=20), "tests/optimize_easy.cpp:35-79 as is"),
    ("rosenbrock", [-0.9, 1.3], dict(max_iters=200, min_rerr_dec=0, max_consec_failures=20), "perturbed start"),
    ("rosenbrock", [-1.5, 0.7], dict(max_iters=200, min_rerr_dec=0, max_consec_failures=20), "perturbed start"),
    ("rosenbrock", [0.0, 1.0], dict(max_iters=200, min_rerr_dec=0, max_consec_failures=20),
     "H(0,0) = -398 < 0: every solve fails, the retry loop of Step runs to max_consec_failures => kSolverFailed"),
    ("rosenbrock", [-1.2, 1.0], dict(max_iters=200, min_rerr_dec=0, max_consec_failures=3), "tight failure limit"),
    ("rosenbrock", [-1.2, 1.0], dict(max_iters=200, min_rerr_dec=0, max_consec_failures=0, max_total_failures=9), "total-failure limit"),
    ("rosenbrock", [-1.2, 1.0], dict(max_iters=12, min_rerr_dec=0, max_consec_failures=20, check_final_cost=True), "check_final_cost: last pass eval-only"),
    ("rosenbrock", [-1.2, 1.0], dict(max_iters=200, min_rerr_dec=0, max_consec_failures=20, use_step_quality_approx=True), "step-quality damping (lm.h:127-129)"),
    ("rosenbrock", [-1.2, 1.0], dict(max_iters=200, min_rerr_dec=0, max_consec_failures=20, damping_init=10.0), "heavy initial damping"),
    ("rosenbrock", [-1.2, 1.0], dict(max_iters=60, min_rerr_dec=0, max_consec_failures=20, solver="gn"), "GaussNewton (exact Newton here)"),
    ("plateau", [3.0, 3.0], dict(damping_init=1e-6), "tests/optimize_easy.cpp:88-144 as is"),
    ("plateau", [2.8, 3.3], dict(damping_init=1e-6), "perturbed start"),
    ("plateau", [3.3, 2.9], dict(damping_init=1e-6, max_consec_failures=9), "perturbed start, looser limit"),
    ("plateau", [2.6, 2.6], dict(damping_init=1e-6), "further out on the plateau: indefinite exact Hessian"),
    ("powell", [3.0, -1.0, 0.0, 1.0], dict(max_iters=200, max_consec_failures=0, min_error=1e-30, min_rerr_dec=1e-30, damping_init=1e-1), "tests/optimize_easy.cpp:153-221 as is"),
    ("powell", [2.5, -0.7, 0.3, 1.2], dict(max_iters=200, max_consec_failures=0, min_error=1e-30, min_rerr_dec=1e-30, damping_init=1e-1), "perturbed start"),
    ("powell", [3.0, -1.0, 0.0, 1.0], dict(max_iters=40, damping_init=1e-1), "default stop tests"),
    ("powell", [3.0, -1.0, 0.0, 1.0], dict(max_iters=40, max_consec_failures=0, solver="gn"), "GaussNewton on the singular problem"),
]

RB = dict(max_iters=200, min_rerr_dec=0, max_consec_failures=20)          # the Rosenbrock test's options
PW = dict(max_iters=200, max_consec_failures=0, min_error=1e-30, min_rerr_dec=1e-30, damping_init=1e-1)   # Powell's
BL = dict(max_iters=200, max_consec_failures=0, min_error=1e-30, damping_init=1e-3)                       # Beale's (optimize_hard.cpp:51-55)
HB = dict(max_iters=200, max_consec_failures=0, min_error=1e-30)                                           # Himmelblau's (:90-93)
CASES += [
    # ---- round 4: the option branches one reading had pinned so far
    ("rosenbrock", [-1.0, 1.2], dict(RB, use_step_quality_approx=True), "step-quality damping (lm.h:126-129)"),
    ("rosenbrock", [-1.4, 0.8], dict(RB, use_step_quality_approx=True), "step-quality damping, another start"),
    ("rosenbrock", [-0.7, 1.1], dict(RB, use_step_quality_approx=True, damping_init=1.0), "step-quality damping from heavy damping"),
    ("powell", [3.0, -1.0, 0.0, 1.0], dict(PW, use_step_quality_approx=True), "step-quality damping on Powell"),
    ("plateau", [2.9, 3.2], dict(damping_init=1e-6, use_step_quality_approx=True, max_consec_failures=8), "step-quality damping through rejected steps"),
    ("rosenbrock", [-1.2, 1.0], dict(RB, grad_clipping=50.0), "gradient clipping (base.h:29-38): |g| starts at 216 / 88"),
    ("rosenbrock", [-1.2, 1.0], dict(RB, grad_clipping=5.0), "hard gradient clipping"),
    ("rosenbrock", [-0.9, 1.3], dict(RB, grad_clipping=50.0), "gradient clipping, another start"),
    ("rosenbrock", [-1.5, 0.7], dict(RB, grad_clipping=30.0), "gradient clipping, another start"),
    ("rosenbrock", [-1.1, 0.9], dict(RB, grad_clipping=100.0), "gradient clipping, another start"),
    ("rosenbrock", [-1.3, 1.1], dict(RB, use_step_quality_approx=True), "step-quality damping, another start"),
    ("rosenbrock", [-0.8, 0.9], dict(RB, use_step_quality_approx=True), "step-quality damping, another start"),
    ("powell", [3.0, -1.0, 0.0, 1.0], dict(PW, grad_clipping=20.0), "gradient clipping on Powell (g up to 306)"),
    ("beale", [1.0, 1.0], dict(BL, grad_clipping=2.0), "gradient clipping on a residual-vector cost"),
    ("rosenbrock", [-1.2, 1.0], dict(RB, check_min_H_diag=250.0), "lm.h:82-86: H(1,1) = 200 < 250 at every point => Build fails => kSolverFailed"),
    ("rosenbrock", [-1.2, 1.0], dict(RB, check_min_H_diag=150.0), "min-diagonal check that passes at the start and bites where H(0,0) gets small"),
    # (with max_consec_failures = 0 a Build that keeps failing never leaves Step's retry loop: `num_consec_failures <= 255` holds
    #  for every uint8_t, optimizer.h:356-358 — the fixture gives the loop a limit)
    ("powell", [3.0, -1.0, 0.0, 1.0], dict(PW, check_min_H_diag=5.0, max_consec_failures=6), "min-diagonal check on Powell: H(0,0) = 2 + 120 t4^2 falls under 5 near the solution"),
    ("rosenbrock", [-1.2, 1.0], dict(RB, use_ldlt=False), "gn.h:157-162: -H.inverse() * g, unchecked"),
    ("rosenbrock", [-0.5, 0.5], dict(RB, use_ldlt=False), "unchecked inverse from a start with an indefinite Hessian"),
    ("powell", [3.0, -1.0, 0.0, 1.0], dict(PW, use_ldlt=False), "unchecked inverse on Powell (n = 4)"),
    ("plateau", [2.8, 3.3], dict(damping_init=1e-6, use_ldlt=False, max_consec_failures=8), "unchecked inverse on the plateau"),
    ("himmelblau", [3.5, 2.5], dict(HB, use_ldlt=False), "unchecked inverse, residual-vector cost"),
    ("rosenbrock", [-1.2, 1.0], dict(RB, use_squared_norm=False), "base.h:42: cost = sqrt(c)"),
    ("rosenbrock", [-1.2, 1.0], dict(RB, downscale_by_2=True), "base.h:43: cost *= 0.5"),
    ("rosenbrock", [-1.2, 1.0], dict(RB, use_squared_norm=False, downscale_by_2=True), "both"),
    ("beale", [1.0, 1.0], dict(BL, normalize=True), "base.h:44: cost /= num_residuals (3)"),
    ("himmelblau", [3.5, 2.5], dict(HB, normalize=True, downscale_by_2=True, use_squared_norm=False), "all three flags, 2 residuals"),
    ("beale", [1.0, 1.0], dict(BL), "tests/optimize_hard.cpp:34-63 as is (residual vector, 3 residuals)"),
    ("beale", [2.0, 0.0], dict(BL), "Beale, another start"),
    ("beale", [0.5, 1.5], dict(BL, max_consec_failures=5), "Beale with the default failure limit"),
    ("himmelblau", [3.5, 2.5], dict(HB), "tests/optimize_hard.cpp:72-102 as is (2 residuals)"),
    ("himmelblau", [-3.0, 3.0], dict(HB), "Himmelblau towards another of its four minima"),
    ("himmelblau", [0.0, 0.0], dict(HB, damping_init=1e-2), "Himmelblau from the saddle region"),
    ("beale", [1.0, 1.0], dict(BL, solver="gn"), "GaussNewton on a residual-vector cost"),
]
# the same reading against the fp32 instantiations: short traces, decisions far from a flip
F32_CASES = [
    ("rosenbrock", [-1.2, 1.0], dict(RB, max_iters=10), "float32: the first iterations of the Rosenbrock test"),
    ("rosenbrock", [-0.9, 1.3], dict(RB, max_iters=8, use_step_quality_approx=True), "float32 with step-quality damping"),
    ("powell", [3.0, -1.0, 0.0, 1.0], dict(PW, max_iters=8), "float32 Powell"),
    ("beale", [1.0, 1.0], dict(BL, max_iters=6), "float32 Beale (residual vector)"),
    ("himmelblau", [3.5, 2.5], dict(HB, max_iters=2, normalize=True), "float32 Himmelblau, normalised cost (stopped before the fp32 noise floor decides the stop test)"),
    ("plateau", [2.8, 3.3], dict(damping_init=1e-6, max_iters=10), "float32 plateau: rejected steps"),
]


# seeded random starts around the reference starts (the same options as the reference tests): more routes through the
# rejected-step / eval-only / failed-solve branches than any hand-picked list
_rng = np.random.default_rng(20260929)
for _ in range(14):
    d = _rng.uniform(-0.4, 0.4, 2)
    CASES.append(("rosenbrock", [round(-1.2 + float(d[0]), 3), round(1.0 + float(d[1]), 3)],
                  dict(max_iters=200, min_rerr_dec=0, max_consec_failures=20), "seeded random start"))
for _ in range(8):
    d = _rng.uniform(-0.45, 0.45, 2)
    CASES.append(("plateau", [round(3.0 + float(d[0]), 3), round(3.0 + float(d[1]), 3)], dict(damping_init=1e-6, max_consec_failures=8),
                  "seeded random start"))


def decisions(out):
    return (tuple(out["successes"]), out["stop"], out["num_iters"], out["num_failures"], tuple(t["rebuilt"] for t in out["trace"]))


def main():
    cases = []
    for name, x0, kw, comment, dtype in [c + ("float64",) for c in CASES] + [c + ("float32",) for c in F32_CASES]:
        opt = Options(**kw)
        out = optimize(name, x0, opt)
        robust = True
        peps, need_margin = (1e-13, 1e-9) if dtype == "float64" else (3e-4, 1e-3)
        for eps in (peps, -peps):
            ctr = [0]

            def perturb(c, eps=eps, ctr=ctr):
                ctr[0] += 1
                return c * (1.0 + eps * (1 if ctr[0] % 2 else -1))
            if decisions(optimize(name, x0, opt, perturb=perturb)) != decisions(out):
                robust = False
        # After a rejected step the loop rolls x back (x + dx - dx) and accumulates again THERE: err - final_cost is exactly 0
        # when both roundings cancel (=> "not good", structurally), and a last-bit coin toss when they do not.  Only traces
        # whose every decision is either that exact zero or further than 1e-9 (relative) from flipping are emitted.
        if out["min_margin"] < need_margin:
            robust = False
        nrej = sum(1 for s in out["successes"][1:] if not s)
        neval = sum(1 for t in out["trace"] if not t["rebuilt"])
        print(f"{name:10s} x0={x0} stop={out['stop']:2d} iters={out['num_iters']:3d} fails={out['num_failures']:3d} rejected={nrej:3d} "
              f"eval-only={neval:3d} robust={robust}  x={['%.6g' % v for v in out['x']]}  # {comment}")
        if not robust:
            print("   -> sits on a round-off tie, NOT emitted")
            continue
        o = {k: (v if not isinstance(v, tuple) else list(v)) for k, v in vars(opt).items()}
        cases.append(dict(
            function=name, x0=x0, options=o, comment=comment, dtype=dtype,
            errs=out["errs"], deltas2=out["deltas2"], successes=[int(s) for s in out["successes"]],
            stop_reason=out["stop"], num_iters=out["num_iters"], num_failures=out["num_failures"], num_consec_failures=out["num_consec"],
            final_cost=out["final_cost"], final_rerr_dec=out["final_rerr_dec"], x=out["x"], final_hessian=out["final_hessian"],
            passes=out["trace"]))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_traces.json")
    with open(path, "w") as f:
        json.dump(dict(generator="tests/golden/make_reference_traces.py (independent Python restatement, see its docstring)",
                       cases=cases), f, indent=1)
    print(f"wrote {len(cases)} cases to {path}")


if __name__ == "__main__":
    main()
