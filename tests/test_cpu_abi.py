"""CPU tests: the C-ABI library loads, exports every symbol include/tinyopt_amd.h declares, and the
POD contracts agree between header, Python mirror and the reference's defaults.  No compute calls."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(built):
    from tinyopt_amd import _capi
    lib = _capi.load()
    hdr = open(os.path.join(ROOT, "include", "tinyopt_amd.h")).read()
    declared = set(re.findall(r"\b(toa_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/tinyopt_amd.h but not exported"
    assert declared == set(_capi.PROTOTYPES), "python prototypes out of sync with the header"


def test_options_defaults_match_reference(built):
    """options.h:43-148 defaults and benchmarks/options.h:10-27."""
    from tinyopt_amd import _capi
    from tinyopt_amd.api import Options
    lib = _capi.load()
    d = _capi.ToaOptions()
    lib.toa_options_default(C.byref(d))
    py = Options().to_pod()
    for f, _ in _capi.ToaOptions._fields_:
        assert getattr(d, f) == getattr(py, f), f
    assert (d.max_iters, d.max_consec_failures, d.max_total_failures) == (50, 5, 0)
    assert d.min_error == pytest.approx(1e-12) and d.min_rerr_dec == pytest.approx(1e-10)
    assert d.min_step_norm2 == pytest.approx(1e-14) and d.min_grad_norm2 == pytest.approx(1e-18)
    assert d.damping_init == pytest.approx(1e-4) and d.good_factor == pytest.approx(1 / 3) and d.bad_factor == 2.0
    assert (d.damping_min, d.damping_max) == (pytest.approx(1e-9), pytest.approx(1e9))
    assert d.use_ldlt == 1 and d.H_is_full == 1 and d.save_last == 1 and d.use_squared_norm == 1
    b = _capi.ToaOptions()
    lib.toa_options_benchmark(C.byref(b))
    pyb = Options.benchmark().to_pod()
    for f, _ in _capi.ToaOptions._fields_:
        assert getattr(b, f) == getattr(pyb, f), f
    assert (b.max_iters, b.max_consec_failures, b.save_last) == (10, 3, 0) and b.min_error == 0.0


def test_dense_row_layout(built):
    from tinyopt_amd import _capi
    lib = _capi.load()
    nb, thin, rs, m4 = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    nbytes = C.c_size_t()
    cases = {  # (dtype, n, m) -> (NB, THIN, RS, m4)
        (0, 50, 2000): (3, 3, 51, 2000),   # 48 MFMA columns + thin tail {a48, a49, b}: exactly m(n+1) floats
        (1, 12, 500): (1, 0, 13, 500), (1, 6, 1000): (1, 0, 7, 1000),
        (0, 15, 5): (1, 0, 16, 8), (0, 16, 5): (1, 1, 17, 8), (0, 19, 5): (1, 4, 20, 8), (0, 20, 5): (2, 0, 22, 8),
        (0, 31, 5): (2, 0, 32, 8), (0, 32, 5): (2, 1, 33, 8), (1, 48, 9): (3, 1, 49, 12), (1, 51, 9): (3, 4, 52, 12),
        (1, 52, 9): (4, 0, 56, 12), (1, 63, 9): (4, 0, 64, 12), (0, 1, 1): (1, 0, 2, 4),
    }
    for (dt, n, m), want in cases.items():
        assert lib.toa_dense_row_layout(dt, n, m, C.byref(nb), C.byref(thin), C.byref(rs), C.byref(m4), C.byref(nbytes)) == 0
        assert (nb.value, thin.value, rs.value, m4.value) == want, (dt, n, m)
        assert nbytes.value == rs.value * m4.value * (4 if dt == 0 else 8)
    assert lib.toa_dense_row_layout(0, 64, 10, None, None, None, None, None) != 0   # n > 63: invalid argument
    assert b"n must be" in lib.toa_last_error()
    assert lib.toa_dense_row_layout(3, 5, 10, None, None, None, None, None) != 0    # bad dtype


def test_stop_reason_values_match_reference():
    """include/tinyopt/stop_reasons.h:14-43."""
    from tinyopt_amd.api import StopReason
    assert [int(s) for s in StopReason] == list(range(-4, 10))
    assert StopReason.kSolverFailed == -3 and StopReason.kMinDeltaNorm == 3 and StopReason.kMaxConsecNoDecr == 7


def test_no_product_import_of_oracle():
    """The product path must never route through the oracle."""
    pkg = os.path.join(ROOT, "tinyopt_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in src and "lm_oracle" not in src, f"{f} references the oracle"
    assert "oracle" not in open(os.path.join(ROOT, "include", "tinyopt_amd.h")).read().replace("oracle/synth.hpp", "")


def test_missing_library_fails_loudly(monkeypatch, built):
    from tinyopt_amd import _capi
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "LIB_PATH", "/nonexistent/libtinyopt_amd.so")
    with pytest.raises(ImportError):
        _capi.load()
