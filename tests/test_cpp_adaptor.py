"""The header-only C++ adaptor: compiles with plain g++ against the C-ABI (CPU check), runs on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_header_adaptor.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "_test_header_adaptor")


def _compile():
    libdir = os.path.join(ROOT, "tinyopt_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), SRC, "-o", EXE,
                    "-L", libdir, "-ltinyopt_amd", "-pthread", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)


def test_header_adaptor_compiles_with_plain_gxx(built):
    """No hipcc, no HIP headers, no Eigen: a host TU only needs include/ and -ltinyopt_amd."""
    _compile()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_header_adaptor_runs(built):
    _compile()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok" in r.stdout
