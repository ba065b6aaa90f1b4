import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU-marked tests skip (instead of erroring at import / handle creation) on a machine without a GPU."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (no GPU visible)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def built():
    """Make sure the native artefacts exist (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def oracle(built):
    from oracle import pyoracle
    pyoracle.load()
    return pyoracle


@pytest.fixture(scope="session")
def ta(built):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import tinyopt_amd
    return tinyopt_amd
