import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """The in-tree libraries normally travel with the checkout; if a fresh tree lacks them, build them once
    (hipcc cross-compiles gfx950 without a GPU; the oracle is plain gcc)."""
    from singlerust_amd import _ffi
    if not os.path.exists(_ffi.LIB_PATH):
        from singlerust_amd import build as _b
        _b.build()
    import oracle
    oracle.lib()            # builds oracle/_build/libsrx_oracle.so on first use
    yield


@pytest.fixture(scope="session")
def ctx():
    """One srx_ctx on cuda:0 for the whole GPU session."""
    import singlerust_amd as sr
    c = sr.Context.default()
    yield c
