import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hk():
    import hikari_jl_amd
    return hikari_jl_amd


@pytest.fixture(scope="session")
def oracle():
    import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def gpu_ctx(hk):
    """hk_ctx on cuda:0; fails loudly (never falls back) when the HIP library is missing."""
    return hk.Context.get(0)
