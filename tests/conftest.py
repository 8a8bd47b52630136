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


class _Knobs:
    """Tuning knobs of the device context for the duration of one test (hk_ctx_set_option — the library reads HK_* from the environment
    only once, in hk_ctx_create).  setenv / delenv mirror pytest's monkeypatch; everything is restored at teardown."""

    def __init__(self, ctx):
        self.ctx, self.saved = ctx, {}

    def setenv(self, name, value):
        self.saved.setdefault(name, self.ctx.get_option(name))
        self.ctx.set_option(name, value)

    def delenv(self, name, raising=False):
        self.saved.setdefault(name, self.ctx.get_option(name))
        self.ctx.set_option(name, None)

    def restore(self):
        for name, value in self.saved.items():
            self.ctx.set_option(name, value)
        self.saved.clear()


@pytest.fixture
def knobs(gpu_ctx):
    k = _Knobs(gpu_ctx)
    yield k
    k.restore()
