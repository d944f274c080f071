import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure). Built on demand with gcc."""
    import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def canopy_lib():
    """The product library. Built on demand with hipcc (cross-compiles gfx950 without a GPU)."""
    import torch  # noqa: F401  before libcanopy.so: the process then maps ONE HIP/HSA runtime (torch's copies carry the
    #               same SONAMEs as /opt/rocm's), whichever of the two libraries touches the GPU first
    from canopy_amd import build
    build.build()
    from canopy_amd import _capi
    return _capi.lib()


@pytest.fixture(scope="session")
def gpu_ctx(canopy_lib):
    import torch  # first, so that one HIP runtime is shared in-process  # noqa: F401
    import canopy_amd
    ctx = canopy_amd.Context(device=0)
    yield ctx
    ctx.close()
