import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # A test that hangs inside a C call (hipStreamSynchronize behind a kernel that never ends, an RCCL collective) would otherwise eat the
    # whole run without a word -- it happened once in round 5 (a loopback multi-device test on one box; it never reproduced). With
    # pytest-timeout's THREAD method the run ends with the Python stacks of every thread instead. The longest test (config 5 at full
    # size against the oracle on every host core) takes ~40 s on the GPU box; the CPU suite's longest ~25 s.
    if config.pluginmanager.hasplugin("timeout") and not config.getoption("timeout", None):
        config.option.timeout = 600
        config.option.timeout_method = "thread"


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_lib
    return oracle_lib.OracleBackend()


@pytest.fixture(scope="session")
def native_ctx():
    """The HIP library on cuda:0. GPU tests must run the native path: no fallback."""
    from evergreen_amd import native
    ctx = native.Context(0)
    yield ctx
    ctx.close()
