import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as _orc
    _orc.build()
    return _orc


@pytest.fixture(autouse=True)
def _azd_environment_does_not_leak():
    """every test starts from, and leaves behind, the session's own AZD_* environment (the engine reads tuning knobs and test hooks
    from it: a knob one test sets must not steer the next -- a leaked AZD_POOL_READY_LANES once made an unrelated pool-step test
    fall back to the asynchronous step now and then)"""
    import os
    before = {k: v for k, v in os.environ.items() if k.startswith("AZD_")}
    yield
    for k in [k for k in os.environ if k.startswith("AZD_") and k not in before]:
        del os.environ[k]
    os.environ.update(before)
