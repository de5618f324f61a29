import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def anet_ctx():
    import allocnet_amd
    return allocnet_amd.default_context(0)


def pytest_collection_finish(session):
    # The first `import torch` on a fresh GPU box pages the whole wheel in -- one to two minutes normally, more than ten on a slow
    # box -- and inside a test it counts against that test's timeout (pytest.ini: 600 s).  Pay it here, once, before any test runs.
    if any(item.get_closest_marker("gpu") for item in session.items):
        try:
            import torch  # noqa: F401
        except Exception:
            pass
