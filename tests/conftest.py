import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Native artefacts: the HIP library (cross-compiled here, prebuilt on the GPU box) and the oracle."""
    import __graft_entry__ as g
    g.build_hip()
    g.build_host()
    g.build_oracle()
    return g


@pytest.fixture(scope="session")
def ctx(built):
    from pixel_aware_gyro_aided_klt_feature_tracker_amd import capi
    c = capi.Context(0)   # raises without a HIP device: there is no fallback
    yield c
    c.close()
