import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, os.path.join(ROOT, "selenite-lite_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import rxcommon
    return rxcommon.oracle_lib()


@pytest.fixture(scope="session")
def ref():
    import rxcommon
    if not rxcommon.ref_available():
        pytest.skip("oracle/_ref/libcmsis_ref.so not built (needs /root/reference; build container only)")
    return rxcommon.ref_lib()
