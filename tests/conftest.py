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


@pytest.fixture(params=[1, 3], ids=["one-launch", "three-launch"])
def auto_form(request, monkeypatch):
    """SELENITE_ARITH_AUTO in both of its forms (selenite_rx_set_auto_launches): every Rx the test opens is pinned to the one-launch form
    (the matrix kernel recomputes its guarded channels itself, where it can) or to the three-launch form (k_hist_exact + the rerun pass).
    Modules opt in with `pytestmark = pytest.mark.usefixtures("auto_form")`."""
    import selenite_rx as sr
    init = sr.Rx.__init__

    def patched(self, *a, **k):
        init(self, *a, **k)
        self.set_auto_launches(request.param)

    monkeypatch.setattr(sr.Rx, "__init__", patched)
    return request.param
