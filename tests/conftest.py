import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref built from /root/reference (build container only)")


@pytest.fixture(scope="session")
def oracle():
    from _oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    from _oracle import Reference, build_ref, have_reference
    if not have_reference():
        try:
            build_ref()
        except Exception:
            pass
    if not have_reference():
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    return Reference()
