import os
import sys

import pytest

try:  # torch first: its bundled HIP runtime must be the one libqgd_hip.so binds to when both live in a process
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def qgd():
    from __graft_entry__ import import_package
    return import_package()


@pytest.fixture(scope="session")
def orc():
    from __graft_entry__ import import_oracle
    o = import_oracle()
    o.lib()
    return o
