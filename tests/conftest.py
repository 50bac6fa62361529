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


# The suite exercises the GENERAL kernels on small problems too (cnot2, Rabi, random N = 4: the padded 16 x 16 tiles of the chain,
# inverse and gradient kernels).  The small-problem path that the library takes by default for such problems (qgd_k_tiny.hip)
# has its own module, tests/test_gpu_tiny.py, which switches it on per handle.
os.environ.setdefault("QGD_TINY", "0")


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


def pytest_sessionfinish(session, exitstatus):
    """Handles the suite left open (the per-problem cache of evolution.device_problem, handles of failed tests) are closed
    here, while the HIP runtime is certainly alive, instead of by finalizers at interpreter shutdown."""
    import gc
    mod = sys.modules.get("quantumgatedesign.jl_amd.evolution") or sys.modules.get("qgd_amd.evolution")
    for m in list(sys.modules.values()):
        if getattr(m, "__name__", "").endswith(".evolution") and hasattr(m, "clear_cache"):
            mod = m
    if mod is not None:
        try:
            mod.clear_cache()
        except Exception:
            pass
    gc.collect()
    if torch is not None and torch.cuda.is_available():
        torch.cuda.synchronize()
