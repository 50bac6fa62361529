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


# The suite runs the library on its DEFAULTS (no environment switch).  Problems with N <= 4 levels (Rabi, cnot2, random
# N = 4) then take the small-problem path (qgd_k_tiny.hip) for calls that return only [grad | scalars]; the GENERAL kernels on
# those problems (the padded 16 x 16 tiles of the chain, inverse and gradient kernels) are a path of their own, so every
# GPU test that works on such a problem runs TWICE: as the library decides ("default"), and with the small-problem path
# switched off per handle -- DeviceProblem.set_small_path(False), i.e. qgd_set_small_path(h, 0) -- ("general").
# tests/test_gpu_tiny.py sets the path per handle itself and is left alone.
_SMALL_WORDS = ("cnot2", "rabi", "gradient_cases", "construct_rand_prob(4", "construct_rand_prob(2", "construct_rand_prob(3", "rand4")


def _works_on_a_small_problem(metafunc):
    import inspect
    try:
        text = inspect.getsource(metafunc.function)
    except (OSError, TypeError):
        text = ""
    if "subprocess" in text:        # (the work happens in child processes, which run on the library's defaults whatever this process does)
        return False
    text += " ".join(str(m.args) for m in metafunc.definition.iter_markers("parametrize"))
    return any(w in text for w in _SMALL_WORDS)


def pytest_generate_tests(metafunc):
    if "_small_path_mode" not in metafunc.fixturenames or metafunc.definition.get_closest_marker("gpu") is None:
        return
    if metafunc.module.__name__.endswith("test_gpu_tiny"):
        return
    if _works_on_a_small_problem(metafunc):
        metafunc.parametrize("_small_path_mode", ["default", "general"], indirect=True)


@pytest.fixture(autouse=True)
def _small_path_mode(request, monkeypatch):
    mode = getattr(request, "param", "default")
    if mode != "general":
        yield mode
        return
    from __graft_entry__ import import_package
    qgd = import_package()
    init = qgd.DeviceProblem.__init__

    def init_general(self, *a, **k):
        init(self, *a, **k)
        self.set_small_path(False)

    qgd.clear_cache()                                   # (handles cached per problem by the functional API were made on the defaults)
    monkeypatch.setattr(qgd.DeviceProblem, "__init__", init_general)
    yield mode
    qgd.clear_cache()


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
