import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the oracle (gcc) and, when hipcc is present, the HIP library once per session."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    so = os.path.join(ROOT, "cfd_hemodynamic_amd", "libcfdh.so")
    if os.path.exists("/opt/rocm/bin/hipcc"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "cfd_hemodynamic_amd", "csrc"), "-s", "-j4"])
    assert os.path.exists(so), "libcfdh.so missing and hipcc not available"
    yield


@pytest.fixture()
def oracle_backend(monkeypatch):
    """Registers the oracle-backed test double as the solver plugin "_oracle_double" (host-logic tests without a GPU)."""
    import types
    import oracle_solver
    mod = types.ModuleType("cfd_hemodynamic_amd.solvers._oracle_double")
    mod.Solver = oracle_solver.Solver
    monkeypatch.setitem(sys.modules, "cfd_hemodynamic_amd.solvers._oracle_double", mod)
    return "_oracle_double"
