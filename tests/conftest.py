import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def oracle():
    import _oracle
    _oracle.build_oracle(with_ref=True)
    # agents are independent inside a step: the oracle's Environment::step may use the host's cores (same bits, the GPU
    # suite spends most of its time in the oracle's brute-force sweep)
    _oracle.lib().oracle_set_threads(min(8, os.cpu_count() or 1))
    return _oracle


@pytest.fixture(scope="session")
def ok():
    """The product package with libokenv.so built; GPU tests fail (not skip) if it cannot load."""
    import openkitchen_amd
    openkitchen_amd.build()
    openkitchen_amd.capi.load(build_if_missing=False)
    return openkitchen_amd


@pytest.fixture(scope="session")
def gpu(ok):
    if not _gpu_available():
        pytest.fail("GPU test selected but no GPU is visible")
    return ok
