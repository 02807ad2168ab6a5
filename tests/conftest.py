import os
import sys

# the release library reads only its documented environment knobs (pseg_env_knobs); the tests keep steering kernel plans with
# monkeypatch.setenv and the Python binding turns those variables into the plan switches of the engines it creates
os.environ["PSEG_PLAN_FROM_ENV"] = "1"

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "page-segmentation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


def _have_gpu():
    try:
        import pseg_amd
        return pseg_amd.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests fail loudly (never skip silently) when the HIP extension or device is missing."""
    # torch bundles its own HIP runtime: when both live in one process torch has to initialise
    # first (bench.py does the same), otherwise torch reports "No HIP GPUs are available".
    try:
        import torch
        torch.cuda.is_available()
    except ImportError:
        pass
    import pseg_amd
    n = pseg_amd.device_count()
    assert n > 0, "no HIP device visible: -m gpu tests need the MI355X box"
    return pseg_amd
