import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _torch_runtime_first(request):
    """GPU sessions that also use torch.distributed (the RCCL test) need torch's bundled HIP runtime initialised before
    libhm_amd.so opens the device (see historymatching_amd/_lib.py:_torch_first): import torch up front so that the
    library's first context creation lets it go first.  No-op on CPU-only boxes."""
    markexpr = request.config.getoption("-m") or ""
    if "not gpu" not in markexpr:  # GPU tests may run in this session
        try:
            import torch

            if torch.cuda.device_count() > 0:
                torch.cuda.init()
        except Exception:
            pass
    yield
