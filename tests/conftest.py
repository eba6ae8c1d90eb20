import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


# the tests run the superseded prefilter kernels (k_prefilter_cf, k_prefilter_cw<0 / 1>) as independent implementations of the candidate
# set: they live in a test-only library that burst_amd.capi loads in front of libburst_hip.so when asked to (a flag of THIS process: child
# processes -- bench.py, burst_hip -- run the product library alone)
from burst_amd import capi as _capi  # noqa: E402
_capi.LOAD_LEGACY_PREFILTERS = True


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    if os.path.exists("/dev/kfd") and os.path.isdir("/dev/dri"):
        return True
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
