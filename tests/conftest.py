import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ab: exercises the superseded A/B kernels (depth-only / row-only Winograd, bf16 operand split): "
                                       "runs only against a library built with ESTD_BUILD_AB=1")


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests need a ROCm device AND the built HIP library: skip them (with the reason) on any other box
    instead of failing with RuntimeErrors when `-m "not gpu"` is forgotten."""
    import torch
    from estdepth_amd import _native
    reason = None
    if not torch.cuda.is_available():
        reason = "no ROCm device visible"
    elif not os.path.exists(_native.LIB_PATH):
        reason = "libestd_hip.so not built (python -m estdepth_amd.build)"
    if reason:
        skip = pytest.mark.skip(reason=reason)
        for item in items:
            if "gpu" in item.keywords:
                item.add_marker(skip)
    elif not _native.has_ab():
        skip_ab = pytest.mark.skip(reason="library built without the A/B kernels (ESTD_BUILD_AB=1)")
        for item in items:
            if "ab" in item.keywords:
                item.add_marker(skip_ab)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
