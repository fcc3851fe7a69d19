import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))  # oracle is test infrastructure: importable from tests only
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _device_status_is_clean(request):
    """After every GPU test: nothing it launched may have reported a failure through the library's device status word (today: the
    one-kernel attention backward's hand-off chain giving up) — a set bit is a hard test failure, never a silent wrong gradient.
    Tests that provoke the failure on purpose read (and clear) the word themselves."""
    yield
    if "gpu" not in request.keywords or not _has_gpu():
        return
    import torch
    from omnibiote_amd import _lib
    torch.cuda.synchronize()
    _lib.check_device_status(request.node.name)
