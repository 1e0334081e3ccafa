import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vit-search_amd"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_sessionfinish(session, exitstatus):
    """Leave the GPU quiet before the interpreter tears down: captured graphs, side streams and workspaces are released while the HIP
    runtime is still alive (an abort inside the runtime's own exit handlers would turn a green run into a non-zero exit code)."""
    try:
        import gc
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            gc.collect()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
    except Exception:       # noqa: BLE001
        pass
