import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun); everything else runs on CPU")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _engine_precision_is_restored(request):
    """GPU tests share one engine: whatever arithmetic a test switches to, the next test starts from the default again."""
    if "gpu" not in request.keywords or not _have_gpu():
        yield
        return
    from tests.gpu_common import engine
    eng = engine()
    eng.set_precision("bf16x6")
    yield
    eng.set_precision("bf16x6")


@pytest.fixture(params=["bf16x6", "fp32"])
def each_precision(request):
    """Parity tests that must hold on both fp32-grade arithmetics of the contraction kernels."""
    from tests.gpu_common import engine
    engine().set_precision(request.param)
    return request.param
