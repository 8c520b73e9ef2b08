import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with `-m gpu` through gpurun)')


@pytest.fixture(scope='session')
def tm():
    """The native library (C-ABI).  GPU tests fail loudly if it is missing -- no fallback."""
    from lmdeploy_amd import _ffi
    return _ffi.load()


@pytest.fixture(scope='session')
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail('GPU test selected but no GPU is visible')
    torch.cuda.set_device(0)
    return torch.device('cuda:0')


@pytest.fixture(autouse=True)
def _release_device_buffers(request):
    yield
    if request.node.get_closest_marker('gpu') is not None:
        from tests import gpu_helpers
        gpu_helpers.release_all()
