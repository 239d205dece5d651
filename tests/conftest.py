import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run on the GPU box with -m gpu)')


@pytest.fixture(scope='session')
def gpu():
    """Initialises device 0 through the C ABI; the HIP extension must be present (no CPU fallback)."""
    from dnncancerannotator_amd import device
    assert device.device_count() >= 1, 'no HIP device visible'
    device.init_device(0)
    return device
