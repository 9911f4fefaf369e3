import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import orc
    return orc.load()


@pytest.fixture(scope="session")
def gpu_lib():
    """The product library; GPU tests fail loudly (never skip) if it is missing."""
    from hydra_amd import capi
    return capi.lib()
