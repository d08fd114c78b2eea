import os, sys
import pytest
try:                      # torch bundles its own HIP runtime: it must be the one the process initialises first,
    import torch          # before libvido_slam_hip.so creates a context (otherwise torch.cuda later finds no GPU)
except ImportError:       # noqa
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def vido():
    import vido_slam_amd
    return vido_slam_amd


@pytest.fixture(scope="session")
def oracle_ops(oracle):
    return oracle.OracleNetOps(oracle)
