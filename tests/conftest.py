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


# Tests that need no GPU but pin rows of SURVEY.md 8(a) (A11, A21-A23, the BA oracle's second algorithm) ask for the `box` fixture
# (pytest.mark.usefixtures("box")): they are collected twice — `[host]` runs under -m "not gpu" in the build container, `[gpubox]` carries the gpu
# marker so that the driver's -m gpu run on the MI355X box executes (and records) them too.
def pytest_generate_tests(metafunc):
    if "box" in metafunc.fixturenames:
        metafunc.parametrize("box", ["host", pytest.param("gpubox", marks=pytest.mark.gpu)], indirect=True)


@pytest.fixture
def box(request):
    return request.param


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
