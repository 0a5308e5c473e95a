import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.setdefault("SOICP_CACHE", "/tmp/soicp_cache")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def soicp():
    """The product library (built in-tree by `python -m superodom_amd.build`)."""
    from superodom_amd import binding
    if not os.path.exists(binding.LIB_PATH):
        from superodom_amd import build
        build.build()
    binding.load()
    return binding


@pytest.fixture(scope="session")
def gpu_slam_factory(soicp):
    import ctypes
    if not soicp.load().so_icp_device_available():
        pytest.fail("no HIP device: the -m gpu tests need the native library on a GPU box (no CPU fallback exists)")

    def make(**kw):
        return soicp.LidarSlamGpu(**kw)
    return make
