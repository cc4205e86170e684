import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The oracle (test infrastructure) and the product library must exist before any test runs."""
    oracle_so = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
    if not os.path.exists(oracle_so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    lib = os.path.join(ROOT, "dust_amd", "libdust_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()
    yield


def has_gpu():
    from dust_amd import _lib
    return _lib.load().dust_hip_device_count() > 0


# the long tests go last (a failure in a short one reports in seconds), the process-isolated stress slices at the very end
_LATE = ("test_gpu_fullsize", "test_gpu_config4k", "test_configs", "test_gpu_stress")


def pytest_collection_modifyitems(config, items):
    def rank(item):
        name = os.path.basename(str(item.fspath))
        for k, late in enumerate(_LATE):
            if name.startswith(late):
                return k + 1
        return 0
    items.sort(key=rank)   # stable: the order inside a rank stays the collection order
