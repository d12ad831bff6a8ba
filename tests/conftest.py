import os
import sys

import pytest

# torch (used by some GPU tests to generate data in HBM) brings its own copy of the HIP runtime: it has to be
# loaded before libgpumap.so pulls in /opt/rocm's, or torch finds no GPU later in the same process.
try:
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def workdir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("gpumap"))
