import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The patched program's DEFAULT dense-sweep arithmetic is the fast build (certified against the reference program to
# north_star's 1e-12 by tests/test_fast_certificate_gpu.py, which removes this variable).  Every other test compares
# snapshots BIT FOR BIT with the reference, which is what the verification mode is for.
os.environ.setdefault("RAMSES_AMD_STRICT", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): builds oracle/liboracle.so on demand."""
    from oracle import pyoracle
    pyoracle.build_oracle()
    return pyoracle


@pytest.fixture(scope="session")
def gpu_lib():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    import ramses_amd
    return ramses_amd.lib()
