import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    """CPU oracle (test infrastructure). Built on demand with gcc."""
    from oracle import binding

    return binding.load_oracle()


@pytest.fixture(scope="session")
def engine_lib():
    """The HIP library. No fallback: a missing library is a test failure, not a skip."""
    from limbo_amd import _capi

    return _capi.load_engine()
