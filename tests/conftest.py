import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import __graft_entry__ as entry  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def amd():
    """The ctypes binding over libg16hip.so; building is __graft_entry__.build()'s job."""
    mod = entry.load_package()
    mod.load()
    return mod


def golden_path(name):
    return os.path.join(ROOT, "tests", "golden", name)
