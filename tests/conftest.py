"""pytest configuration: registers the ``gpu`` marker and makes the repo root importable.

``-m "not gpu"``: oracle vs golden vectors, host logic, C-ABI symbol checks (runs in the build container).
``-m gpu``      : parity tests proper - every call goes through the C-ABI into the HIP kernels on a MI355X.
"""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package. Its directory is named ``ts-asr_amd`` (not an identifier) -> importlib."""
    return importlib.import_module("ts-asr_amd")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    class G:
        def __getitem__(self, name):
            return np.load(os.path.join(GOLDEN, name + ".npz"))

    return G()
