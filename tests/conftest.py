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


# Collection order of the GPU suite: the oracle / golden-vector parity tests of the kernels come FIRST (the RNN-T joint + loss with the
# reference's only known-answer vector at the very front), the whole-model parity tests next, and the tests of the runtime around the
# step (hipGraph replay, recipe loops, process groups, stress repeats) last - so that with `-x` one failure in the runtime cannot hide
# the parity contract (round 2's verdict: 43 parity tests unreached behind one graph-replay failure).
_ORDER = ["test_rnnt_oracle", "test_oracle_golden", "test_host_cpu", "test_rnnt_gpu", "test_blocks_gpu", "test_frontend_block_gpu",
          "test_wgrad_gpu", "test_dataio_gpu", "test_variants_gpu", "test_longform_gpu", "test_model_gpu", "test_hygiene_gpu", "test_recipe_gpu",
          "test_dist_gpu", "test_determinism_gpu"]
_RUNTIME_WORDS = ("hip_graph", "graph_", "determinis", "stress", "replay")


def pytest_collection_modifyitems(session, config, items):
    def key(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        rank = _ORDER.index(mod) if mod in _ORDER else len(_ORDER)
        runtime = any(w in item.name for w in _RUNTIME_WORDS)       # runtime tests of a parity module go behind every parity test
        return (1 if runtime else 0, rank)
    items.sort(key=key)          # stable: the order inside a module is kept


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
