"""ts-asr_amd: MI355X-native hot path of lucadellalib/ts-asr (Conformer-Transducer TS-ASR training step).

The directory name is not a Python identifier; import it with ``importlib.import_module("ts-asr_amd")``
(``tests/conftest.py`` and ``__graft_entry__.py`` do). Everything that computes goes through the C-ABI
of ``lib/libtsasr_hip.so`` (hand-written HIP for gfx950); there is NO CPU fallback: calling an op
without the library or with CPU tensors raises.
"""
from . import _capi  # noqa: F401

__version__ = "0.1.0"
