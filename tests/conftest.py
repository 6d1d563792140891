import os
import sys

import pytest

# torch first: its wheel bundles its own ROCm runtime (libamdhip64.so.7 of ROCm 7.0); librt_amd.so links the image's
# (ROCm 7.2) under the same soname. Whichever is loaded first serves both; torch cannot find the GPU through the image's.
# (pytest tests/ imports torch during collection anyway - test_tiles_gloo.py - this makes single-file runs behave the same.)
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    """The parity oracle (test infrastructure). Built on demand from oracle/rt_oracle.c."""
    from oracle import oracle as O
    O.build()
    return O
