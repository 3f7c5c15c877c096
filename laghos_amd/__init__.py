"""laghos_amd — MI355X-native partial-assembly hot path of CEED/Laghos.

The product is the HIP library `liblaghos_hip.so` (C ABI: include/laghos_hip.h)
plus the C++ host layer in laghos_amd/host/ that mirrors the reference's
LagrangianHydroOperator / ForcePAOperator / MassPAOperator / QUpdate classes.
The Python modules here are ctypes plumbing for tests and bench.py.
"""
from . import _lib  # noqa: F401


def load():
    """Load the HIP extension (raises if liblaghos_hip.so is missing)."""
    return _lib.load()
