"""exahype_amd -- MI355X-native drop-in for the cell-local kernel stack of
xdslproject/ExaHyPE (operator surface: KernelBuilder / TypedFunction; back-end:
hand-written HIP kernels behind the C-ABI in include/exahype_hip.h)."""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
