"""exahype_amd -- MI355X-native drop-in for the cell-local kernel stack of
xdslproject/ExaHyPE.  Operator surface: KernelBuilder / TypedFunction (same as
the reference's `exahype/__init__.py:1-3`, minus the xDSL lowering); back-end:
hand-written HIP kernels behind the C-ABI of include/exahype_hip.h, reached
through `exahype_amd.printers.HIPPrinter` or `exahype_amd.solvers`."""
from .KernelBuilder import KernelBuilder
from .TypedFunction import TypedFunction

__all__ = ["KernelBuilder", "TypedFunction"]
