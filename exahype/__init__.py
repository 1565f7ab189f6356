"""Alias package: lets scripts written against the reference (`from exahype import
KernelBuilder`, `from exahype.printers import CPPPrinter, MLIRPrinter`) run unchanged on
exahype_amd."""
from exahype_amd import KernelBuilder, TypedFunction  # noqa: F401
