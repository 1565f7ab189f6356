from .CodePrinter import CodePrinter
from .HIPPrinter import HIPPrinter, UnrecognisedKernel
from .MLIRPrinter import MLIRPrinter

__all__ = ["CodePrinter", "HIPPrinter", "MLIRPrinter", "UnrecognisedKernel"]
