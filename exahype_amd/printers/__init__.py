from .CodePrinter import CodePrinter
from .CPPPrinter import CPPPrinter
from .HIPPrinter import HIPPrinter, UnrecognisedKernel
from .MLIRPrinter import MLIRPrinter

__all__ = ["CodePrinter", "CPPPrinter", "HIPPrinter", "MLIRPrinter", "UnrecognisedKernel"]
