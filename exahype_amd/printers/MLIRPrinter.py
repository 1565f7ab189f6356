"""The reference's MLIRPrinter (`exahype/printers/MLIRPrinter.py:45-143`) lowers the builder
state to xDSL/MLIR text.  That path is out of scope for the MI355X build (north-star: "not codegen
through the xDSL/MLIR path"; it also has no downstream in the reference).  The name
is kept so `from exahype.printers import CPPPrinter, MLIRPrinter` keeps importing."""
from .CodePrinter import CodePrinter


class MLIRPrinter(CodePrinter):
    def __init__(self, kernel, function_name: str = "time_step"):
        raise NotImplementedError("MLIRPrinter: the xDSL/MLIR lowering is out of scope of exahype_amd; "
                                  "use HIPPrinter (MI355X kernels) or CPPPrinter (reference-compatible C++ text)")

    def loop(self, expr, direction, below, struct_inclusion):  # pragma: no cover
        raise NotImplementedError
