from exahype_amd.printers import *  # noqa: F401,F403
from exahype_amd.printers import CodePrinter, CPPPrinter, HIPPrinter, MLIRPrinter  # noqa: F401
