from exahype_amd.printers import *  # noqa: F401,F403
from exahype_amd.printers import CodePrinter, HIPPrinter, MLIRPrinter  # noqa: F401
