"""Named, opaque PDE terms with type tags -- the operator-surface twin of the
reference's `exahype/TypedFunction.py:9-34`.

`TypedFunction(name)` returns a SymPy undefined-function CLASS (so `F(x, y)`
builds an applied call and `str(type(F(x))) == name`, which is how
`KernelBuilder.single` tells call statements apart, reference
`exahype/KernelBuilder.py:147`).  The class carries `return_type` /
`parameter_types` and the accessors `returnType()` / `parameterTypes()`.
SymPy caches undefined-function classes by name, so -- exactly as in the
reference -- the tags of a name are shared by every kernel that uses it.

Extension (default off): `device_term` names the built-in HIP device function
the term dispatches to (see exahype_amd/printers/HIPPrinter.py).
"""
import sympy


def _tagged(func):
    def returnType(returnType=None):
        if returnType is not None:
            func.return_type = returnType
        return func.return_type

    def parameterTypes(parameterTypes=None):
        if parameterTypes is not None:
            func.parameter_types = parameterTypes
        return func.parameter_types

    func.return_type = None
    func.parameter_types = None
    func.returnType = returnType
    func.parameterTypes = parameterTypes
    return func


class TypedFunction:
    """`TypedFunction("Flux")` -> SymPy function class with type tags."""

    def __new__(cls, *args, **options):
        return _tagged(sympy.Function(*args, **options))
