"""Printer base class -- same surface as the reference's
`exahype/printers/CodePrinter.py:46-71`: a printer wraps a KernelBuilder, exposes
`.kernel()`, `.functionName()`, `.file()`, `.here()`, the `.code` text, and an
abstract `.loop(expr, direction, below, struct_inclusion)`."""
from __future__ import annotations

from abc import ABC, abstractmethod


class CodePrinter(ABC):
    def __init__(self: CodePrinter, kernel, function_name: str):
        self._kernel = kernel
        self._functionName = function_name
        self.code = ''

    def kernel(self: CodePrinter, kernel=None):
        if kernel is not None:
            self._kernel = kernel
        return self._kernel

    def functionName(self: CodePrinter, function_name: str = None) -> str:
        if function_name is not None:
            self._functionName = function_name
        return self._functionName

    def file(self: CodePrinter, file_name: str, header_file_name: str = None):
        with open(file_name, 'w') as out:
            out.write(self.code)

    def here(self: CodePrinter):
        print(self.code)

    @abstractmethod
    def loop(self, expr, direction: int, below: int, struct_inclusion: int):
        ...
