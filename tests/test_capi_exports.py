"""The C-ABI library loads on a machine without a GPU, exports every function include/exahype_hip.h
declares, and fails loudly (no CPU fallback) when asked to compute without a device."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "exahype_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(exa_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_all_exported_and_bound():
    from exahype_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
        assert n in _lib.SIGNATURES, "ctypes binding missing for %s" % n
    assert sorted(_lib.SIGNATURES) == names
    assert lib.exa_version() >= 100


def test_no_silent_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check is for GPU-less machines")
    from exahype_amd import _lib, solvers
    with pytest.raises(_lib.ExaHypeHipError) as e:
        solvers.FVRusanovKernel(2, 4, 1, 5, 5)
    assert "no CPU fallback" in str(e.value) or "no HIP device" in str(e.value)
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.exa_dg_plan_create(0, 3, 6, 5, 1, -1, _lib.larr([2, 2, 2]), C.byref(h)) == -3        # EXA_ERR_NO_DEVICE
    assert lib.exa_fv_plan_create(0, 0, 1, 4, 1, 5, 5, 1, 0, C.byref(h)) == -1                        # bad dim -> EXA_ERR_INVALID
    assert b"check viability of inputs" in lib.exa_last_error()


def test_product_never_imports_oracle():
    """The product path must not route through the oracle (or the reference)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "exahype_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "liborc" not in src and "/root/reference" not in src, f
