"""The scripts under examples/ run (tiny sizes) and do what they print."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "examples", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_fv_rusanov_patches_example():
    Q = _load("fv_rusanov_patches").main(64)
    assert np.isfinite(Q).all()


def test_aderdg_sympy_euler_example():
    m0, m1 = _load("aderdg_sympy_euler").main(cells=3, p=3, steps=3)
    assert np.max(np.abs(m1 - m0) / np.abs(m0)) < 1e-12


def test_fv_patch_grid_example():
    m0, m1 = _load("fv_patch_grid").main(8, 0.01)
    assert np.allclose(m0, m1, rtol=1e-12, atol=1e-12)          # periodic grid: every variable is conserved
