"""The CPU restatement under AddressSanitizer + UndefinedBehaviorSanitizer (the reference's own
kernel reads uninitialised heap, F6; the restatement must not).  Host-side only -- GPU sanitizers are not available."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys
sys.path.insert(0, %(root)r)
import numpy as np
import oracle
from oracle.dg_operators import operators
from tests.util import euler_dg_state, euler_patches, euler_ref2d_patches
Q = euler_ref2d_patches(3, 6, 10, seed=1)
oracle.fv_faithful(Q, 0.3, 2, 4, 1, 5, 5, 3, oracle.PDE_EULER_REF2D)
for dim, P, H in ((2, 5, 1), (3, 4, 2)):
    Q = euler_patches(2, dim, P + 2 * H, 5, seed=dim)
    oracle.fv_faithful(Q, 0.05, dim, P, H, 5, 0, 2, oracle.PDE_EULER)
    oracle.fv_corrected(Q, 0.01, 0.1, dim, P, H, 5, 0, 2, oracle.PDE_EULER)
for dim, N, nc in ((2, 3, (3, 2)), (3, 2, (2, 2, 3)), (3, 4, (1, 2, 1))):
    u = euler_dg_state(tuple(nc) + (N,) * dim, seed=N)
    dx = [1.0 / c for c in nc]
    for n_it in (N, 0):
        oracle.aderdg_step(u.reshape(-1), 1e-3, dx, operators(N), dim, N, 5, oracle.PDE_EULER, n_it, nc)
print("sanitizer run clean")
'''


def test_oracle_clean_under_asan_ubsan(tmp_path):
    so = os.path.join(ROOT, "oracle", "liborc_san.so")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), so], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build not available: " + r.stderr[-300:])
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan.so not found")
    script = tmp_path / "run.py"
    script.write_text(SCRIPT % dict(root=ROOT))
    env = dict(os.environ, EXA_ORACLE_LIB=so, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="2")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "sanitizer run clean" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
