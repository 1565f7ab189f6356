"""ORACLE package -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's hot path (oracle/exa_oracle.c, plain C) plus
the compiled reference itself where it builds (oracle/_ref, from
`/root/reference/Unit test/{test,Functions}.cpp`).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the
product (exahype_amd/) never does and fails loudly without its HIP library.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PDE_EULER_REF2D, PDE_EULER, PDE_ADVECTION = 0, 1, 2

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_lp = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile liborc.so (and _ref when the reference tree is present)."""
    so = os.path.join(HERE, "liborc.so")
    src = os.path.join(HERE, "exa_oracle.c")
    need = force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src)
    ref_so = os.path.join(HERE, "_ref", "libexa_ref.so")
    if need or (os.path.isdir("/root/reference/Unit test") and not os.path.exists(ref_so)):
        subprocess.run(["make", "-C", HERE, "-B"] if force else ["make", "-C", HERE], check=True,
                       stdout=subprocess.DEVNULL)
    return so


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        so = os.environ.get("EXA_ORACLE_LIB") or os.path.join(HERE, "liborc.so")   # override: the sanitizer build of the tests
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.orc_pde_flux.argtypes = [C.c_int, C.c_int, _dp, C.c_int, _dp]
        L.orc_pde_maxeig.argtypes = [C.c_int, _dp, C.c_int]
        L.orc_pde_maxeig.restype = C.c_double
        L.orc_fv_rusanov_faithful.argtypes = [_dp, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_long, C.c_int]
        L.orc_fv_rusanov.argtypes = [_dp, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_long, C.c_int]
        L.orc_aderdg_predictor.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, C.c_double, _dp, _dp]
        L.orc_aderdg_stage_a.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.c_long, _dp, C.c_double, _dp, _dp, _dp]
        L.orc_aderdg_stage_b.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _dp, _lp, _dp, _dp, C.c_double, _dp, _dp]
        L.orc_aderdg_step.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _lp, _dp, C.c_double, _dp]
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_aderdg_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _lp, _dp, C.c_double, _dp, C.c_int]
        _lib = L
    return _lib


def ref():
    """The compiled reference (or None when oracle/_ref did not travel / cannot be built)."""
    global _ref
    if _ref is None:
        so = os.path.join(HERE, "_ref", "libexa_ref.so")
        if not os.path.exists(so):
            if os.path.isdir("/root/reference/Unit test"):
                build()
            if not os.path.exists(so):
                return None
        R = C.CDLL(so)
        R.ref_time_step.argtypes = [_dp, C.c_double]
        R.ref_Flux.argtypes = [_dp, C.c_int, _dp]
        R.ref_maxEigenvalue.argtypes = [_dp, C.c_int]
        R.ref_maxEigenvalue.restype = C.c_double
        R.ref_time_step_batched.argtypes = [_dp, C.c_double, C.c_long, C.c_long]
        _ref = R
    return _ref


def pack_ops(ops):
    """[w D Kxi phiL phiR iK1 F0] as one contiguous fp64 array (exa_oracle.c:ctx_from_ops)."""
    return np.ascontiguousarray(np.concatenate([
        ops['w'], ops['D'].ravel(), ops['Kxi'].ravel(), ops['phiL'], ops['phiR'], ops['iK1'].ravel(), ops['F0']]))


def fv_faithful(Q, dt, dim, P, H, n_real, n_aux, n_patches=1, pde=PDE_EULER_REF2D):
    Q = np.ascontiguousarray(Q, dtype=np.float64).copy()
    lib().orc_fv_rusanov_faithful(Q.ravel(), dt, dim, P, H, n_real, n_aux, n_patches, pde)
    return Q


def fv_corrected(Q, dt, h, dim, P, H, n_real, n_aux, n_patches=1, pde=PDE_EULER):
    Q = np.ascontiguousarray(Q, dtype=np.float64).copy()
    lib().orc_fv_rusanov(Q.ravel(), dt, h, dim, P, H, n_real, n_aux, n_patches, pde)
    return Q


def aderdg_stage_a(u, dt, dx, ops, dim, N, m, pde, n_it):
    """u[ncells, N^dim, m] -> (ustar, trace[dim,2,ncells,2,m,Nf])."""
    u = np.ascontiguousarray(u, dtype=np.float64)
    ncells = u.size // (N ** dim * m)
    Nf = N ** (dim - 1)
    us = np.empty_like(u)
    tr = np.zeros((dim, 2, ncells, 2, m, Nf))
    lib().orc_aderdg_stage_a(dim, N, m, pde, n_it, pack_ops(ops), ncells, u.ravel(), dt,
                             np.asarray(dx, dtype=np.float64), us.ravel(), tr.ravel())
    return us, tr


def aderdg_stage_b(us, tr, dt, dx, ops, dim, N, m, pde, nc):
    un = np.empty_like(us)
    lib().orc_aderdg_stage_b(dim, N, m, pde, pack_ops(ops), np.asarray(nc, dtype=np.int64), us.ravel(), tr.ravel(), dt,
                             np.asarray(dx, dtype=np.float64), un.ravel())
    return un


def aderdg_step(u, dt, dx, ops, dim, N, m, pde, n_it, nc):
    u = np.ascontiguousarray(u, dtype=np.float64).copy()
    lib().orc_aderdg_step(dim, N, m, pde, n_it, pack_ops(ops), np.asarray(nc, dtype=np.int64), u.ravel(), dt,
                          np.asarray(dx, dtype=np.float64))
    return u


def aderdg_run(u, dt, dx, ops, dim, N, m, pde, n_it, nc, n_steps):
    """n_steps steps IN PLACE on the C-contiguous float64 array u (work arrays allocated once; timing aid)."""
    lib().orc_aderdg_run(dim, N, m, pde, n_it, pack_ops(ops), np.asarray(nc, dtype=np.int64), u.ravel(), dt,
                         np.asarray(dx, dtype=np.float64), n_steps)
    return u


def aderdg_predictor(u_cell, dt, dx, ops, dim, N, m, pde, n_it):
    q = np.empty((N,) + tuple(u_cell.shape))
    lib().orc_aderdg_predictor(dim, N, m, pde, n_it, pack_ops(ops), np.ascontiguousarray(u_cell).ravel(), dt,
                               np.asarray(dx, dtype=np.float64), q.ravel())
    return q
