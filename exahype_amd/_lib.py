"""ctypes binding of libexahype_hip.so (include/exahype_hip.h).

There is no CPU fallback: if the library is missing it is built with hipcc, and
if that fails -- or a call reports an error -- an exception is raised.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libexahype_hip.so")

PDE_EULER_REF2D, PDE_EULER, PDE_ADVECTION = 0, 1, 2
FV_FAITHFUL, FV_RUSANOV = 0, 1

# every symbol include/exahype_hip.h declares: (restype, argtypes)
_vp, _dp, _lp = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_long)
SIGNATURES = {
    "exa_version": (C.c_int, []),
    "exa_last_error": (C.c_char_p, []),
    "exa_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "exa_register_pde": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "exa_pde_flags": (C.c_int, [C.c_int]),
    "exa_pde_eval_device": (C.c_int, [C.c_int, C.c_int, C.c_long, C.c_int, _vp, _vp, _vp, _vp]),
    "exa_pde_eval_device_at": (C.c_int, [C.c_int, C.c_int, C.c_long, C.c_int, _vp, _vp, C.c_double, _vp, _vp, _vp]),
    "exa_fv_plan_create": (C.c_int, [C.c_int] * 7 + [C.c_long, C.c_int, C.POINTER(_vp)]),
    "exa_fv_plan_destroy": (C.c_int, [_vp]),
    "exa_fv_q_count": (C.c_long, [_vp]),
    "exa_fv_time_step_host": (C.c_int, [_vp, _vp, C.c_double, C.c_double]),
    "exa_fv_time_step_device": (C.c_int, [_vp, _vp, C.c_double, C.c_double, _vp]),
    "exa_fv_time_step_device_oop": (C.c_int, [_vp, _vp, _vp, _vp, C.c_double, C.c_double, C.c_double, _vp]),
    "exa_fv_time_step_device_at": (C.c_int, [_vp, _vp, _vp, C.c_double, C.c_double, C.c_double, _vp]),
    "exa_fv_qout_count": (C.c_long, [_vp]),
    "exa_fv_grid_step_device": (C.c_int, [_vp, _vp, _vp, _lp, _vp, _vp, C.c_double, C.c_double, C.c_double, _vp, _vp]),
    "exa_fv_max_eigenvalue": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_double, C.c_double, _vp, _vp]),
    "exa_fv_time_step_device_masked": (C.c_int, [_vp, _vp, _vp, C.c_double, C.c_double, _vp]),
    "exa_fv_time_step_device_masked_at": (C.c_int, [_vp, _vp, _vp, _vp, C.c_double, C.c_double, C.c_double, _vp]),
    "exa_dg_plan_create": (C.c_int, [C.c_int] * 6 + [_lp, C.POINTER(_vp)]),
    "exa_dg_plan_set_stage_a": (C.c_int, [_vp, C.c_int]),
    "exa_dg_stage_a_kernel": (C.c_char_p, [_vp]),
    "exa_dg_plan_set_stage_a_reserve": (C.c_int, [_vp, C.c_int]),
    "exa_dg_plan_destroy": (C.c_int, [_vp]),
    "exa_dg_dof_count": (C.c_long, [_vp]),
    "exa_dg_trace_count": (C.c_long, [_vp]),
    "exa_dg_face_count": (C.c_long, [_vp, C.c_int]),
    "exa_dg_operators": (C.c_int, [_vp] + [_vp] * 7),
    "exa_dg_work": (C.c_int, [_vp, _dp, _dp, _dp, _dp]),
    "exa_dg_predictor_volume": (C.c_int, [_vp, _vp, _vp, C.c_double, _dp, _vp]),
    "exa_dg_predictor_volume_box": (C.c_int, [_vp, _vp, _vp, _lp, _lp, C.c_double, _dp, _vp]),
    "exa_dg_riemann_corrector": (C.c_int, [_vp, _vp, _vp, C.POINTER(_vp), _lp, _lp, C.c_double, _dp, _vp]),
    "exa_dg_riemann_corrector_cfl": (C.c_int, [_vp, _vp, _vp, C.POINTER(_vp), _lp, _lp, C.c_double, _dp, _vp, _vp]),
    "exa_dg_plan_set_origin_time": (C.c_int, [_vp, _dp, C.c_double]),
    "exa_dg_has_corrector_predictor": (C.c_int, [_vp]),
    "exa_dg_corrector_predictor": (C.c_int, [_vp, _vp, _vp, _vp, C.POINTER(_vp), _lp, _lp, C.c_double, C.c_double, _dp, _vp, _vp]),
    "exa_dg_pack_face": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, _vp]),
    "exa_lim_operators": (C.c_int, [_vp, _vp, _vp]),
    "exa_lim_patch_count": (C.c_long, [_vp]),
    "exa_dg_project_patches": (C.c_int, [_vp, _vp, _vp, C.c_long, _vp, _vp]),
    "exa_dg_project_patches_ghost": (C.c_int, [_vp, _vp, _vp, C.c_long, _vp, C.POINTER(_vp), _vp]),
    "exa_lim_face_layer_count": (C.c_long, [_vp, C.c_int]),
    "exa_lim_face_layers": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "exa_dg_reconstruct_patches": (C.c_int, [_vp, _vp, _vp, C.c_long, _vp, _vp]),
    "exa_dg_max_eigenvalue": (C.c_int, [_vp, _vp, _vp, _vp]),
    "exa_dg_has_fused_step": (C.c_int, [_vp]),
    "exa_dg_step_fused": (C.c_int, [_vp, _vp, _vp, C.c_double, _dp, _vp]),
    "exa_dg_step_periodic": (C.c_int, [_vp, _vp, _vp, C.c_double, _dp, C.c_int, _vp]),
    "exa_dg_step_host": (C.c_int, [_vp, _vp, C.c_double, _dp, C.c_int]),
}


class ExaHypeHipError(RuntimeError):
    pass


_lib = None


def load(build_if_missing=True):
    """Load (building first if needed) the HIP library; raises if it cannot be had."""
    global _lib
    if _lib is not None:
        return _lib
    # The kernels are launched on torch's streams and on torch-owned HBM, so both must sit on ONE HIP
    # runtime: torch ships its own libamdhip64 and it has to be the copy already mapped when our
    # library's DT_NEEDED entry is resolved (two runtimes in one process: the second sees no device).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    from . import build as _build
    override = os.environ.get("EXA_LIB")                 # development aid: a variant build (exahype_amd.build --variant)
    if override:
        if not os.path.exists(override):
            raise ExaHypeHipError(f"EXA_LIB={override} does not exist")
        global LIB_PATH
        LIB_PATH = override
    elif build_if_missing:
        _build.build()                   # no-op when the library matches the sources (content stamp), else rebuilds
    elif not os.path.exists(LIB_PATH):
        raise ExaHypeHipError(f"{LIB_PATH} is missing (run `python -m exahype_amd.build`); there is no CPU fallback")
    elif not _build.up_to_date():
        raise ExaHypeHipError(f"{LIB_PATH} is older than its sources (run `python -m exahype_amd.build`)")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise ExaHypeHipError(f"cannot load {LIB_PATH}: {e}; there is no CPU fallback") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise ExaHypeHipError("libexahype_hip error %d: %s" % (rc, load().exa_last_error().decode()))


def device_count():
    n = C.c_int(0)
    rc = load().exa_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def register_pde(library_path):
    """Register a user PDE side library (exahype_amd/pde_codegen.py); returns its pde id (>= 100)."""
    pid = C.c_int(-1)
    check(load().exa_register_pde(library_path.encode(), C.byref(pid)))
    return pid.value


def darr(vals):
    return (C.c_double * len(vals))(*[float(v) for v in vals])


def larr(vals):
    return (C.c_long * len(vals))(*[int(v) for v in vals])
