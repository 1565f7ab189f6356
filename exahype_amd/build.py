"""Build libexahype_hip.so (gfx950) in-tree with hipcc.

    python -m exahype_amd.build [--force] [--jobs N]

hipcc cross-compiles for gfx950 without a GPU; the resulting .so travels to the
GPU box with the source snapshot (it is git-ignored, not gpurun-ignored).
"""
import argparse
import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "lib", "libexahype_hip.so")
ARCH = "gfx950"

COMMON = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
          "-Wno-pass-failed", "-I", CSRC] + os.environ.get("EXA_EXTRA_FLAGS", "").split()

# the ADER-DG units are scheduled for instruction-level parallelism rather than for occupancy (their big kernels run at the two waves per
# SIMD their register-resident state allows anyway): the register-resident stage A of p = 5 measured 19.17 against 19.69 ms per 64^3 launch.
# Stage B (HBM-bound) loses 13 % under that strategy, so dg_inst.hip is compiled twice: unit A with it, unit B (stage B only) without.
DG_SCHED = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]

# (source, object stem, extra flags)
UNITS = [
    ("capi.cpp", "capi", ["-x", "hip"]),
    ("dg_operators_host.cpp", "dg_operators_host", ["-x", "hip"]),
    # bit-exact with the g++ build of the reference: no FMA contraction in the FV unit
    ("fv_rusanov.hip", "fv_rusanov", ["-ffp-contract=off"]),
    ("limiter.hip", "limiter", []),
] + [
    unit
    for dim, pde in ((3, 1), (2, 1), (2, 0), (3, 2), (2, 2))
    for unit in (("dg_inst.hip", f"dg_{dim}_{pde}", [f"-DEXA_DIM={dim}", f"-DEXA_PDE_ID={pde}", "-DEXA_UNIT_A"] + DG_SCHED),
                 ("dg_inst.hip", f"dgb_{dim}_{pde}", [f"-DEXA_DIM={dim}", f"-DEXA_PDE_ID={pde}", "-DEXA_UNIT_B"]))
]


class build_lock:
    """Inter-process lock for a build directory: N ranks that import the package at once (bench.py --gpus N, torchrun) must not
    compile into the same object files and link the same library together."""

    def __init__(self, directory):
        os.makedirs(directory, exist_ok=True)
        self.path = os.path.join(directory, ".build.lock")

    def __enter__(self):
        import fcntl
        self.f = open(self.path, "w")
        fcntl.flock(self.f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl
        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()


def link_atomically(cmd_without_output, target):
    """Run a link command into a temporary name and rename it into place: a reader never maps a half-written library, and a killed
    hipcc leaves no truncated file behind under the final name."""
    tmp = "%s.tmp.%d" % (target, os.getpid())
    r = subprocess.run(cmd_without_output + ["-o", tmp], capture_output=True, text=True, env=compiler_env())
    if r.returncode != 0:
        if os.path.exists(tmp):
            os.unlink(tmp)
        raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
    os.replace(tmp, target)


def compiler_env():
    """Environment for compiler children.  Under a profiler (rocprofv3 sets LD_PRELOAD / ROCP_TOOL_LIBRARIES / HSA_TOOLS_LIB; with --pmc the preloaded
    library initialises the GPU in EVERY child) hipcc would touch the GPU and then exec clang -- the exec-after-GPU-init hop this pool forbids.  The
    compilers need none of those variables: they are removed, so a build that happens inside a profiled run is an ordinary CPU job."""
    env = dict(os.environ)
    for k in list(env):
        if k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "HSA_TOOLS_REPORT_LOAD_FAILURE", "ROCPROFILER_LIBRARY_CTOR") or k.startswith(("ROCPROF", "ROCP_")):
            del env[k]
    return env


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libexahype_hip.so cannot be built")
    return exe


STAMP = LIB + ".stamp"


def _sources_digest():
    """Content hash of everything the library is built from (file times do not survive the copy to the GPU box)."""
    h = hashlib.sha256()
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in sorted(os.listdir(root)):
            h.update(f.encode())
            h.update(open(os.path.join(root, f), "rb").read())
    h.update(open(os.path.abspath(__file__), "rb").read())
    h.update((ARCH + " " + os.environ.get("EXA_EXTRA_FLAGS", "")).encode())   # (no absolute paths: the tree moves)
    return h.hexdigest()


def up_to_date():
    """True if the built library was made from exactly the sources that are in the tree now."""
    try:
        return os.path.exists(LIB) and open(STAMP).read().strip() == _sources_digest()
    except OSError:
        return False


def _compile(unit, verbose):
    src, stem, extra = unit
    obj = os.path.join(OBJ, stem + ".o")
    cmd = [_hipcc()] + COMMON + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True, env=compiler_env())
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (stem, " ".join(cmd), r.stderr[-4000:]))
    if verbose and r.stderr.strip():
        print(r.stderr[-2000:], file=sys.stderr)
    return obj


def build_variant(tag, flags, jobs=None):
    """Development aid: the library built with extra -D flags into lib/var_<tag>/ (never the one the product loads by
    default; select it with EXA_LIB=<path>).  Used to time kernel variants side by side on one GPU box."""
    global OBJ, COMMON
    lib = os.path.join(HERE, "lib", "var_" + tag, "libexahype_hip.so")
    saved = (OBJ, COMMON)
    try:
        OBJ = os.path.join(HERE, "_build", "var_" + tag)
        COMMON = COMMON + list(flags)
        os.makedirs(OBJ, exist_ok=True)
        os.makedirs(os.path.dirname(lib), exist_ok=True)
        jobs = jobs or min(len(UNITS), os.cpu_count() or 4)
        with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
            objs = list(ex.map(lambda u: _compile(u, False), UNITS))
        r = subprocess.run([_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib] + objs, capture_output=True, text=True, env=compiler_env())
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
    finally:
        OBJ, COMMON = saved
    return lib


def build(force=False, jobs=None, verbose=False):
    """Compile every unit (in parallel) and link the shared library; returns its path."""
    if not force and up_to_date():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    with build_lock(OBJ):
        if not force and up_to_date():                # another process built it while this one waited for the lock
            return LIB
        jobs = jobs or min(len(UNITS), os.cpu_count() or 4)
        with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
            objs = list(ex.map(lambda u: _compile(u, verbose), UNITS))
        link_atomically([_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}"] + objs, LIB)
        tmp = "%s.tmp.%d" % (STAMP, os.getpid())
        with open(tmp, "w") as f:
            f.write(_sources_digest() + "\n")
        os.replace(tmp, STAMP)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    ap.add_argument("-v", "--verbose", action="store_true")
    ap.add_argument("--variant", metavar="TAG", help="development aid: build lib/var_TAG/ with the macros given by --define")
    ap.add_argument("--define", action="append", default=[], metavar="MACRO", help="with --variant: -DMACRO (repeatable)")
    ap.add_argument("--mllvm", action="append", default=[], metavar="OPT", help="with --variant: -mllvm -OPT (repeatable), e.g. amdgpu-sched-strategy=max-ilp")
    a = ap.parse_args()
    if a.variant:
        extra = ["-D" + d for d in a.define]
        for o in a.mllvm:
            extra += ["-mllvm", "-" + o]
        print(build_variant(a.variant, extra, a.jobs))
    else:
        print(build(a.force, a.jobs, a.verbose))
