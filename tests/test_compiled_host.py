"""A NON-Python caller of the C-ABI (native function boundary): the reference's caller is a compiled C++ `main`
(`Unit test/correctness_test.cpp:102-106,177-183,195-204`), so tests/host/correctness_host.cpp is built by plain g++ against
include/exahype_hip.h, linked with -lexahype_hip and run on the GPU -- sin input, one `exa_fv_time_step_host` call where the reference calls
`time_step(Q1, 1)`, the reference's exact `!=` loop against the golden values of the compiled reference.  The CPU part checks that the header is
valid C99 and that the host compiles and links without any HIP header."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
HOST = os.path.join(ROOT, "tests", "host", "correctness_host.cpp")


def _lib_dir():
    from exahype_amd import build
    build.build()
    return os.path.dirname(build.LIB)


def test_header_is_valid_c99(tmp_path):
    """The boundary is a C ABI: the header must compile as C99 with nothing but the C library beside it, and every prototype must be usable from C."""
    src = tmp_path / "use_header.c"
    src.write_text('#include "exahype_hip.h"\n'
                   'int probe(void) { exa_fv_plan* p = 0; return exa_fv_plan_create(0, EXA_FV_FAITHFUL, 2, 4, 1, 5, 5, 1, EXA_PDE_EULER_REF2D, &p) == EXA_OK\n'
                   '                   && exa_last_error() != 0; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-pedantic-errors", "-Wall", "-Werror", "-I", INC, "-c", str(src), "-o", str(tmp_path / "use_header.o")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def _build_host(tmp_path):
    exe = str(tmp_path / "correctness_host")
    lib = _lib_dir()
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", INC, HOST, "-o", exe, "-L", lib, "-lexahype_hip", "-Wl,-rpath," + lib,
                        "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_compiled_host_builds_and_links_without_hip_headers(tmp_path):
    """g++ (not hipcc), no -I /opt/rocm: the host sees only include/exahype_hip.h and resolves every symbol it uses from libexahype_hip.so."""
    assert shutil.which("g++")
    exe = _build_host(tmp_path)
    out = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True).stdout
    assert "exa_fv_time_step_host" in out and "exa_fv_plan_create" in out and "hip" not in out.lower().replace("exahype_hip", "")


@pytest.mark.gpu
def test_compiled_host_runs_the_reference_protocol_bit_exact(tmp_path):
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "fv_ref2d_sin.json")))
    gold = tmp_path / "golden.txt"
    with open(gold, "w") as f:
        for i, v in zip(g["valid_modified_idx"], g["valid_modified_val"]):
            f.write("v %d %s\n" % (i, float(v).hex()))
        for i in g["passthrough_idx"]:
            f.write("p %d\n" % i)
    exe = _build_host(tmp_path)
    r = subprocess.run([exe, str(gold)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "correct: %d entries" % (len(g["valid_modified_idx"]) + len(g["passthrough_idx"])) in r.stdout
