#!/usr/bin/env python3
"""Generate the committed golden fixtures from the reference ITSELF.

Runs only in the build container (needs /root/reference); the fixtures it
writes (JSON / text: inputs and expected outputs, never reference source) are
what travels.  Two sources:

 1. The reference's Python operator surface, imported from /root/reference by
    file path (the xdsl-free slice TypedFunction -> KernelBuilder -> CodePrinter
    -> CPPPrinter; `import exahype` itself needs xdsl, which is absent --
    SURVEY.md F7).  The bodies of examples/Batched_stateless.py:9-35 and
    examples/kernel-generator.py:6-45 are exec'd from the reference files (not
    copied here) and the builder state + CPPPrinter(...).code are dumped.
 2. The reference's native kernel, compiled from its own sources by
    oracle/Makefile into oracle/_ref/libexa_ref.so: time_step on the
    sin(3.141 i/N) input of correctness_test.cpp:102-106 (+ validity mask,
    SURVEY.md F6) and Flux/maxEigenvalue on seeded admissible random states.

Usage: python tests/golden/make_golden.py
"""
import importlib.util
import io
import json
import os
import sys
import types
import contextlib

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


def load_reference_slice():
    pkg = types.ModuleType("exahype"); pkg.__path__ = [os.path.join(REF, "exahype")]
    sub = types.ModuleType("exahype.printers"); sub.__path__ = [os.path.join(REF, "exahype", "printers")]
    sys.modules["exahype"] = pkg
    sys.modules["exahype.printers"] = sub
    for name, rel in (("exahype.TypedFunction", "exahype/TypedFunction.py"),
                      ("exahype.KernelBuilder", "exahype/KernelBuilder.py"),
                      ("exahype.printers.CodePrinter", "exahype/printers/CodePrinter.py"),
                      ("exahype.printers.CPPPrinter", "exahype/printers/CPPPrinter.py")):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
    pkg.KernelBuilder = sys.modules["exahype.KernelBuilder"].KernelBuilder
    pkg.TypedFunction = sys.modules["exahype.TypedFunction"].TypedFunction
    sub.CPPPrinter = sys.modules["exahype.printers.CPPPrinter"].CPPPrinter
    return pkg, sub


def builder_state(k):
    return dict(
        dim=k.dim, patch_size=k.patch_size, halo_size=k.halo_size, n_patches=k.n_patches, n_real=k.n_real,
        n_aux=k.n_aux, indexes=[str(i) for i in k.indexes],
        inputs=list(k.inputs), input_types=list(k.input_types), items=list(k.items),
        directional_items=list(k.directional_items),
        directional_consts={a: list(b) for a, b in k.directional_consts.items()},
        functions=list(k.functions), item_struct=dict(k.item_struct), parents=dict(k.parents),
        literals=list(k.literals), all_items=sorted(k.all_items.keys()),
        LHS=[str(x) for x in k.LHS], RHS=[str(x) for x in k.RHS], directions=list(k.directions),
        struct_inclusion=list(k.struct_inclusion),
        function_types={f: dict(return_type=str(k.all_items[f].returnType()),
                                parameter_types=[str(p) for p in (k.all_items[f].parameterTypes() or [])])
                        for f in k.functions},
    )


def run_example_body(path, first, last, pkg, sub):
    """exec lines [first,last] (1-based, inclusive) of a reference example with the slice injected."""
    lines = open(path).read().split("\n")[first - 1:last]
    src = "\n".join(l for l in lines if not l.startswith(("from exahype", "import sys")))
    env = dict(KernelBuilder=pkg.KernelBuilder, CPPPrinter=sub.CPPPrinter)
    exec("from sympy import IndexedBase\nfrom sympy.codegen.ast import integer, real, none\n" + src, env)
    return env["kernel"]


def main():
    pkg, sub = load_reference_slice()
    out = {}
    # NB: sympy caches Function(name) classes globally, so the type tags of a TypedFunction are shared
    # by every kernel that names it: dump each kernel's state before the next kernel is built.
    for name, (path, first, last) in (("batched_stateless", ("examples/Batched_stateless.py", 3, 35)),
                                      ("kernel_generator", ("examples/kernel-generator.py", 3, 45))):
        k = run_example_body(os.path.join(REF, path), first, last, pkg, sub)
        st = builder_state(k)
        with open(os.path.join(HERE, f"builder_state_{name}.json"), "w") as f:
            json.dump(st, f, indent=1, sort_keys=True)
        code = sub.CPPPrinter(k).code
        with open(os.path.join(HERE, f"cppprinter_{name}.txt"), "w") as f:
            f.write(code)
        out[name] = len(code)

    # 3-D, multi-patch variant of the Batched_stateless body (dim=3,P=15,H=1,n_patches=2)
    from sympy.codegen.ast import integer, real, none
    k3 = pkg.KernelBuilder(dim=3, patch_size=15, halo_size=1, n_real=5, n_aux=0, n_patches=2)
    Q = k3.item('Q'); Qc = k3.item('Q_copy'); tf = k3.directional_item('tmp_flux'); te = k3.directional_item('tmp_eigen', struct=False)
    dt = k3.const('dt'); normal = k3.directional_const('normal', [0, 1, 2])
    Flux = k3.function('Flux', parameter_types=[Q, real, Q], return_type=integer)
    Eigen = k3.function('maxEigenvalue', parameter_types=[Q, real], return_type=real)
    Max = k3.function('max', parameter_types=[Q, Q], return_type=none)
    k3.single(Qc[0], Q[0]); k3.directional(Flux(Qc[0], normal, tf[0])); k3.directional(te[0], Eigen(Qc[0], normal))
    k3.directional(Qc[0], Qc[0] + 0.5 * (tf[-1] - tf[1]))
    left = -Max(te[-1], te[0]) * (Q[0] - Q[-1]); right = -Max(te[1], te[0]) * (Q[0] - Q[1])
    k3.directional(Qc[0], Qc[0] + 0.5 * dt * (left - right), struct=True); k3.single(Q[0], Qc[0])
    with open(os.path.join(HERE, "builder_state_3d_p15.json"), "w") as f:
        json.dump(builder_state(k3), f, indent=1, sort_keys=True)
    with open(os.path.join(HERE, "cppprinter_3d_p15.txt"), "w") as f:
        f.write(sub.CPPPrinter(k3).code)

    # the first input used as an OBJECT (a const with an in_type that is the parent of the items): the reference's parse() post-pass rewrites its members
    km = pkg.KernelBuilder(dim=2, patch_size=4, halo_size=1, n_real=4, n_aux=0)
    Data = km.const('patchData', in_type='::exahype2::CellData&')
    Qo = km.item('QOut', parent=Data); Qi = km.item('QIn', parent=Data); dtm = km.const('dt', parent=Data)
    km.single(Qi[0], Qo[0]); km.single(Qo[0], Qi[0] * dtm)
    with open(os.path.join(HERE, "cppprinter_member_input.txt"), "w") as f:
        f.write(sub.CPPPrinter(km).code)

    # error behaviour
    errs = {}
    for kw in (dict(dim=1, patch_size=4, halo_size=1), dict(dim=2, patch_size=0, halo_size=1), dict(dim=3, patch_size=4, halo_size=-1)):
        try:
            pkg.KernelBuilder(n_real=1, n_aux=0, **kw)
        except Exception as e:  # noqa
            errs[json.dumps(kw, sort_keys=True)] = [type(e).__name__, str(e)]
    try:
        pkg.KernelBuilder(2, 4, 1, 1, 0).directional_const('n', [0])
    except Exception as e:  # noqa
        errs["directional_const_len"] = [type(e).__name__, str(e)]
    with open(os.path.join(HERE, "builder_errors.json"), "w") as f:
        json.dump(errs, f, indent=1, sort_keys=True)

    # ---- native reference vectors ------------------------------------------------
    import oracle
    R = oracle.ref()
    assert R is not None, "oracle/_ref did not build"
    N = 360
    Q = np.sin(3.141 * np.arange(N) / N)           # correctness_test.cpp:102-106
    outs = []
    for perturb in ("0", "191", "63"):             # heap-fill experiment, run out of process
        import subprocess
        code = ("import sys; sys.path.insert(0, %r); import numpy as np, oracle; R = oracle.ref();"
                "Q = np.sin(3.141*np.arange(360)/360); R.ref_time_step(Q, 1.0); print(' '.join(repr(float(x)) for x in Q))" % ROOT)
        env = dict(os.environ, MALLOC_PERTURB_=perturb)
        txt = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout
        outs.append(np.array([float(t) for t in txt.split()]))
    stable = np.all([np.array_equal(outs[0], o, equal_nan=False) for o in outs[1:]], axis=0) if False else \
        np.logical_and(outs[0] == outs[1], outs[0] == outs[2])
    changed = outs[0] != Q
    mask = np.nonzero(np.logical_and(stable, changed))[0]
    passthrough = np.nonzero(np.logical_and(stable, ~changed))[0]
    fx = dict(
        source="Unit test/test.cpp + Functions.cpp compiled by oracle/Makefile (g++ -O2 -ffp-contract=off)",
        dim=2, patch_size=4, halo_size=1, n_real=5, n_aux=5, dt=1.0, input="sin(3.141*i/360), i=0..359",
        valid_modified_idx=[int(i) for i in mask], valid_modified_val=[float(outs[0][i]) for i in mask],
        n_passthrough=int(len(passthrough)), passthrough_idx=[int(i) for i in passthrough],
        undefined_idx=[int(i) for i in np.nonzero(~stable)[0]],
    )
    with open(os.path.join(HERE, "fv_ref2d_sin.json"), "w") as f:
        json.dump(fx, f, indent=1)

    # seeded admissible random states through the reference's Flux / maxEigenvalue
    rng = np.random.default_rng(20241008)
    n = 64
    rho = rng.uniform(0.5, 2.0, n); u = rng.uniform(-1, 1, n); v = rng.uniform(-1, 1, n); p = rng.uniform(0.5, 2.0, n)
    Qs = np.zeros((n, 10)); Qs[:, 0] = rho; Qs[:, 1] = rho * u; Qs[:, 2] = rho * v
    Qs[:, 3] = p / 0.4 + 0.5 * rho * (u * u + v * v); Qs[:, 4:] = rng.uniform(-1, 1, (n, 6))
    fl = np.zeros((n, 2, 4)); ev = np.zeros((n, 2))
    for i in range(n):
        for d in range(2):
            F = np.zeros(5)
            R.ref_Flux(np.ascontiguousarray(Qs[i]), d, F)
            fl[i, d] = F[:4]
            ev[i, d] = R.ref_maxEigenvalue(np.ascontiguousarray(Qs[i]), d)
    with open(os.path.join(HERE, "euler_terms_ref2d.json"), "w") as f:
        json.dump(dict(source="Unit test/Functions.cpp:9-62 via oracle/_ref", Q=Qs.tolist(), flux=fl.tolist(), maxeig=ev.tolist()), f)

    # a random-state patch batch through time_step, valid mask only (cells {2,3}^2, vars 0..3)
    npatch = 8
    Qb = np.zeros((npatch, 6, 6, 10))
    rho = rng.uniform(0.5, 2.0, (npatch, 6, 6)); u = rng.uniform(-1, 1, (npatch, 6, 6)); v = rng.uniform(-1, 1, (npatch, 6, 6)); p = rng.uniform(0.5, 2, (npatch, 6, 6))
    Qb[..., 0] = rho; Qb[..., 1] = rho * u; Qb[..., 2] = rho * v; Qb[..., 3] = p / 0.4 + 0.5 * rho * (u * u + v * v)
    Qb[..., 4:] = rng.uniform(-1, 1, (npatch, 6, 6, 6))
    Qo = Qb.copy()
    for pidx in range(npatch):
        tmp = np.ascontiguousarray(Qo[pidx]).ravel()
        R.ref_time_step(tmp, 0.37)
        Qo[pidx] = tmp.reshape(6, 6, 10)
    with open(os.path.join(HERE, "fv_ref2d_random.json"), "w") as f:
        json.dump(dict(source="oracle/_ref time_step, dt=0.37; compare cells (i,j) in {2,3}^2, vars 0..3 only (SURVEY F6)",
                       dt=0.37, Q_in=Qb.tolist(), Q_out_valid=Qo[:, 2:4, 2:4, 0:4].tolist()), f)
    print("golden written:", sorted(os.listdir(HERE)))
    print("mask", list(mask), "n_passthrough", len(passthrough), "n_undefined", int((~stable).sum()))


if __name__ == "__main__":
    main()
