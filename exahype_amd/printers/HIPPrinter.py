"""HIPPrinter -- the MI355X back-end behind the reference's printer seam.

Where the reference's `CPPPrinter` (`exahype/printers/CPPPrinter.py:45-354`) turns
the KernelBuilder state into C++ text (one loop nest per statement, heap
temporaries, out-of-line PDE calls), this printer RECOGNISES the scheme the
statement list describes and dispatches it, through the ctypes C-ABI
(include/exahype_hip.h), to a hand-written fused HIP kernel.  `.code` is the
launch plan, for inspection; `.compile()` binds the library; `.run(Q, dt)` is
the analogue of calling the generated `time_step(Q, dt)` (`Unit test/test.h:3`).

Recognised schemes
  "fv-rusanov-faithful"  the statement list of `examples/Batched_stateless.py:25-35`
                         (any dim / patch / halo / variable counts / names):
                         detected structurally, executed with the exact semantics
                         of the generated `Unit test/test.cpp`.
  "fv-rusanov"           same declarations, corrected Rusanov update (dt/h, all
                         n_real variables) -- explicit hint only.
  "aderdg"               ADER-DG step on cells = patches (patch_size = order+1,
                         halo_size = 0, n_patches = number of cells) -- explicit
                         hint only; not expressible in the reference's surface.
  "statements"           (automatic, for a statement list that is none of the above) lowered statement by statement
                         as the reference's CPPPrinter does for every builder state -- one generated kernel per
                         statement with the reference's ranges, strides and expression text (printers/lowering.py);
                         refused, with the reason, if it calls opaque functions, reaches outside an array over its
                         loop range (the reference's own examples do) or carries a dependence through a loop.
Anything else raises: there is no CPU path.

The user's opaque PDE terms (resolved at link time to `Unit test/Functions.cpp` in the
reference) must be named: `pde=` selects a built-in device term set or a SympyPDE,
SymPy bodies on the builder's functions generate one; only functions named exactly
`Flux`, `maxEigenvalue`, `max` (the reference's example) default to that example's
Functions.cpp terms.  The choice is recorded in `.code`.
"""
from __future__ import annotations

import numpy as np

from ..KernelBuilder import KernelBuilder
from .CodePrinter import CodePrinter

PDE_IDS = {"euler_ref2d": 0, "euler": 1, "advection": 2}
_PDE_DOC = {0: "EulerRef2D  (Unit test/Functions.cpp:9-62 as compiled by the reference, 2-D branch)",
            1: "Euler       (rho, m0, m1, m2, E), gamma = 1.4",
            2: "Advection   (a = (1, 0.5, -0.75))"}


class UnrecognisedKernel(NotImplementedError):
    pass


def _rusanov_template(k: KernelBuilder):
    """The Batched_stateless statement list rebuilt with the user's names; None if the declarations cannot match."""
    if len(k.items) != 2 or len(k.directional_items) != 2 or len(k.inputs) != 1 or len(k.functions) != 3 \
            or len(k.directional_consts) != 1:
        return None
    flux_items = [n for n in k.directional_items if k.item_struct[n] == 1]
    eig_items = [n for n in k.directional_items if k.item_struct[n] == 0]
    if len(flux_items) != 1 or len(eig_items) != 1:
        return None
    (cname, cvals), = k.directional_consts.items()
    if list(cvals) != list(range(k.dim)):
        return None
    t = KernelBuilder(k.dim, k.patch_size, k.halo_size, k.n_real, k.n_aux, k.n_patches)
    Q = t.item(k.items[0])
    Qc = t.item(k.items[1])
    # declare the directional items in the user's order (it fixes dict order, not semantics)
    di = {n: t.directional_item(n, struct=(k.item_struct[n] == 1)) for n in k.directional_items}
    fl, ei = di[flux_items[0]], di[eig_items[0]]
    dt = t.const(k.inputs[0])
    normal = t.directional_const(cname, list(cvals))
    F, E, M = (t.all_items.__setitem__(n, k.all_items[n]) or k.all_items[n] for n in k.functions)
    t.functions = list(k.functions)
    t.single(Qc[0], Q[0])
    t.directional(F(Qc[0], normal, fl[0]))
    t.directional(ei[0], E(Qc[0], normal))
    t.directional(Qc[0], Qc[0] + 0.5 * (fl[-1] - fl[1]))
    left = -M(ei[-1], ei[0]) * (Q[0] - Q[-1])
    right = -M(ei[1], ei[0]) * (Q[0] - Q[1])
    t.directional(Qc[0], Qc[0] + 0.5 * dt * (left - right), struct=True)
    t.single(Q[0], Qc[0])
    return t


def _celldata_template(k: KernelBuilder):
    """The statement list of `examples/kernel-generator.py:36-45` (the `exahype2::CellData` flavour: input item with halo,
    halo-less output item, both members of a CellData object; PDE terms that take the volume centre, the volume size, t and dt)
    rebuilt with the user's names; None if the declarations cannot match."""
    if len(k.items) != 3 or len(k.directional_items) != 2 or len(k.inputs) != 1 or len(k.functions) != 5 \
            or len(k.directional_consts) != 1 or len(k.input_types) != 2 or "CellData" not in str(k.input_types[0]):
        return None
    data, qout, qin = k.items
    if k.parents.get(qout) != data or k.parents.get(qin) != data:
        return None
    members = [n for n, par in k.parents.items() if par == data and n not in k.items]
    if len(members) != 4:
        return None
    flux_items = [n for n in k.directional_items if k.item_struct[n] == 1]
    eig_items = [n for n in k.directional_items if k.item_struct[n] == 0]
    if len(flux_items) != 1 or len(eig_items) != 1:
        return None
    (cname, cvals), = k.directional_consts.items()
    if list(cvals) != list(range(k.dim)):
        return None
    t = KernelBuilder(k.dim, k.patch_size, k.halo_size, k.n_real, k.n_aux, k.n_patches)
    Data = t.item(data, in_type=k.input_types[0])
    t.const(k.inputs[0], in_type=k.input_types[1])
    Q = t.item(qout, parent=Data)
    Qc = t.item(qin, parent=Data)
    di = {n: t.directional_item(n, struct=(k.item_struct[n] == 1)) for n in k.directional_items}
    fl, ei = di[flux_items[0]], di[eig_items[0]]
    # members in the order of the example: dt, t, (the directional constant), cellCentre, cellSize
    dt = t.const(members[0], parent=Data)
    tt = t.const(members[1], parent=Data)
    normal = t.directional_const(cname, tuple(cvals))
    centre = t.const(members[2], parent=Data)
    size = t.const(members[3], parent=Data)
    fn = []
    for n in k.functions:
        par = k.parents.get(n)
        fn.append(t.function(n, parent=par) if par is not None else t.function(n))
    F, E, M, C, S = fn
    i, j, kk = t.all_items["i"], t.all_items["j"], t.all_items["k"]
    patch_size = t.all_items["patch_size"]
    t.single(Qc[0], Q[0])
    t.directional(F(Qc[0], C(centre, size, patch_size, {i, j} if k.dim == 2 else {i, j, kk}), S(size, patch_size), tt, dt, normal, fl[0]))
    t.directional(ei[0], F(Qc[0], C(centre, size, patch_size), S(size, patch_size), tt, dt, normal))
    t.directional(Qc[0], Qc[0] + 0.5 * (fl[-1] - fl[1]))
    left = -M(ei[-1], ei[0]) * (Q[0] - Q[-1])
    right = -M(ei[1], ei[0]) * (Q[0] - Q[1])
    t.directional(Qc[0], Qc[0] + 0.5 * dt * (left - right), struct=True)
    t.single(Q[0], Qc[0])
    return t


def _same_statements(a: KernelBuilder, b: KernelBuilder):
    return ([str(x) for x in a.LHS] == [str(x) for x in b.LHS] and [str(x) for x in a.RHS] == [str(x) for x in b.RHS]
            and list(a.directions) == list(b.directions) and list(a.struct_inclusion) == list(b.struct_inclusion))


class HIPPrinter(CodePrinter):
    def __init__(self: HIPPrinter, kernel: KernelBuilder, function_name: str = "time_step", scheme: str = None,
                 pde: str = None, grid=None, n_picard: int = -1, device: int = 0):
        super().__init__(kernel, function_name=function_name)
        k = kernel
        self.device = device
        self._impl = None
        self.lowering = None
        self.cell_data = False
        if scheme is None:
            c = None
            try:
                c = _celldata_template(k)
            except Exception:                                   # declarations that resemble the flavour but do not rebuild: not this scheme
                c = None
            if c is not None and _same_statements(k, c):
                # The CellData flavour (examples/kernel-generator.py).  As written the statement list is not an executable scheme (it reads the
                # OUTPUT item before anything wrote it and takes the wave speed from the flux function); it stands for ExaHyPE 2's batched
                # Rusanov patch update on a CellData object (`Unit test/correctness_test.cpp:142-155`): QOut = Rusanov update of QIn, out of
                # place, PDE terms that see the volume centre and the time.  That scheme is what is dispatched; the terms must be named.
                if pde is None and self._pde_from_bodies(k) is None:
                    raise UnrecognisedKernel(
                        "the statement list is the exahype2::CellData flavour of the Rusanov patch update (examples/kernel-generator.py); its "
                        "PDE terms %s are methods of a Peano solver object -- say which device term set they are with pde= (%s, or a "
                        "pde_codegen.SympyPDE, whose terms may depend on the volume centre and on t) or give the functions SymPy bodies"
                        % (list(k.functions[:2]), ", ".join(sorted(PDE_IDS))))
                self.cell_data = True
                scheme = "fv-rusanov"
        if scheme is None:
            t = _rusanov_template(k)
            if t is None or not _same_statements(k, t):
                # not a scheme with a fused hand-written kernel: lower it statement by statement, as the reference's printer
                # does for every builder state (printers/lowering.py) -- or say why that cannot be done on a GPU
                from .lowering import LoweringRefused, StatementLowering
                try:
                    self.lowering = StatementLowering(k, function_name)
                except LoweringRefused as why:
                    raise UnrecognisedKernel(
                        "the statement list is not a scheme with a hand-written HIP kernel (recognised: the FV Rusanov "
                        "patch update of examples/Batched_stateless.py; by hint: scheme='fv-rusanov' | 'aderdg') and cannot be "
                        "lowered statement by statement: %s.  There is no CPU fallback" % why) from None
                self.scheme = "statements"
                self.pde, self.user_pde, self.pde_origin, self.n_picard = None, None, "none (no PDE term in the statements)", n_picard
                self.code = ("// exahype_amd HIP lowering of `%s` (MI355X / gfx950): %d statements -> %d kernels, launched in order\n"
                             % (function_name, len(k.LHS), len(self.lowering.statements))) + self.lowering.source()
                return
            scheme = "fv-rusanov-faithful"
        if scheme not in ("fv-rusanov-faithful", "fv-rusanov", "aderdg"):
            raise ValueError("unknown scheme %r" % scheme)
        self.scheme = scheme
        # The PDE terms are opaque symbols in the reference (resolved at link time to the user's C++, Functions.h:2-4); a
        # HIP kernel needs a device term set.  It is never guessed: explicit pde=, SymPy bodies, or -- for functions named
        # exactly as in the reference's example (Flux / maxEigenvalue / max) -- that example's Functions.cpp.
        self.pde_origin = "pde= argument"
        if scheme == "fv-rusanov-faithful" and len(k.functions) == 3 and k.functions[2] != "max":
            raise UnrecognisedKernel("the faithful FV Rusanov kernel implements the reference's `max` (Functions.cpp:64-66) as "
                                     "third term; got %r" % k.functions[2])
        if pde is None:
            pde = self._pde_from_bodies(k)
            if pde is not None:
                self.pde_origin = "SymPy bodies of the builder's functions"
        if pde is None:
            if list(k.functions) == ["Flux", "maxEigenvalue", "max"] and scheme != "aderdg":
                pde = "euler_ref2d" if k.dim == 2 else "euler"
                self.pde_origin = "default: functions are named Flux / maxEigenvalue / max as in the reference's example, whose terms are Unit test/Functions.cpp"
            else:
                raise UnrecognisedKernel(
                    "the PDE terms %s are opaque symbols: say which device term set they are with pde= (%s, or a "
                    "pde_codegen.SympyPDE) or give the functions SymPy bodies -- a term set is never guessed"
                    % (list(k.functions) or "(none declared)", ", ".join(sorted(PDE_IDS))))
        self.user_pde = None
        if hasattr(pde, "register") and hasattr(pde, "source"):      # a SympyPDE: compiled for the device on compile()
            self.user_pde = pde
            self.pde = -1
        elif pde not in PDE_IDS:
            raise ValueError("unknown PDE term set %r (have %s, or a pde_codegen.SympyPDE)" % (pde, sorted(PDE_IDS)))
        else:
            self.pde = PDE_IDS[pde]
        self.n_picard = n_picard
        if scheme == "aderdg":
            if k.halo_size != 0 or k.n_aux != 0:
                raise ValueError("scheme='aderdg' maps a DG cell onto a patch with halo_size = 0 and n_aux = 0")
            grid = tuple(grid) if grid is not None else None
            if grid is None:
                e = round(k.n_patches ** (1.0 / k.dim))
                grid = (e,) * k.dim
            if int(np.prod(grid)) != k.n_patches or len(grid) != k.dim:
                raise ValueError("grid %s does not hold n_patches = %d cells" % (grid, k.n_patches))
            self.grid = grid
        elif k.halo_size < 1:
            raise ValueError("the Rusanov stencil reads one halo layer: halo_size must be >= 1")
        self.code = self._plan_text()

    # -- inspection --------------------------------------------------------------------------------
    def loop(self, expr, direction: int, below: int, struct_inclusion: int):
        """Index ranges [lo, hi) per loop level the statement spans in the generated reference kernel
        (`Unit test/test.cpp`): patch, then each axis (interior along `direction`, full range along
        the others for directional statements), then the variable range."""
        k = self.kernel()
        S, H, P = k.patch_size + 2 * k.halo_size, k.halo_size, k.patch_size
        rng = [(0, k.n_patches)]
        for axis in range(1, k.dim + 1):
            if direction >= 1:
                rng.append((H, P + H) if axis == direction else (0, S))
            else:
                rng.append((0, S) if expr is not None and expr[0] == k.LHS[0] else (H, P + H))
        # variable range: the reference takes the minimum over the statement's items (matched as
        # substrings, CPPPrinter.py:118-126) and the statement's own struct_inclusion
        text = str(expr)
        span = min([v for name, v in k.item_struct.items() if name in text] + [struct_inclusion])
        rng.append({0: (0, 1), 1: (0, k.n_real), 2: (0, k.n_real + k.n_aux)}.get(span, (0, 1)))
        return rng

    def _stage_a_family(self):
        """Which stage-A kernel the library picks for (dim, order) -- the rule of `exa_dg_stage_a_kernel()` (dg_inst.hip), restated for the text that
        exists before a plan does; compile() replaces it by the name the plan itself reports."""
        k = self.kernel()
        N = k.patch_size
        if self.n_picard == 0:
            return "dg_stage_a_single_kernel<%d,%d>" % (k.dim, N)
        if k.dim == 3 and N == 6:
            return "dg_stage_a_reg_kernel<6> (iterate in registers, two cells per workgroup)"
        if k.dim == 3 and N == 8:
            return "dg_stage_a_m8_kernel (iterate in registers, derivative contraction on v_mfma_f64_4x4x4_4b_f64)"
        if k.dim == 3 and N == 7:
            return "dg_stage_a_stream_kernel<7> (level-streamed)"
        return "dg_stage_a_kernel<%d,%d>" % (k.dim, N)

    def _plan_text(self, stage_a=None):
        k = self.kernel()
        S = k.patch_size + 2 * k.halo_size
        V = k.n_real + k.n_aux
        L = ["// exahype_amd HIP dispatch plan for `%s` (MI355X / gfx950)" % self.functionName(),
             "// scheme     : %s" % self.scheme,
             "// pde terms  : %s  <- %s" % (_PDE_DOC.get(self.pde, "user term set from SymPy expressions (pde_codegen.SympyPDE, JIT-compiled)"),
                                            ", ".join(k.functions)),
             "// chosen by  : %s" % self.pde_origin]
        if self.scheme == "aderdg":
            N = k.patch_size
            L += ["// kernels    : %s (predictor + volume + face traces%s), dg_stage_b_kernel<%d,%d> (Riemann + corrector)"
                  % (stage_a or self._stage_a_family(), "" if stage_a else "; exa_dg_stage_a_kernel(plan) names the launched one", k.dim, N),
                  "// C-ABI      : exa_dg_plan_create(dev, %d, %d, %d, %d, %d, {%s}, &plan); exa_dg_predictor_volume; exa_dg_riemann_corrector"
                  % (k.dim, N, k.n_real, self.pde, self.n_picard, ",".join(map(str, self.grid))),
                  "// arrays     : u[%d][%s][%d] fp64, AoS (reference layout), updated in place" % (k.n_patches, "][".join([str(N)] * k.dim), k.n_real)]
        else:
            mode = 0 if self.scheme == "fv-rusanov-faithful" else 1
            if self.cell_data:
                L += ["// flavour    : exahype2::CellData (examples/kernel-generator.py): %s with halo is read, %s (halo-less) is written; the"
                      % (k.items[2], k.items[1]),
                      "//              statement list as written is not executable (reads the output item first, wave speed through the flux",
                      "//              function): the corrected Rusanov update it stands for is dispatched, out of place",
                      "// C-ABI      : exa_fv_time_step_device_oop(plan, %s, %s, cellCentre, t, dt, h = cellSize / patch_size, stream)"
                      % (k.items[2], k.items[1])]
            L += ["// kernel     : fv_rusanov_kernel<%d, mode %d> -- %d statements fused into one launch, one workgroup per patch"
                  % (k.dim, mode, len(k.LHS)),
                  "// C-ABI      : exa_fv_plan_create(dev, %d, %d, %d, %d, %d, %d, %d, %d, &plan); exa_fv_time_step_device(plan, %s, %s, h, stream)"
                  % (mode, k.dim, k.patch_size, k.halo_size, k.n_real, k.n_aux, k.n_patches, self.pde, k.items[0] if k.items else "Q",
                     k.inputs[0] if k.inputs else "dt"),
                  "// arrays     : %s[%d][%s][%d] fp64, AoS (reference layout), interior updated in place"
                  % (k.items[0] if k.items else "Q", k.n_patches, "][".join([str(S)] * k.dim), V)]
        L.append("// statements :")
        for n, (l, r, d, s) in enumerate(zip(k.LHS, k.RHS, k.directions, k.struct_inclusion)):
            if str(l) in k.directional_consts:
                L.append("//   [%2d] %s = %s" % (n, l, r))
            else:
                L.append("//   [%2d] %s%s   ranges %s" % (n, l, "" if r is None else " = %s" % (r,), self.loop([l, r], d, k.dim + 1, s)))
        return "\n".join(L) + "\n"

    # -- execution ----------------------------------------------------------------------------------
    @staticmethod
    def _pde_from_bodies(k):
        """SymPy bodies given with `kernel.function(..., body=...)` for the flux and the eigenvalue -> a SympyPDE."""
        bodies = getattr(k, "function_bodies", {})
        flux = next((bodies[n] for n in bodies if n.lower() == "flux"), None)
        eig = next((bodies[n] for n in bodies if n.lower() in ("maxeigenvalue", "max_eigenvalue")), None)
        if flux is None and eig is None:
            return None
        if flux is None or eig is None:
            raise ValueError("SymPy bodies are needed for both the flux and the eigenvalue function (got one)")
        # optional: the algebraic source term, under the name the reference's harness uses for the hook
        # (`Unit test/correctness_test.cpp:16-23`): body(q) -> n_real expressions
        src = next((bodies[n] for n in bodies if n.lower() in ("sourceterm", "source")), None)
        from ..pde_codegen import SympyPDE
        return SympyPDE(k.n_real, flux=flux, max_eigenvalue=eig, max_dim=k.dim, source=src)

    def compile(self):
        """Bind libexahype_hip.so (built with hipcc if missing) and create the plan.  Raises without a GPU."""
        from .. import solvers
        k = self.kernel()
        if self.scheme == "statements":
            if self._impl is None:
                self._impl = self.lowering.bind(self.device)
            return self._impl
        if self.user_pde is not None and self.pde < 0:
            self.pde = self.user_pde.register()
        if self._impl is None:
            if self.scheme == "aderdg":
                self._impl = solvers.AderDgSolver(k.dim, k.patch_size, self.grid, pde=self.pde, n_vars=k.n_real,
                                                  n_picard=self.n_picard, device=self.device)
                self.code = self._plan_text(stage_a=self._impl.stage_a_kernel_name())     # the plan's own word for its stage A
            else:
                mode = solvers.FV_FAITHFUL if self.scheme == "fv-rusanov-faithful" else solvers.FV_RUSANOV
                self._impl = solvers.FVRusanovKernel(k.dim, k.patch_size, k.halo_size, k.n_real, k.n_aux, k.n_patches,
                                                     pde=self.pde, mode=mode, device=self.device)
        return self._impl

    def run(self, Q, *consts, h=1.0, dx=None, steps=1):
        """FV: `time_step(Q, dt)` in place on a numpy array (staged) or CUDA tensor (resident).
        ADER-DG: `steps` time steps of u (numpy AoS [cells][nodes][vars]) in place.
        Lowered statement list: `time_step(Q, c0, c1, ...)` with the builder's input constants in declaration order."""
        if self.cell_data:
            raise TypeError("the CellData flavour is out of place: use run_cell_data(QIn, dt, t, cell_centre, cell_size)")
        impl = self.compile()
        if self.scheme == "statements":
            for _ in range(steps):
                impl.run(Q, *consts)
            return Q
        if len(consts) != 1:
            raise TypeError("run(Q, dt): this scheme takes the time step as its one constant")
        dt = consts[0]
        if self.scheme == "aderdg":
            if dx is not None:
                impl.dx = [float(x) for x in dx]
            impl.upload(np.asarray(Q))
            for _ in range(steps):
                impl.step(dt)
            Q[...] = impl.download().reshape(Q.shape)
            return Q
        for _ in range(steps):
            impl.time_step(Q, dt, h)
        return Q

    def run_cell_data(self, QIn, dt, t=0.0, cell_centre=None, cell_size=1.0, out=None):
        """`time_step(patchData)` of the CellData flavour: QIn [n_patches][(P+2H)^dim][n_real+n_aux] is read, the halo-less QOut
        [n_patches][P^dim][n_real+n_aux] is returned (numpy in -> numpy out, CUDA tensors stay on the device).  cell_centre:
        [n_patches][dim] (default: the origin), cell_size: edge length of a patch's cell (volume size h = cell_size / patch_size)."""
        if not self.cell_data:
            raise TypeError("this kernel updates its array in place: run(Q, dt)")
        impl = self.compile()
        return impl.time_step_oop(QIn, dt, h=float(cell_size) / self.kernel().patch_size, t=t, centres=cell_centre, out=out)

    __call__ = run
