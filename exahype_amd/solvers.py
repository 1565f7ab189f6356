"""Host-side drivers over the C-ABI: device buffers (torch), streams, the Cartesian
block partition and the face-trace halo exchange (torch.distributed; backend
"nccl" is RCCL over xGMI on MI355X).

The reference has no host driver of its own -- its generated `time_step` is called
from a hand-written `main` (`Unit test/correctness_test.cpp:176-205`) or, by
intent, from a Peano enclave task (`exahype/printers/CPPPrinter.py:346`).  These
classes play that role for the HIP kernels: they own nothing numerical.
"""
import ctypes as C
import math
import os

import numpy as np

from . import _lib
from ._lib import (FV_FAITHFUL, FV_RUSANOV, PDE_ADVECTION, PDE_EULER, PDE_EULER_REF2D, check, darr, larr)


def _torch():
    import torch
    return torch


def _stream_ptr():
    torch = _torch()
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ----------------------------------------------------------------------------------------------
# Finite-Volume Rusanov patch kernel == device form of `time_step` (Unit test/test.h:3)
# ----------------------------------------------------------------------------------------------
class FVRusanovKernel:
    """`time_step(Q, dt)` on MI355X for `n_patches` patches of the reference layout
    Q[patch][i][j]([k])[var] (halo included)."""

    def __init__(self, dim, patch_size, halo_size, n_real, n_aux, n_patches=1, pde=PDE_EULER_REF2D,
                 mode=FV_FAITHFUL, device=0):
        self.lib = _lib.load()
        self.shape = (n_patches,) + (patch_size + 2 * halo_size,) * dim + (n_real + n_aux,)
        self.device = device
        h = C.c_void_p()
        check(self.lib.exa_fv_plan_create(device, mode, dim, patch_size, halo_size, n_real, n_aux, n_patches, pde,
                                          C.byref(h)))
        self._plan = h
        self.count = self.lib.exa_fv_q_count(h)

    def time_step(self, Q, dt, h=1.0, slot=None, t=0.0, centres=None):
        """In place.  numpy array -> staged through HBM by the library; CUDA tensor -> no copies.
        slot (CUDA int64 tensor, one entry per patch): patches with a negative entry are skipped -- for patch arrays
        whose number of entries in use is known on the device only.
        centres ([n_patches][dim], CUDA tensor or array) and t: the patch centres and the time for term sets whose terms depend on position /
        time (CUDA Q only; default: every patch centred at the origin, t = 0)."""
        if isinstance(Q, np.ndarray):
            if Q.dtype != np.float64 or not Q.flags.c_contiguous or Q.size != self.count:
                raise ValueError("Q must be a C-contiguous float64 array of %d entries" % self.count)
            if centres is not None or t != 0.0:                      # coordinates: staged through a device tensor here
                torch = _torch()
                qd = torch.as_tensor(Q).to(torch.device("cuda", self.device))
                self.time_step(qd, dt, h, slot, t, centres)
                Q[...] = qd.cpu().numpy()
                return Q
            check(self.lib.exa_fv_time_step_host(self._plan, Q.ctypes.data_as(C.c_void_p), dt, h))
            return Q
        torch = _torch()
        if not (Q.is_cuda and Q.dtype == torch.float64 and Q.is_contiguous() and Q.numel() == self.count):
            raise ValueError("Q must be a contiguous float64 CUDA tensor of %d entries" % self.count)
        if slot is not None:
            if not (slot.is_cuda and slot.dtype == torch.int64 and slot.is_contiguous() and slot.numel() >= self.shape[0]):
                raise ValueError("slot must be a contiguous int64 CUDA tensor with one entry per patch")
            if centres is not None or t != 0.0:
                dim = len(self.shape) - 2
                if centres is None or not (isinstance(centres, torch.Tensor) and centres.is_cuda and centres.dtype == torch.float64 and centres.is_contiguous()
                                           and centres.numel() == self.shape[0] * dim):
                    raise ValueError("the masked patch update with coordinates needs centres: a contiguous float64 CUDA tensor [n_patches][dim]")
                check(self.lib.exa_fv_time_step_device_masked_at(self._plan, C.c_void_p(Q.data_ptr()), C.c_void_p(slot.data_ptr()),
                                                                 C.c_void_p(centres.data_ptr()), t, dt, h, _stream_ptr()))
                return Q
            check(self.lib.exa_fv_time_step_device_masked(self._plan, C.c_void_p(Q.data_ptr()), C.c_void_p(slot.data_ptr()),
                                                          dt, h, _stream_ptr()))
            return Q
        if centres is not None or t != 0.0:
            cen = None
            if centres is not None:
                cen = centres if isinstance(centres, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(centres), dtype=torch.float64).to(Q.device)
                dim = len(self.shape) - 2
                if not (cen.is_cuda and cen.dtype == torch.float64 and cen.is_contiguous() and cen.numel() == self.shape[0] * dim):
                    raise ValueError("centres must be a contiguous float64 CUDA tensor [n_patches][dim]")
            check(self.lib.exa_fv_time_step_device_at(self._plan, C.c_void_p(Q.data_ptr()), C.c_void_p(cen.data_ptr()) if cen is not None else None,
                                                      t, dt, h, _stream_ptr()))
            return Q
        check(self.lib.exa_fv_time_step_device(self._plan, C.c_void_p(Q.data_ptr()), dt, h, _stream_ptr()))
        return Q

    def time_step_oop(self, QIn, dt, h=1.0, t=0.0, centres=None, out=None):
        """The `exahype2::CellData` flavour (`examples/kernel-generator.py`): QIn [n_patches][(P+2H)^dim][n_real+n_aux] is only read,
        the result goes to a separate halo-less array [n_patches][P^dim][n_real+n_aux] (returned; `out` to reuse one).  centres:
        [n_patches][dim] cell centres (default: the origin) and t reach PDE terms that depend on position and time.  numpy in ->
        numpy out (staged through the device); CUDA tensors stay on the device."""
        torch = _torch()
        host = isinstance(QIn, np.ndarray)
        dev = torch.device("cuda", self.device)
        qi = torch.as_tensor(np.ascontiguousarray(QIn), dtype=torch.float64).to(dev) if host else QIn
        if not (qi.is_cuda and qi.dtype == torch.float64 and qi.is_contiguous() and qi.numel() == self.count):
            raise ValueError("QIn must hold %d contiguous float64 entries" % self.count)
        n_out = self.lib.exa_fv_qout_count(self._plan)
        dim = len(self.shape) - 2
        P = round((n_out // (self.shape[0] * self.shape[-1])) ** (1.0 / dim))
        if host and out is not None and not (isinstance(out, np.ndarray) and out.dtype == np.float64 and out.flags.c_contiguous and out.size == n_out):
            raise ValueError("out (numpy input): a C-contiguous float64 array of %d entries" % n_out)
        qo = torch.empty((self.shape[0],) + (P,) * dim + (self.shape[-1],), dtype=torch.float64, device=dev) if out is None or host else out
        if not (isinstance(qo, torch.Tensor) and qo.is_cuda and qo.device == dev and qo.dtype == torch.float64 and qo.is_contiguous() and qo.numel() == n_out):
            raise ValueError("out must be a contiguous float64 CUDA tensor of %d entries on %s" % (n_out, dev))
        cen = None
        if centres is not None:
            cen = torch.as_tensor(np.ascontiguousarray(centres), dtype=torch.float64).to(dev) if not isinstance(centres, torch.Tensor) else centres
            if not (cen.is_cuda and cen.device == dev and cen.dtype == torch.float64 and cen.is_contiguous() and cen.numel() == self.shape[0] * dim):
                raise ValueError("centres must be a contiguous float64 CUDA tensor [n_patches][dim] on %s (or an array)" % dev)
        check(self.lib.exa_fv_time_step_device_oop(self._plan, C.c_void_p(qi.data_ptr()), C.c_void_p(qo.data_ptr()),
                                                   C.c_void_p(cen.data_ptr()) if cen is not None else None, t, dt, h, _stream_ptr()))
        if host:
            res = qo.cpu().numpy()
            if out is not None:
                out.reshape(res.shape)[...] = res
                return out
            return res
        return qo

    def __del__(self):
        try:
            self.lib.exa_fv_plan_destroy(self._plan)
        except Exception:
            pass


def pde_eval(pde, normal, Q):
    """Point-wise Flux / maxEigenvalue (Functions.h:2-3) of states Q[n][stride] on the GPU."""
    torch = _torch()
    lib = _lib.load()
    Qd = torch.as_tensor(np.ascontiguousarray(Q), dtype=torch.float64).cuda()
    n, stride = Qd.shape
    F = torch.zeros_like(Qd)
    lam = torch.zeros(n, dtype=torch.float64, device=Qd.device)
    check(lib.exa_pde_eval_device(pde, normal, n, stride, C.c_void_p(Qd.data_ptr()), C.c_void_p(F.data_ptr()),
                                  C.c_void_p(lam.data_ptr()), _stream_ptr()))
    torch.cuda.synchronize()
    return F.cpu().numpy(), lam.cpu().numpy()


# ----------------------------------------------------------------------------------------------
# Cartesian partition of a regular patch over the ranks of one node
# ----------------------------------------------------------------------------------------------
class CartesianPartition:
    """Process grid for `world` ranks in `dim` dimensions (8 -> 2x2x2, 4 -> 2x2x1,
    2 -> 2x1x1), rank = row-major index; periodic neighbours."""

    def __init__(self, world, rank, dim, pdims=None, exchange_self=()):
        # exchange_self: directions of process-grid extent 1 whose periodic wrap goes through the exchange anyway (a rank is
        # its own neighbour: RCCL send/recv to self) -- lets ONE GPU run the sharded step over the real transport
        self.exchange_self = tuple(int(d) for d in exchange_self)
        if pdims is None:
            pdims = [1] * dim
            w, a = world, 0
            while w > 1:
                f = next(p for p in (2, 3, 5, 7, w) if w % p == 0)
                pdims[a % dim] *= f
                w //= f
                a += 1
        pdims = list(pdims) + [1] * (3 - len(pdims))
        if int(np.prod(pdims)) != world:
            raise ValueError("process grid %s does not match world size %d" % (pdims, world))
        self.world, self.rank, self.dim, self.pdims = world, rank, dim, pdims
        self.coords = [rank // (pdims[1] * pdims[2]), (rank // pdims[2]) % pdims[1], rank % pdims[2]]

    def rank_of(self, coords):
        c = [coords[a] % self.pdims[a] for a in range(3)]
        return (c[0] * self.pdims[1] + c[1]) * self.pdims[2] + c[2]

    def neighbour(self, d, sign):
        c = list(self.coords)
        c[d] += sign
        return self.rank_of(c)

    def partitioned(self, d):
        return self.pdims[d] > 1 or d in self.exchange_self

    def shell_and_interior(self, nc):
        """Disjoint cell boxes: the layers touching a partitioned block face, and the rest."""
        dim = self.dim
        lo, hi = [0] * dim, [int(c) for c in nc]
        shell = []
        for d in range(dim):
            if not self.partitioned(d):
                continue
            n = hi[d] - lo[d]
            if n <= 0:
                break
            layers = [(lo[d], lo[d] + 1)] if n == 1 else [(lo[d], lo[d] + 1), (hi[d] - 1, hi[d])]
            for a, b in layers:
                b_lo, b_hi = list(lo), list(hi)
                b_lo[d], b_hi[d] = a, b
                shell.append((b_lo, b_hi))
            lo[d] += 1
            hi[d] = max(hi[d] - 1, lo[d])
        return shell, (lo, hi)


class HaloExchange:
    """Face-trace exchange between neighbouring blocks.  Works on any torch tensors
    (CUDA with the nccl/RCCL backend; CPU with gloo for the rehearsal tests).

    trace: tensor [dim, 2, nc0, nc1, nc2, TS]; ghosts[d*2+s]: [transverse cells, TS].
    A process grid of extent 2 in a direction (both neighbours are the same rank) sends one
    message per peer holding both layers; larger extents send one layer to each neighbour."""

    def __init__(self, part, nc, ts, device, dtype=None, stage_through_host=False):
        torch = _torch()
        self.part, self.nc, self.ts = part, list(nc) + [1] * (3 - len(nc)), ts
        self.stage = stage_through_host
        dtype = dtype or torch.float64
        self.ghost = [None] * 6
        self.send = [None] * 6
        # per partitioned direction one contiguous pair buffer each way, so that a process grid of extent 2
        # (both neighbours are the same rank: every case at 2, 4 and 8 GPUs) needs ONE message per peer:
        #   send pair = (my L layer, my R layer);  ghost pair = (peer's L layer -> my high ghost, peer's R -> my low)
        self._send_pair = [None] * 3
        self._ghost_pair = [None] * 3
        for d in range(part.dim):
            if part.partitioned(d):
                nt = int(np.prod(self.nc)) // self.nc[d]
                self._send_pair[d] = torch.zeros(2, nt, ts, dtype=dtype, device=device)
                self._ghost_pair[d] = torch.zeros(2, nt, ts, dtype=dtype, device=device)
                self.send[d * 2 + 0], self.send[d * 2 + 1] = self._send_pair[d][0], self._send_pair[d][1]
                self.ghost[d * 2 + 1], self.ghost[d * 2 + 0] = self._ghost_pair[d][0], self._ghost_pair[d][1]

    def pack(self, trace):
        """CPU rehearsals only (tests/test_partition_halo.py: gloo, plain tensors).  The product path packs on the device with
        exa_dg_pack_face (AderDgSolver._pack_faces)."""
        for d in range(self.part.dim):
            if not self.part.partitioned(d):
                continue
            # side 0: L traces of the layer c_d = 0; side 1: R traces of the layer c_d = nc_d - 1
            lo = trace[d, 0].select(d, 0)
            hi = trace[d, 1].select(d, self.nc[d] - 1)
            self.send[d * 2 + 0].copy_(lo.reshape(-1, self.ts))
            self.send[d * 2 + 1].copy_(hi.reshape(-1, self.ts))

    def start(self):
        import torch.distributed as dist
        ops, self._staged = [], []
        for d in range(self.part.dim):
            if not self.part.partitioned(d):
                continue
            lo_n, hi_n = self.part.neighbour(d, -1), self.part.neighbour(d, +1)
            if lo_n == hi_n:                           # extent 2: one message per peer, both layers
                sp, gp = self._send_pair[d], self._ghost_pair[d]
                if self.stage:                         # gloo rehearsal with device tensors: go through host copies
                    hp = gp.cpu()
                    self._staged.append((gp, hp))
                    sp, gp = sp.cpu(), hp
                ops += [dist.P2POp(dist.isend, sp, lo_n, tag=2 * d), dist.P2POp(dist.irecv, gp, lo_n, tag=2 * d)]
                continue
            s_lo, s_hi = self.send[d * 2 + 0], self.send[d * 2 + 1]
            g_lo, g_hi = self.ghost[d * 2 + 0], self.ghost[d * 2 + 1]
            if self.stage:
                s_lo, s_hi = s_lo.cpu(), s_hi.cpu()
                h_lo, h_hi = g_lo.cpu(), g_hi.cpu()
                self._staged += [(g_lo, h_lo), (g_hi, h_hi)]
                g_lo, g_hi = h_lo, h_hi
            # my L layer -> low neighbour (its high ghost); my R layer -> high neighbour (its low ghost).
            # receives are posted in the order the peers send.
            ops += [dist.P2POp(dist.isend, s_lo, lo_n, tag=2 * d), dist.P2POp(dist.isend, s_hi, hi_n, tag=2 * d + 1),
                    dist.P2POp(dist.irecv, g_hi, hi_n, tag=2 * d), dist.P2POp(dist.irecv, g_lo, lo_n, tag=2 * d + 1)]
        self._reqs = dist.batch_isend_irecv(ops) if ops else []

    def finish(self):
        for r in self._reqs:
            r.wait()
        for dev, host in getattr(self, "_staged", []):
            dev.copy_(host)
        self._reqs, self._staged = [], []

    def ghost_ptrs(self):
        arr = (C.c_void_p * 6)()
        for f in range(6):
            arr[f] = self.ghost[f].data_ptr() if self.ghost[f] is not None else None
        return arr


def _sees_position_and_time(lib, pde):
    """include/exahype_hip.h EXA_PDE_FLAG_XT: the CFL scans of such a term set hand the positions of the states and the time to
    exa_pde_eval_device_at instead of using the coordinate-free reductions."""
    return bool(lib.exa_pde_flags(int(pde)) & 1)


def _max_eigenvalue_at(lib, pde, dim, states, positions, t):
    """max over the directions and the states [n][V] (CUDA) of the eigenvalue at positions [n][3] and time t; a 1-element CUDA tensor"""
    torch = _torch()
    lam = torch.empty(states.shape[0], dtype=torch.float64, device=states.device)
    best = torch.zeros(1, dtype=torch.float64, device=states.device)
    for d in range(dim):
        check(lib.exa_pde_eval_device_at(int(pde), d, states.shape[0], states.shape[1], C.c_void_p(states.data_ptr()),
                                         C.c_void_p(positions.data_ptr()), float(t), None, C.c_void_p(lam.data_ptr()), _stream_ptr()))
        best = torch.maximum(best, lam.max().reshape(1))
    return best


# ----------------------------------------------------------------------------------------------
# ADER-DG solver on a regular Cartesian block
# ----------------------------------------------------------------------------------------------
class AderDgSolver:
    """One block of `ncells` DG cells of order N-1 on one MI355X.

    u     : CUDA tensor [nc0, nc1, (nc2,) N, N, (N,) n_vars]   (reference AoS layout)
    trace : CUDA tensor [dim, 2, nc0, nc1, nc2, 2*n_vars*N^(dim-1)]

    With `part` (CartesianPartition) the block is one shard of a periodic global
    grid: stage A runs on the boundary shell first, the face traces travel over
    RCCL on a second stream while the interior cells run stage A, stage B follows.

    stage_a          "auto" | "lds" | "reg": which stage-A kernel serves 3-D N = 6 / N = 8 (include/exahype_hip.h EXA_STAGE_A_*)
    one_kernel_step  True: Riemann solve + corrector of a step run in front of the next step's predictor (3-D, N = 6, "reg"); step() then
                     leaves the corrector pending and reading `u` applies it.  Measured equal to two kernels: off by default.
    origin, time     physical coordinates of the GLOBAL grid's origin and the start time: only term sets whose terms depend on position /
                     time see them (pde_codegen.SympyPDE with flux(q, x, t, d) ...); step() advances `time`.
    """

    STAGE_A = {"auto": 0, "lds": 1, "reg": 2}      # include/exahype_hip.h EXA_STAGE_A_*

    def __init__(self, dim, N, ncells, pde=PDE_EULER, n_vars=5, n_picard=-1, dx=None, device=0, part=None,
                 backend_is_gloo=False, fused_single_stage=True, stage_a="auto", reserve_cus=None, one_kernel_step=False, origin=None, time=0.0):
        torch = _torch()
        self.lib = _lib.load()
        self.dim, self.N, self.nv, self.pde = dim, N, n_vars, pde
        self.nc = [int(c) for c in ncells]
        if len(self.nc) != dim:
            raise ValueError("ncells must have %d entries" % dim)
        self.dev = torch.device("cuda", device)
        h = C.c_void_p()
        check(self.lib.exa_dg_plan_create(device, dim, N, n_vars, pde, n_picard, larr(self.nc), C.byref(h)))
        self._plan = h
        if stage_a != "auto":
            check(self.lib.exa_dg_plan_set_stage_a(h, self.STAGE_A[stage_a]))
        self.dx = [float(x) for x in (dx if dx is not None else [1.0 / c for c in self.nc])]
        # where and when (term sets with position- / time-dependent terms only: include/exahype_hip.h exa_dg_plan_set_origin_time): physical
        # coordinates of the GLOBAL grid's origin -- a shard adds its offset -- and the time, which step() advances
        go = [float(x) for x in (origin if origin is not None else [0.0] * dim)]
        self.origin = [go[a] + (part.coords[a] * self.nc[a] * self.dx[a] if part is not None else 0.0) for a in range(dim)]
        self.time = float(time)
        self.nf = N ** (dim - 1)
        self.ts = 2 * n_vars * self.nf
        self._u = torch.zeros(tuple(self.nc) + (N,) * dim + (n_vars,), dtype=torch.float64, device=self.dev)
        nc3 = self.nc + [1] * (3 - dim)
        self.trace = torch.zeros((dim, 2) + tuple(nc3) + (self.ts,), dtype=torch.float64, device=self.dev)
        assert self._u.numel() == self.lib.exa_dg_dof_count(h) and self.trace.numel() == self.lib.exa_dg_trace_count(h)
        # The step as ONE kernel (include/exahype_hip.h exa_dg_corrector_predictor; 3-D, N = 6, register-resident stage A): step() leaves
        # the block as (u*, traces) with the corrector PENDING, the next step's kernel applies it in front of its predictor, and reading
        # `u` (download, max_eigenvalue, ...) applies it with the stand-alone stage B first.  Costs a second trace array.  Off by default:
        # measured equal to 4 % slower than the two-kernel step (22.2 against 21.4 - 22.1 ms per step at 64^3, profiles/r03_one_kernel_step.txt) --
        # stage B's 2.0 ms are traded for a prologue of 0.8 ms of arithmetic + 0.9 ms of exposed trace-load latency + two barriers.
        has = bool(self.lib.exa_dg_has_corrector_predictor(h))
        if one_kernel_step and not has:
            raise ValueError("one_kernel_step: not built for these settings (3-D, N = 6, stage_a 'reg', n_picard >= 1)")
        self._one_kernel = bool(one_kernel_step)
        self._pending_dt = None
        self._trace2 = None
        self.part = part
        self.halo = None
        self._fused = bool(fused_single_stage) and bool(self.lib.exa_dg_has_fused_step(h))
        self._u2 = None
        # measurement hooks (bench.py): stage_a_events collects one (start, end) event pair per stage-A launch on the
        # launching stream; exchange_events one (shell done, comm start, comm end, interior start, interior end, pack end) tuple
        # per sharded step.  None = off: the product path records nothing.
        self.stage_a_events = None
        self.exchange_events = None
        self._cfl_out = None              # run(): 1-element tensor the step's stage B leaves the next CFL scan in (riemann_corrector(lam_out=))
        if part is not None and any(part.partitioned(d) for d in range(dim)):
            self.halo = HaloExchange(part, self.nc, self.ts, self.dev, stage_through_host=backend_is_gloo)
            # High priority: the pack copies and the RCCL transport kernels are dispatched ahead of the persistent interior launch, which
            # otherwise fills every CU first (measured on the RCCL-to-self rehearsal: the exchange span fell from 166 ms -- served as
            # interior workgroups retired -- to 0.44 ms, profiles/r03_bench_cfg2_self_exchange.json).  `reserve_cus` > 0 additionally keeps
            # the persistent grids that many workgroups below the resident count; it costs 0.4 % of stage A per CU and bought nothing
            # in the rehearsal, so the default is 0 (EXA_RESERVE_CUS / the argument are there for a node where RCCL needs resident CUs).
            self.comm_stream = torch.cuda.Stream(device=self.dev, priority=-1)
            self.set_reserve_cus(int(os.environ.get("EXA_RESERVE_CUS", "0")) if reserve_cus is None else int(reserve_cus))
            self.shell, self.interior = part.shell_and_interior(self.nc)

    def set_reserve_cus(self, workgroups):
        """Keep the persistent stage-A grids `workgroups` below the resident count (CUs left to RCCL's transport kernels while the interior
        launch runs); bench.py tries 0 and 8 in its warm-up on a real multi-GPU node and keeps the faster."""
        self.reserve_cus = int(workgroups)
        check(self.lib.exa_dg_plan_set_stage_a_reserve(self._plan, self.reserve_cus))

    # -- data movement ---------------------------------------------------------------------
    @property
    def u(self):
        """The degrees of freedom (a pending corrector of the one-kernel step is applied first)."""
        self.flush()
        return self._u

    @u.setter
    def u(self, value):
        self._pending_dt = None
        self._u = value

    def flush(self):
        """Apply the corrector the last one-kernel step left pending (stand-alone stage B); a no-op otherwise."""
        if self._pending_dt is not None:
            dt, self._pending_dt = self._pending_dt, None
            self.riemann_corrector(dt)

    def upload(self, u_host):
        torch = _torch()
        self._pending_dt = None
        self._u.copy_(torch.as_tensor(np.ascontiguousarray(u_host), dtype=torch.float64).reshape(self._u.shape))

    def download(self):
        return self.u.cpu().numpy()

    def operators(self):
        N = self.N
        out = {k: np.zeros(s) for k, s in (("xi", N), ("w", N), ("D", (N, N)), ("Kxi", (N, N)), ("phiL", N), ("phiR", N),
                                             ("iK1", (N, N)))}
        check(self.lib.exa_dg_operators(self._plan, *[out[k].ctypes.data_as(C.c_void_p) for k in
                                                      ("xi", "w", "D", "Kxi", "phiL", "phiR", "iK1")]))
        out["N"] = N
        out["F0"] = out["phiL"].copy()
        return out

    def stage_a_kernel_name(self):
        """The kernel `predictor_volume` launches for this plan (as rocprofv3 names it, without the argument list)."""
        return self.lib.exa_dg_stage_a_kernel(self._plan).decode()

    def work(self):
        v = [C.c_double() for _ in range(4)]
        check(self.lib.exa_dg_work(self._plan, *[C.byref(x) for x in v]))
        return dict(flop_a=v[0].value, flop_b=v[1].value, bytes_a=v[2].value, bytes_b=v[3].value)

    # -- kernels -------------------------------------------------------------------------------
    def _where_and_when(self):
        check(self.lib.exa_dg_plan_set_origin_time(self._plan, darr(self.origin), self.time))

    def predictor_volume(self, dt, lo=None, hi=None):
        self._where_and_when()
        ev = None
        if self.stage_a_events is not None:
            torch = _torch()
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        check(self.lib.exa_dg_predictor_volume_box(self._plan, C.c_void_p(self.u.data_ptr()), C.c_void_p(self.trace.data_ptr()),
                                                   larr(lo) if lo is not None else None, larr(hi) if hi is not None else None,
                                                   dt, darr(self.dx), _stream_ptr()))
        if ev is not None:
            ev[1].record()
            self.stage_a_events.append(ev)

    def corrector_predictor(self, dt_prev, dt, lo=None, hi=None):
        """One-kernel step on a cell box: corrector of the previous step (its traces in self.trace, its time step dt_prev) + predictor of
        this one; the new traces go to the second trace array (swap_traces() once every box of the step has run)."""
        torch = _torch()
        if self._trace2 is None:
            self._trace2 = torch.zeros_like(self.trace)
        ev = None
        if self.stage_a_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        ghosts = self.halo.ghost_ptrs() if self.halo is not None else None
        check(self.lib.exa_dg_corrector_predictor(self._plan, C.c_void_p(self._u.data_ptr()), C.c_void_p(self.trace.data_ptr()),
                                                  C.c_void_p(self._trace2.data_ptr()), ghosts, larr(lo) if lo is not None else None,
                                                  larr(hi) if hi is not None else None, dt_prev, dt, darr(self.dx), None, _stream_ptr()))
        if ev is not None:
            ev[1].record()
            self.stage_a_events.append(ev)

    def swap_traces(self):
        self.trace, self._trace2 = self._trace2, self.trace

    def riemann_corrector(self, dt, lo=None, hi=None, lam_out=None):
        """Stage B on the box (default: the whole block).  lam_out (1-element float64 CUDA tensor): the launch also leaves the largest eigenvalue of
        the corrected u there -- the next step's CFL scan without a pass of its own (exa_dg_riemann_corrector_cfl; not for term sets that see x, t)."""
        ghosts = self.halo.ghost_ptrs() if self.halo is not None else None
        self._where_and_when()
        lo_, hi_ = (larr(lo) if lo is not None else None), (larr(hi) if hi is not None else None)
        if lam_out is not None:
            check(self.lib.exa_dg_riemann_corrector_cfl(self._plan, C.c_void_p(self._u.data_ptr()), C.c_void_p(self.trace.data_ptr()),
                                                        ghosts, lo_, hi_, dt, darr(self.dx), C.c_void_p(lam_out.data_ptr()), _stream_ptr()))
            return
        check(self.lib.exa_dg_riemann_corrector(self._plan, C.c_void_p(self._u.data_ptr()), C.c_void_p(self.trace.data_ptr()),
                                                ghosts, lo_, hi_, dt, darr(self.dx), _stream_ptr()))

    def can_fuse_cfl_scan(self):
        """True where step() ends in a stage-B launch over the whole block that can carry the next CFL scan: the two-kernel step (sharded or not) of a
        term set whose eigenvalue does not see x, t."""
        return not (self._fused or self._one_kernel or _sees_position_and_time(self.lib, self.pde))

    def node_positions(self):
        """[n_cells * N^dim][3] physical coordinates of the nodes (CUDA, cached): origin + (cell + xi_i) dx"""
        torch = _torch()
        if getattr(self, "_xnodes", None) is None:
            xi = torch.as_tensor(self.operators()["xi"], dtype=torch.float64, device=self.dev)
            shape = tuple(self.nc) + (self.N,) * self.dim
            X = torch.zeros(shape + (3,), dtype=torch.float64, device=self.dev)
            for a in range(self.dim):
                cs, ns = [1] * (2 * self.dim), [1] * (2 * self.dim)
                cs[a], ns[self.dim + a] = self.nc[a], self.N
                c = torch.arange(self.nc[a], dtype=torch.float64, device=self.dev).reshape(cs)
                X[..., a] = self.origin[a] + (c + xi.reshape(ns)) * self.dx[a]
            self._xnodes = X.reshape(-1, 3).contiguous()
        return self._xnodes

    def max_eigenvalue(self):
        torch = _torch()
        if _sees_position_and_time(self.lib, self.pde):
            return _max_eigenvalue_at(self.lib, self.pde, self.dim, self.u.reshape(-1, self.nv), self.node_positions(), self.time)
        out = torch.zeros(1, dtype=torch.float64, device=self.dev)
        check(self.lib.exa_dg_max_eigenvalue(self._plan, C.c_void_p(self.u.data_ptr()), C.c_void_p(out.data_ptr()), _stream_ptr()))
        return out

    def _pack_faces(self):
        """Outward traces of the block's boundary layers -> the contiguous send buffers (exa_dg_pack_face: one strided
        device copy per face, on the current stream)."""
        for d in range(self.dim):
            if not self.part.partitioned(d):
                continue
            for side in range(2):
                check(self.lib.exa_dg_pack_face(self._plan, C.c_void_p(self.trace.data_ptr()), d, side,
                                                C.c_void_p(self.halo.send[d * 2 + side].data_ptr()), _stream_ptr()))

    def step(self, dt):
        """One ADER-DG time step of the block (periodic, or one shard of a periodic grid); advances `time`."""
        self._step(dt)
        self.time += dt

    def _step(self, dt):
        torch = _torch()
        if self.halo is None:
            if self._fused:
                # single-stage 2-D scheme on one periodic block: one fused launch, traces stay on chip (exa_dg_fused.hpp);
                # u ping-pongs with a second array
                # (self.u is rebound: fetch it again after a step rather than holding on to the tensor)
                if self._u2 is None:
                    self._u2 = torch.empty_like(self.u)
                check(self.lib.exa_dg_step_fused(self._plan, C.c_void_p(self.u.data_ptr()), C.c_void_p(self._u2.data_ptr()),
                                                 dt, darr(self.dx), _stream_ptr()))
                self.u, self._u2 = self._u2, self.u
                return
            if self._one_kernel:
                if self._pending_dt is None:
                    self.predictor_volume(dt)
                else:
                    self.corrector_predictor(self._pending_dt, dt)
                    self.swap_traces()
                self._pending_dt = dt
                return
            self.predictor_volume(dt)
            self.riemann_corrector(dt, lam_out=self._cfl_out)
            return
        # (one-kernel step on a shard: the shell cells' kernel reads the ghosts the PREVIOUS step received -- complete, every step ends
        # with the wait for its exchange -- and is done before this step's exchange overwrites them; interior cells read no ghosts)
        fused = self._one_kernel and self._pending_dt is not None
        stage = (lambda lo, hi: self.corrector_predictor(self._pending_dt, dt, lo, hi)) if fused else (lambda lo, hi: self.predictor_volume(dt, lo, hi))
        cur = torch.cuda.current_stream(self.dev)
        timed = self.exchange_events is not None
        mk = (lambda: torch.cuda.Event(enable_timing=True)) if timed else (lambda: None)
        for lo, hi in self.shell:                      # boundary shell first ...
            stage(lo, hi)
        if fused:
            self.trace, self._trace2 = self._trace2, self.trace      # (the pack below reads the NEW traces; the interior kernel gets them swapped back)
        ready = torch.cuda.Event(enable_timing=timed)
        ready.record(cur)
        c0, p1, c1, i0, i1 = mk(), mk(), mk(), mk(), mk()
        with torch.cuda.stream(self.comm_stream):      # ... its traces travel on the comm stream ...
            self.comm_stream.wait_event(ready)
            if timed:
                c0.record()
            self._pack_faces()
            if timed:
                p1.record()                            # pack copies: c0 .. p1; the RCCL send / recv group: p1 .. c1
            self.halo.start()
        lo, hi = self.interior                         # ... while the interior cells run stage A
        if timed:
            i0.record()
        if fused:
            self.trace, self._trace2 = self._trace2, self.trace
        if all(h > l for l, h in zip(lo, hi)):
            stage(lo, hi)
        if fused:
            self.swap_traces()
        if timed:
            i1.record()
        with torch.cuda.stream(self.comm_stream):
            self.halo.finish()
            if timed:
                c1.record()
        cur.wait_stream(self.comm_stream)
        if self._one_kernel:
            self._pending_dt = dt
        else:
            self.riemann_corrector(dt, lam_out=self._cfl_out)
        if timed:
            self.exchange_events.append((ready, c0, c1, i0, i1, p1))

    def __del__(self):
        try:
            self.lib.exa_dg_plan_destroy(self._plan)
        except Exception:
            pass


# ----------------------------------------------------------------------------------------------
# The steps either side of the FV kernel: halo fill + time loop on a regular grid
# ----------------------------------------------------------------------------------------------
def fill_halos_periodic(Q, grid, dim, patch_size, halo_size):
    """Fill the halo layers of every patch of a periodic Cartesian grid of patches from its neighbours'
    interior layers (the role Peano's enclave task plays before it calls the generated `time_step`,
    reference `exahype/printers/CPPPrinter.py:346`).

    Q: tensor (torch, any device) or numpy array [g0, g1, (g2,) S, S, (S,) V], updated in place.
    Axis by axis over the full transverse range, so edges and corners are filled too."""
    P, H, S = patch_size, halo_size, patch_size + 2 * halo_size
    is_np = isinstance(Q, np.ndarray)
    roll = (lambda a, s, ax: np.roll(a, s, axis=ax)) if is_np else (lambda a, s, ax: a.roll(s, dims=ax))
    assert tuple(Q.shape[:dim]) == tuple(grid) and all(s == S for s in Q.shape[dim:2 * dim])
    for a in range(dim):
        lo = [slice(None)] * Q.ndim
        hi = [slice(None)] * Q.ndim
        src_hi = [slice(None)] * Q.ndim
        src_lo = [slice(None)] * Q.ndim
        lo[dim + a] = slice(0, H)                     # low halo  <- high interior layers of the low neighbour
        src_hi[dim + a] = slice(P, P + H)
        hi[dim + a] = slice(P + H, S)                 # high halo <- low interior layers of the high neighbour
        src_lo[dim + a] = slice(H, 2 * H)
        Q[tuple(lo)] = roll(Q[tuple(src_hi)], 1, a)
        Q[tuple(hi)] = roll(Q[tuple(src_lo)], -1, a)
    return Q


def fill_halos_dirichlet(Q, grid, dim, patch_size, halo_size, boundary):
    """Halo fill of a NON-periodic Cartesian grid of patches: interior patch faces take the neighbour's interior layers
    (as fill_halos_periodic), the halo layers on the domain boundary are set to the prescribed state.

    boundary: array-like broadcastable to [..., V] (one fixed state for the whole boundary), or a dict
    {(axis, side): state} with side 0 = low, 1 = high face.  Edge / corner halo entries on the boundary get the state of
    the LAST axis that touches them (axes are filled in order); the 2*dim+1-point Rusanov stencil never reads them."""
    P, H, S = patch_size, halo_size, patch_size + 2 * halo_size
    fill_halos_periodic(Q, grid, dim, patch_size, halo_size)      # interior faces (the wrap entries are overwritten below)
    is_np = isinstance(Q, np.ndarray)

    def state(a, side):
        b = boundary[(a, side)] if isinstance(boundary, dict) else boundary
        if is_np:
            return np.asarray(b, dtype=Q.dtype)
        torch = _torch()
        return torch.as_tensor(np.asarray(b, dtype=np.float64), dtype=Q.dtype, device=Q.device)
    for a in range(dim):
        lo = [slice(None)] * Q.ndim
        hi = [slice(None)] * Q.ndim
        lo[a], lo[dim + a] = 0, slice(0, H)                      # first patch along a: its low halo layers
        hi[a], hi[dim + a] = grid[a] - 1, slice(P + H, S)        # last patch along a: its high halo layers
        Q[tuple(lo)] = state(a, 0)
        Q[tuple(hi)] = state(a, 1)
    return Q


class FVPatchGrid:
    """A Cartesian grid of FV patches resident in HBM, advanced by the fused Rusanov kernel -- the role of the enclave task around the
    generated `time_step` (reference `exahype/printers/CPPPrinter.py:346`; the patch loop of `Unit test/correctness_test.cpp:118-174`).

    The states live HALO-LESS, `self.U` [g0, g1, (g2,) P, P, (P,) V] (as Peano keeps its patches); step(dt) is ONE launch
    (`exa_fv_grid_step_device`): the patch kernel assembles the patch with halo on chip, taking the states beyond a patch face from the face
    neighbour's interior layers (periodic wrap, or the prescribed state on a domain face with `boundary=`), and writes the new states to a
    second array; the two swap (`self.U` is rebound: fetch it again after a step).  No halo pass and no halo bytes in HBM.  run(t_end) takes dt
    from the CFL condition: the kernel that writes the new states reduces their eigenvalues on the way, so a step costs ONE host read of one
    double and no scan pass.  `with_halo()` materialises the reference layout with halo -- interiors + filled halo layers -- for a caller who
    wants it (output, the plain `time_step` entry points).  fused=False keeps the two-pass form on an array with halo (halo fill by torch ops,
    then the in-place update): the comparison the tests and the benchmark use."""

    def __init__(self, dim, grid, patch_size, halo_size=1, n_real=5, n_aux=0, pde=PDE_EULER, mode=FV_RUSANOV,
                 length=1.0, device=0, boundary=None, origin=None, time=0.0, fused=True):
        """origin, time: physical coordinates of the grid's low corner and the start time -- reach term sets whose terms depend on position /
        time (the patch centres follow from them); step() / run() advance the time."""
        torch = _torch()
        self.dim, self.grid, self.P, self.H = dim, tuple(int(g) for g in grid), patch_size, halo_size
        self.n_real, self.n_aux, self.pde = n_real, n_aux, pde
        self.h = length / (self.grid[0] * patch_size)                  # volume size
        n_patches = int(np.prod(self.grid))
        self.kernel = FVRusanovKernel(dim, patch_size, halo_size, n_real, n_aux, n_patches, pde, mode, device)
        self.lib = self.kernel.lib
        self.dev = torch.device("cuda", device)
        self.fused = bool(fused)
        V = n_real + n_aux
        if self.fused:
            self.U = torch.zeros(self.grid + (patch_size,) * dim + (V,), dtype=torch.float64, device=self.dev)
            self._U2 = None                                           # the other array of the step (allocated on first use)
        else:
            S = patch_size + 2 * halo_size
            self._Q = torch.zeros(self.grid + (S,) * dim + (V,), dtype=torch.float64, device=self.dev)
        self.time = float(time)
        self.boundary = boundary                                      # None: periodic
        self._bstate = None
        if boundary is not None:                                      # [2 dim][V]: state of the domain face (axis, side)
            b = np.zeros((2 * dim, V))
            for a in range(dim):
                for side in range(2):
                    b[a * 2 + side] = np.broadcast_to(np.asarray(boundary[(a, side)] if isinstance(boundary, dict) else boundary, dtype=np.float64), (V,))
            self._bstate = torch.as_tensor(b).to(self.dev).contiguous()
            # the prescribed states take part in the CFL scan (they sit in the halo layers a two-pass driver would scan); term sets that
            # depend on position / time: at the origin and t = 0 -- a bound the caller may tighten through `cfl`
            self._lam_boundary = max(float(np.max(pde_eval(pde, d, b)[1])) for d in range(dim))
        og = [float(x) for x in (origin if origin is not None else [0.0] * dim)]
        idx = np.stack(np.meshgrid(*[np.arange(g) for g in self.grid], indexing="ij"), axis=-1).reshape(-1, dim)
        self.centres = torch.as_tensor(np.asarray(og)[None, :] + (idx + 0.5) * patch_size * self.h, dtype=torch.float64).to(self.dev).contiguous()
        self._lam = torch.zeros(1, dtype=torch.float64, device=self.dev)
        self._lam_next = torch.zeros(1, dtype=torch.float64, device=self.dev)
        self._lam_valid = False                                       # _lam holds the scan of the CURRENT states (left there by the last fused step)
        self._grid_arr = larr(list(self.grid))

    # -- the reference layout (with halo) ----------------------------------------------------------
    def _inner(self):
        H, P = self.H, self.P
        return (slice(None),) * self.dim + (slice(H, H + P),) * self.dim

    @property
    def Q(self):
        """two-pass form: the array with halo itself, [g.., S.., V] (halo layers as the last fill left them).  The fused form has no such array
        (writing into a copy would be lost without a word): `with_halo()` builds one, `U` is where the states live."""
        if not self.fused:
            return self._Q
        raise AttributeError("FVPatchGrid keeps its states halo-less in `.U` [g.., P.., V]; `.with_halo()` returns a copy in the layout with halo "
                             "(`fused=False` keeps the array with halo as `.Q`)")

    def with_halo(self):
        """A NEW array [g.., S.., V] in the reference layout (S = P + 2 H): interiors + the halo layers filled from the neighbours' boundary layers /
        the boundary states -- what `time_step` expects of its caller."""
        if not self.fused:
            self.fill_halos()
            return self._Q.clone()
        torch = _torch()
        S = self.P + 2 * self.H
        Q = torch.zeros(self.grid + (S,) * self.dim + (self.n_real + self.n_aux,), dtype=torch.float64, device=self.dev)
        Q[self._inner()] = self.U
        if self.boundary is None:
            fill_halos_periodic(Q, self.grid, self.dim, self.P, self.H)
        else:
            fill_halos_dirichlet(Q, self.grid, self.dim, self.P, self.H, self.boundary)
        return Q

    def fill_halos(self):
        """two-pass form only: the halo layers of the array with halo from the neighbours' interiors / the boundary states (torch ops)"""
        if self.fused:
            return
        if self.boundary is None:
            fill_halos_periodic(self._Q, self.grid, self.dim, self.P, self.H)
        else:
            fill_halos_dirichlet(self._Q, self.grid, self.dim, self.P, self.H, self.boundary)

    def set_interior(self, values):
        """values: [g.., P.., V] (numpy or tensor)."""
        torch = _torch()
        self._lam_valid = False
        v = torch.as_tensor(np.asarray(values) if not isinstance(values, torch.Tensor) else values, dtype=torch.float64).to(self.dev)
        if self.fused:
            self.U.copy_(v.reshape(self.U.shape))
        else:
            self._Q[self._inner()] = v

    def interior_device(self):
        return self.U if self.fused else self._Q[self._inner()]

    def interior(self):
        return self.interior_device().cpu().numpy()

    def max_eigenvalue_device(self):
        """Largest eigenvalue over the (interior) volumes and the directions at the current time: a 1-element CUDA tensor, no host
        synchronisation.  After a fused step() it is already there -- the patch kernel reduced the eigenvalues of the states it wrote --
        otherwise one reduction launch (`exa_fv_max_eigenvalue`).  (Whoever writes `self.U` directly calls invalidate().)"""
        if not self._lam_valid:
            arr = self.U if self.fused else self._Q
            check(self.lib.exa_fv_max_eigenvalue(self.kernel._plan, C.c_void_p(arr.data_ptr()), 1 if self.fused else 0,
                                                 C.c_void_p(self.centres.data_ptr()), self.time, self.h, C.c_void_p(self._lam.data_ptr()), _stream_ptr()))
            if self._bstate is not None:
                self._lam.clamp_(min=self._lam_boundary)
            self._lam_valid = True
        return self._lam

    def invalidate(self):
        """the states were written from outside: the next CFL scan reads the array again"""
        self._lam_valid = False

    def max_eigenvalue(self):
        return float(self.max_eigenvalue_device()[0])

    def step(self, dt):
        if not self.fused:
            self.fill_halos()
            self.kernel.time_step(self._Q, dt, self.h, t=self.time, centres=self.centres)
            self.time += dt
            self._lam_valid = False
            return
        torch = _torch()
        if self._U2 is None:
            self._U2 = torch.empty_like(self.U)
        check(self.lib.exa_fv_grid_step_device(self.kernel._plan, C.c_void_p(self.U.data_ptr()), C.c_void_p(self._U2.data_ptr()), self._grid_arr,
                                               C.c_void_p(self._bstate.data_ptr()) if self._bstate is not None else None,
                                               C.c_void_p(self.centres.data_ptr()), self.time, dt, self.h,
                                               C.c_void_p(self._lam_next.data_ptr()), _stream_ptr()))
        self.U, self._U2 = self._U2, self.U
        self._lam, self._lam_next = self._lam_next, self._lam         # the scan of the new states, by the kernel that wrote them
        if self._bstate is not None:
            self._lam.clamp_(min=self._lam_boundary)
        self._lam_valid = True
        self.time += dt

    def run(self, t_end, cfl=0.4, max_steps=1000000):
        """Advance until `self.time` reaches t_end (AderDgSolver.run means the same) with the CFL step cfl * h / (dim * lambda_max); returns the number of steps.
        A non-finite lambda_max (a state that has blown up: negative pressure gives a NaN eigenvalue) raises instead of ending the loop silently."""
        steps = 0
        while self.time < t_end * (1 - 1e-14) and steps < max_steps:
            lam = self.max_eigenvalue()                                                         # (the step's one host read)
            dt = _cfl_step(lam, cfl * self.h / self.dim, t_end - self.time, "FVPatchGrid.run")
            self.step(dt)
            steps += 1
        return steps


def _cfl_step(lam, scale, left, who):
    """dt = min(scale / lambda_max, time left); lambda_max == 0.0 exactly (nothing moves): one step to the end; NaN / inf: the run has diverged."""
    if not math.isfinite(lam) or lam < 0.0:
        raise FloatingPointError("%s: the largest eigenvalue is %r -- the solution has left the admissible states (diverged run)" % (who, lam))
    return min(scale / lam, left) if lam > 0.0 else left


def _dg_run(self, t_end, cfl=0.4, max_steps=1000000):
    """Advance until `self.time` reaches t_end (as FVPatchGrid.run; r4 integrated a DURATION from a local t = 0 here) with the CFL step
    dt = cfl * min(dx) / ((2p+1) * d * lambda_max); with a partition the maximum eigenvalue is reduced over the ranks (the only true collective
    of the scheme).  Returns the number of steps; raises on a non-finite eigenvalue."""
    torch = _torch()
    steps, lam = 0, None
    fuse = self.can_fuse_cfl_scan()        # r5: the scan of step n + 1 rides in stage B of step n (one pass over u per step less); the first step scans
    try:
        if fuse:
            self._cfl_out = torch.zeros(1, dtype=torch.float64, device=self.dev)
        while self.time < t_end * (1 - 1e-14) and steps < max_steps:
            if lam is None:
                lam = self.max_eigenvalue()
            if self.part is not None and self.part.world > 1:
                import torch.distributed as dist
                dist.all_reduce(lam, op=dist.ReduceOp.MAX)
            dt = _cfl_step(float(lam[0]), cfl * min(self.dx) / ((2 * self.N - 1) * self.dim), t_end - self.time, "AderDgSolver.run")
            self.step(dt)
            lam = self._cfl_out if fuse else None
            steps += 1
    finally:
        self._cfl_out = None
    return steps


AderDgSolver.run = _dg_run


# ----------------------------------------------------------------------------------------------
# FV subcell limiter (BASELINE configs[4])
# ----------------------------------------------------------------------------------------------
class SubcellLimiter:
    """Limited ADER-DG step: untroubled cells take the DG step, troubled cells the FV Rusanov patch update
    (patch_size 2p+1, halo 1 -- the reference's kernel shape) of their projected data.  The troubled mask is an
    input (synthetic Bernoulli mask in the benchmark; a physical detector is host logic outside this path).

    On a sharded grid (solver.part) two small exchanges precede the projection: the troubled flags of the blocks'
    boundary layers, then -- only where the cell across the face is troubled -- the adjacent subcell layer of the
    boundary cells; the patches of troubled cells at a block face take their halo from those."""

    DEFAULT_FRACTION = 0.1          # default capacity: this share of the block's cells (at least 16)

    def __init__(self, solver, capacity=None):
        """capacity: upper bound of the number of troubled cells per step; the FV patch array [capacity][(N_s+2)^dim][n_vars]
        is allocated once (196 KB per patch at p = 7) and the glue kernels launch `capacity` workgroups per step, so the
        default is a bounded share of the block -- max(16, 10 % of its cells) -- not the whole block (262 144 cells at p = 7
        would be 51 GB).  A step with more troubled cells than that is reported by check(): construct the limiter again with a
        larger capacity (capacity=n_cells serves every mask).  Raises if the patch array does not fit the free device memory."""
        torch = _torch()
        # term sets whose terms depend on position / time: the patch update of a troubled cell gets the cell's centre and the step's start time
        self._xt = bool(solver.lib.exa_pde_flags(int(solver.pde)) & 1)
        self.s = solver
        self.exchange_events = None       # measurement hook (bench.py): one (start, end) event pair per step around the limiter's own two exchanges
        self.Ns = 2 * solver.N - 1
        self.patch_doubles = solver.lib.exa_lim_patch_count(solver._plan)
        ncell = int(np.prod(solver.nc))
        if capacity is None:
            capacity = max(16, int(self.DEFAULT_FRACTION * ncell))
        self.capacity = max(1, min(int(capacity), ncell))
        need = self.capacity * self.patch_doubles * 8
        free = torch.cuda.mem_get_info(solver.dev)[0]
        if need > free:
            raise MemoryError("SubcellLimiter: capacity %d x %d B per patch = %.1f GB, %.1f GB of device memory are free; "
                              "pass a smaller capacity" % (self.capacity, self.patch_doubles * 8, need / 1e9, free / 1e9))
        self._patches = torch.empty((self.capacity, self.patch_doubles), dtype=torch.float64, device=solver.dev)
        self._cells = torch.full((self.capacity + 1,), -1, dtype=torch.int64, device=solver.dev)     # + dump slot
        self._arange = torch.arange(ncell, dtype=torch.int64, device=solver.dev)
        self._fv = FVRusanovKernel(solver.dim, self.Ns, 1, solver.nv, 0, self.capacity, pde=solver.pde, mode=FV_RUSANOV,
                                   device=solver.dev.index or 0)
        self.overflow = torch.zeros((), dtype=torch.bool, device=solver.dev)
        self._ovf_host = torch.zeros(1, dtype=torch.uint8).pin_memory()
        self._ovf_event = torch.cuda.Event()
        self._ovf_event.record(torch.cuda.current_stream(solver.dev))
        self.hx_mask = self.hx_layer = None
        if solver.halo is not None:
            stage = solver.halo.stage
            self.hx_mask = HaloExchange(solver.part, solver.nc, 1, solver.dev, stage_through_host=stage)
            self.hx_layer = HaloExchange(solver.part, solver.nc, self.Ns ** (solver.dim - 1) * solver.nv, solver.dev,
                                         stage_through_host=stage)

    def operators(self):
        N, Ns = self.s.N, self.Ns
        P, R = np.zeros((Ns, N)), np.zeros((N, Ns))
        check(self.s.lib.exa_lim_operators(self.s._plan, P.ctypes.data_as(C.c_void_p), R.ctypes.data_as(C.c_void_p)))
        return P, R

    def detect(self, dmp_tol=0.5, floor=1e-12):
        """A-priori troubled-cell indicator on the current state (positivity + a relaxed discrete maximum
        principle): a cell is troubled if density or pressure is non-positive / not finite at a node, or if the nodal
        density leaves the range of the cell means of its face neighbourhood by more than dmp_tol times that range.
        Euler variables (rho, m, E) with gamma = 1.4.  Returns a bool tensor [nc0, nc1, (nc2)] on the device.
        Host-level logic in torch ops -- the kernels take the mask as an input."""
        torch = _torch()
        s = self.s
        dim = s.dim
        u = s.u
        nodes = tuple(range(dim, 2 * dim))
        rho = u[..., 0]
        ke = sum(u[..., 1 + a] ** 2 for a in range(min(3, s.nv - 2)))
        p = 0.4 * (u[..., s.nv - 1] - 0.5 * ke / rho)
        bad = (rho.amin(nodes) <= floor) | (p.amin(nodes) <= floor) | ~torch.isfinite(u).all(-1).flatten(dim).all(-1)
        w = torch.as_tensor(s.operators()["w"], device=u.device)
        mean = rho
        for _ in range(dim):
            mean = torch.tensordot(mean, w, dims=([dim], [0]))     # contracts the first remaining node axis
        lo, hi = mean.clone(), mean.clone()
        ghost = None
        if s.halo is not None:                                   # neighbour means across block faces
            hx = self.hx_mask
            nc3 = s.nc + [1] * (3 - dim)
            m3 = mean.reshape(nc3)
            for d in range(dim):
                if s.part.partitioned(d):
                    hx.send[d * 2 + 0].copy_(m3.select(d, 0).reshape(-1, 1))
                    hx.send[d * 2 + 1].copy_(m3.select(d, nc3[d] - 1).reshape(-1, 1))
            hx.start()
            hx.finish()
            ghost = hx.ghost
        for d in range(dim):
            up, dn = mean.roll(-1, d), mean.roll(1, d)
            if ghost is not None and s.part.partitioned(d):
                other = [s.nc[a] for a in range(dim) if a != d]
                up.select(d, s.nc[d] - 1).copy_(ghost[d * 2 + 1].reshape(other))
                dn.select(d, 0).copy_(ghost[d * 2 + 0].reshape(other))
            lo = torch.minimum(lo, torch.minimum(up, dn))
            hi = torch.maximum(hi, torch.maximum(up, dn))
        span = (hi - lo).clamp_min(floor)
        bad |= (rho.amax(nodes) > hi + dmp_tol * span) | (rho.amin(nodes) < lo - dmp_tol * span)
        return bad

    def _exchange_subcell_layers(self, m):
        """m: troubled flags of the block, float64 [nc0, nc1, nc2] on the device.  Returns the ghost-layer pointer array."""
        s, hm, hl = self.s, self.hx_mask, self.hx_layer
        nc3 = s.nc + [1] * (3 - s.dim)
        for d in range(s.dim):
            if s.part.partitioned(d):
                hm.send[d * 2 + 0].copy_(m.select(d, 0).reshape(-1, 1))
                hm.send[d * 2 + 1].copy_(m.select(d, nc3[d] - 1).reshape(-1, 1))
        hm.start()
        hm.finish()
        for d in range(s.dim):
            if not s.part.partitioned(d):
                continue
            for side in range(2):
                # ghost[d*2+side] = flags of the cells across my face (d, side): where set, they need my layer
                check(s.lib.exa_lim_face_layers(s._plan, C.c_void_p(s.u.data_ptr()), d, side,
                                                C.c_void_p(hm.ghost[d * 2 + side].data_ptr()),
                                                C.c_void_p(hl.send[d * 2 + side].data_ptr()), _stream_ptr()))
        hl.start()
        hl.finish()
        return hl.ghost_ptrs()

    def step(self, dt, mask):
        """One limited step.  mask: troubled flags [nc0, nc1, (nc2)] (CUDA bool tensor stays on the device; numpy is
        uploaded).  Nothing in here waits for the GPU: the troubled cells are compacted on the device into a
        capacity-sized list (empty slots = -1) that the projection, the FV patch update and the reconstruction
        skip, and one FV plan of `capacity` patches serves every step.  Returns the number of troubled cells as a
        0-dim CUDA tensor (int(...) of it synchronises; the kernels do not need it).  More troubled cells than
        `capacity` cannot be served: `self.overflow` (0-dim CUDA bool) says so -- see check()."""
        torch = _torch()
        s = self.s
        # the FV patch update takes ONE volume size h (the reference's generated `time_step` has no cell size at all: SURVEY.md Appendix B): a grid with
        # different cell sizes per axis would get a silently wrong update -- refused (as in oracle/limiter_numpy.py, which restates this glue)
        if max(s.dx) - min(s.dx) > 1e-12 * max(s.dx):
            raise ValueError("SubcellLimiter.step: the FV patch update takes one volume size, the grid has dx = %s; use cells of equal size per axis" % (list(s.dx),))
        self.check()                                           # a COMPLETED earlier step past the capacity raises here (no synchronisation)
        if isinstance(mask, torch.Tensor):
            m = mask.to(device=s.dev, dtype=torch.bool)
        else:
            m = torch.as_tensor(np.ascontiguousarray(mask), dtype=torch.bool).to(s.dev, non_blocking=True)
        m = m.reshape(-1)
        ncell, cap = m.numel(), self.capacity
        # device-side compaction without a host-visible size: cell c goes to slot (number of troubled cells before it)
        rank = torch.cumsum(m, 0, dtype=torch.int64)
        count = rank[-1]
        pos = torch.where(m & (rank <= cap), rank - 1, torch.full_like(rank, cap))       # cap = dump slot
        self._cells.fill_(-1)
        self._cells.scatter_(0, pos, self._arange)
        self._cells[cap] = -1
        self.overflow |= count > cap                           # sticky until check() has reported it
        self._post_overflow()
        cells = self._cells
        ghosts = None
        if self.hx_layer is not None:                          # every rank takes part, troubled cells or not
            nc3 = s.nc + [1] * (3 - s.dim)
            if self.exchange_events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(torch.cuda.current_stream(s.dev))
            ghosts = self._exchange_subcell_layers(m.to(torch.float64).reshape(nc3))
            if self.exchange_events is not None:
                e1.record(torch.cuda.current_stream(s.dev))
                self.exchange_events.append((e0, e1))
        patches = self._patches
        check(s.lib.exa_dg_project_patches_ghost(s._plan, C.c_void_p(s.u.data_ptr()), C.c_void_p(cells.data_ptr()), cap,
                                                 C.c_void_p(patches.data_ptr()), ghosts, _stream_ptr()))
        t0 = s.time
        s.step(dt)                                             # candidate DG solution everywhere
        if self._xt:
            # centre of the patch in slot k = centre of its DG cell (device ops on the compacted list; unused slots: any value)
            c = cells[:cap].clamp(min=0)
            cen = torch.empty((cap, s.dim), dtype=torch.float64, device=s.dev)
            for a in range(s.dim - 1, -1, -1):
                cen[:, a] = s.origin[a] + ((c % s.nc[a]).to(torch.float64) + 0.5) * s.dx[a]
                c = c // s.nc[a]
            self._fv.time_step(patches.reshape(-1), dt, s.dx[0] / self.Ns, slot=cells, t=t0, centres=cen)
        else:
            self._fv.time_step(patches.reshape(-1), dt, s.dx[0] / self.Ns, slot=cells)
        check(s.lib.exa_dg_reconstruct_patches(s._plan, C.c_void_p(patches.data_ptr()), C.c_void_p(cells.data_ptr()), cap,
                                               C.c_void_p(s.u.data_ptr()), _stream_ptr()))
        return count

    def download(self):
        """The solver's u on the host -- after a WAITING check(): a result in which some troubled cells kept the unlimited DG solution is
        not handed out silently."""
        self.check(wait=True)
        return self.s.download()

    def _post_overflow(self):
        """Copy the overflow flag to pinned host memory behind an event; check() reads it once the event has passed."""
        torch = _torch()
        self._ovf_host.copy_(self.overflow.to(torch.uint8), non_blocking=True)
        self._ovf_event.record(torch.cuda.current_stream(self.s.dev))

    def check(self, wait=False):
        """Raise if ANY step since the last report had more troubled cells than `capacity` (those beyond it kept the DG
        result): the flag accumulates on the device and is cleared only here.  Without `wait` only steps the GPU has
        already completed are looked at (no synchronisation)."""
        if wait:
            self._ovf_event.synchronize()
        if self._ovf_event.query() and bool(self._ovf_host[0]):
            self._ovf_event.synchronize()                                                   # (the latest copy into the pinned byte has landed)
            self.overflow = _torch().zeros((), dtype=_torch().bool, device=self.s.dev)      # reported: start over
            self._ovf_host[0] = 0
            raise RuntimeError("SubcellLimiter: a step since the last check() had more troubled cells than capacity = %d "
                               "(those beyond it kept the unlimited DG result)" % self.capacity)
