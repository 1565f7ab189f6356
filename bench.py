#!/usr/bin/env python3
"""Headline benchmark: DoF-updates/s of the 3-D Euler p=5 ADER-DG step
(space-time predictor Picard loop + volume integral + face Riemann solve +
corrector) on 1/2/4/8 MI355X.

    python bench.py --gpus N --steps K --warmup W [--config cfg2|cfg1|cfg4|fv-ref]

--config cfg2 (default, the configuration BASELINE.json's metric is quoted on):
  N = 1 runs BASELINE.json configs[2] (128^3 cells, one GPU).  N > 1 keeps 128^3 cells per GPU
  (weak scaling; N = 8 is configs[3], 256^3 cells) on a Cartesian process grid with the RCCL
  face-trace halo exchange overlapped with the interior predictor work.  `python bench.py --gpus N`
  starts its N ranks itself (one process per GPU, before the parent touches a GPU); launched under
  `torch.distributed.run` it uses the ranks it is given.
--config cfg1 / cfg4 / fv-ref (one GPU each): configs[1] (2-D p=3, 512^2, single-stage
  step), configs[4]'s per-GPU shape (p=7, 64^3 cells + FV subcell limiter, 5 % troubled), and the
  reference-native FV Rusanov patch update (2-D, P=4, H=1, 5+5 variables, 2^20 patches).

Synthetic data: smooth Euler density wave + seeded 1e-3 noise, fixed dt.  One "step" =
one full time step of every cell.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6     # MI355X datasheet: fp64 vector peak (== the fp64 matrix peak); DESIGN.md "Roofs"
FP64_MEASURED_TFLOPS = 66.8 # bare v_fma_f64 loop on this chip (scripts/fp64_peak.hip; 63.2 at the 3 waves per SIMD stage A runs with:
                            # profiles/r01_stage_a_stamps.txt) -- reported beside the datasheet fraction, never instead of it
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_MEASURED_GBS = 6290.0   # ... and its measured float4-copy rate


# ------------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` without a launcher
# ------------------------------------------------------------------------------------------------------
def spawn_ranks(n, argv):
    """Start n worker processes (one rank each) and relay rank 0's output.  The parent never initialises a GPU (no
    torch.cuda call, no HIP call): it only starts children and waits -- nothing is exec'ed over a GPU process."""
    import socket
    from exahype_amd import build as exa_build
    exa_build.build()                      # hipcc only, no GPU call: the ranks must not race to compile the same library
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      stderr=None, text=True))
    out0 = procs[0].communicate()[0]
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out0)
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write("bench.py: rank(s) failed: %s\n" % bad)
        sys.exit(1)
    sys.exit(0)


# ------------------------------------------------------------------------------------------------------
# synthetic inputs
# ------------------------------------------------------------------------------------------------------
def synthetic_state(solver, part_coords, pdims, seed):
    """Density wave rho = 1 + 0.2 sin(2 pi sum x), u = (1,1,1), p = 1, times 1 + 1e-3 U(-1,1); built on the device."""
    import torch
    s = solver
    dim, N, nc = s.dim, s.N, s.nc
    xi = torch.as_tensor(s.operators()["xi"], device=s.dev)
    phase = torch.zeros(s.u.shape[:-1], dtype=torch.float64, device=s.dev)
    for a in range(dim):
        ncg = nc[a] * pdims[a]
        cells = torch.arange(nc[a], device=s.dev, dtype=torch.float64) + part_coords[a] * nc[a]
        x = (cells[:, None] + xi[None, :]) / ncg
        sh = [1] * (2 * dim)
        sh[a], sh[dim + a] = nc[a], N
        phase = phase + x.reshape(sh)
    g = torch.Generator(device=s.dev)
    g.manual_seed(seed)
    rho = 1.0 + 0.2 * torch.sin(2 * torch.pi * phase)
    E = 1.0 / 0.4 + 0.5 * rho * dim
    for v, val in enumerate((rho, rho, rho, rho if dim == 3 else torch.zeros_like(rho), E)):
        noise = 1.0 + 1e-3 * (2 * torch.rand(val.shape, generator=g, device=s.dev, dtype=torch.float64) - 1)
        s.u[..., v] = val * noise
    del phase, rho, E
    return 1.0 + (1.4 * 1.0 / 0.8) ** 0.5          # |u_d| + c_s upper bound per direction


# ------------------------------------------------------------------------------------------------------
# the same Euler system as a USER would hand it over: SymPy expressions (north_star: "a SymPy-specified flux/eigenvalue system drops in")
# ------------------------------------------------------------------------------------------------------
def sympy_euler():
    """3-D compressible Euler (gamma = 1.4) written as plain SymPy expressions -- the arithmetic of the reference's `Flux` / `maxEigenvalue`
    (Unit test/Functions.cpp:9-62) without any hint about caching, reciprocals or directions; pde_codegen.SympyPDE turns it into the device
    term set (`other_configs.cfg2_sympy` / `cfg4_sympy` run the benchmark configurations on it)."""
    import sympy
    from exahype_amd.pde_codegen import SympyPDE

    def pressure(q):
        return sympy.Float(0.4) * (q[4] - (q[1] ** 2 + q[2] ** 2 + q[3] ** 2) / (2 * q[0]))

    def flux(q, d):
        p, un = pressure(q), q[d + 1] / q[0]
        f = [un * q[0], un * q[1], un * q[2], un * q[3], un * (q[4] + p)]
        f[d + 1] += p
        return f

    def max_eigenvalue(q, d):
        return sympy.Abs(q[d + 1] / q[0]) + sympy.sqrt(sympy.Float(1.4) * pressure(q) / q[0])
    return SympyPDE(5, flux, max_eigenvalue, max_dim=3, name="euler_sympy_bench")


# ------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle, timed on the GPU box's host cores; reported, never the target)
# ------------------------------------------------------------------------------------------------------
def host_cpu_info():
    """(cpus this process may use, physical cores of the machine, model name)."""
    model, phys = "unknown", set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    try:                                                     # a cgroup CPU quota bounds what the threads really get
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            usable = max(1, min(usable, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return usable, (len(phys) or (os.cpu_count() or 1)), model


def cpu_threads():
    usable, phys, _ = host_cpu_info()
    return int(os.environ.get("OMP_NUM_THREADS", max(1, min(usable, phys))))


def cpu_baseline_dg(dim, N, n_it, seconds=10.0, samples=3):
    """The oracle (oracle/exa_oracle.c, OpenMP over cells) on the host cores: the same ADER-DG step on a bounded block,
    >= 16 cells per thread, work arrays allocated once outside the timed steps.  `value` is the MEDIAN of `samples` samples of
    `seconds` each (one sample moved by +-25 % between runs on the shared host)."""
    import numpy as np
    threads = cpu_threads()
    import oracle
    from oracle.dg_operators import operators
    from tests.util import euler_dg_state
    oracle.lib().orc_set_threads(threads)
    threads = oracle.lib().orc_get_max_threads()
    n = 4
    while n ** dim < 16 * threads:
        n += 2
    nc = (n,) * dim
    ops = operators(N)
    u = np.ascontiguousarray(euler_dg_state(nc + (N,) * dim, seed=2, amp=0.1)).reshape(-1)
    dx = [1.0 / c for c in nc]
    dt = 1e-4 / N
    for _ in range(2):                                                                # warm-up (pages touched, team started)
        t0 = time.perf_counter()
        oracle.aderdg_run(u, dt, dx, ops, dim, N, 5, oracle.PDE_EULER, n_it, nc, 1)
        one = time.perf_counter() - t0
    chunk = int(max(1, min(500, 1.0 / max(one, 1e-6))))                                # ~1 s of steps per call, arrays allocated once per call
    dof = int(np.prod(nc)) * N ** dim * 5
    rates, total_steps, total_el = [], 0, 0.0
    for _ in range(samples):
        steps, t0 = 0, time.perf_counter()
        while True:
            oracle.aderdg_run(u, dt, dx, ops, dim, N, 5, oracle.PDE_EULER, n_it, nc, chunk)
            steps += chunk
            el = time.perf_counter() - t0
            if el >= seconds:
                break
        rates.append(dof * steps / el)
        total_steps += steps
        total_el += el
    usable, phys, model = host_cpu_info()
    return {"value": sorted(rates)[len(rates) // 2], "unit": "DoF-updates/s", "cores": threads, "kind": "port",
            "sample": "median of %d samples (%s DoF-updates/s), %d steps in all of a %s-cell block (%d cells per thread), same p=%d Euler "
                      "ADER-DG step (%d Picard iterations), oracle/exa_oracle.c (gcc -O3 -fopenmp), %.1f s"
                      % (samples, " / ".join("%.3g" % r for r in rates), total_steps, "x".join(map(str, nc)),
                         int(np.prod(nc)) // threads, N - 1, n_it, total_el),
            "host": {"cpu_model": model, "physical_cores": phys, "usable_cpus": usable, "nproc": os.cpu_count()}}


def cpu_reference_fv(seconds=6.0):
    """Where the reference is today: its generated `time_step` (Unit test/test.cpp, compiled where it lies
    into oracle/_ref) on ONE core, 2-D 4x4 patch, 5+5 variables, back-to-back calls over independent patches -- or,
    when oracle/_ref did not travel, the statement-for-statement restatement (oracle.fv_faithful)."""
    import numpy as np
    import oracle
    n = 1 << 14
    Q = np.ascontiguousarray(np.tile(2.0 + np.sin(3.141 * np.arange(360) / 360), (n, 1)))
    R = oracle.ref()
    if R is not None:
        def call():
            R.ref_time_step_batched(Q.ravel(), 1e-3, n, 360)
        kind, what = "reference", "Unit test/test.cpp + Functions.cpp (g++ -O2, oracle/_ref)"
    else:
        oracle.lib().orc_set_threads(1)

        def call():
            oracle.lib().orc_fv_rusanov_faithful(Q.ravel(), 1e-3, 2, 4, 1, 5, 5, n, oracle.PDE_EULER_REF2D)
        kind, what = "port", "oracle/exa_oracle.c faithful restatement (gcc -O3), 1 thread"
    call()
    t0, reps = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        call()
        reps += 1
    el = time.perf_counter() - t0
    return {"value": n * reps * 16 * 5 / el, "unit": "DoF-updates/s", "volume_updates_per_s": n * reps * 16 / el, "cores": 1,
            "kind": kind, "sample": "%d calls of time_step on 4x4 patches (5+5 variables), %s, %.1f s" % (n * reps, what, el)}


def read_traffic(name, **match):
    """HBM bytes per launch of the dominant kernel from the PMC passes (scripts/pmc_traffic.sh writes profiles/<name>;
    FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE).  Not measured inside this run: `traffic_source` says where from."""
    tf = os.path.join(ROOT, "profiles", name)
    if os.path.exists(tf):
        rec = json.load(open(tf))
        if all(rec.get(k) == v for k, v in match.items()):
            return rec.get("hbm_bytes_per_launch"), "profiles/" + name + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2)"
    return None, None


# ------------------------------------------------------------------------------------------------------
# fields shared by the sharded lines (cfg 2 / 3 and cfg 4 on N ranks)
# ------------------------------------------------------------------------------------------------------
def sharded_roofline(a, torch, dist, s, work, ta, world, bound, extra=None):
    """Per-rank roofline of an N > 1 line: every rank sums its stage-A launches per step (shell boxes + interior box, events on the launching
    stream); the SLOWEST rank's sum prices the per-GPU algorithmic flops of `exa_dg_work` -- `frac` is what one GPU of the job achieves."""
    t = torch.tensor([ta], dtype=torch.float64, device="cuda" if a.backend == "nccl" else "cpu")
    allt = [torch.zeros_like(t) for _ in range(world)]
    if world > 1:
        dist.all_gather(allt, t)
    else:
        allt = [t]
    per_rank = [float(v[0]) for v in allt]
    slow = max(per_rank)
    ach = work["flop_a"] / slow / 1e12
    rf = {"kernel": s.stage_a_kernel_name(), "bound": bound, "bound_basis": "algorithmic flops per GPU (SURVEY.md 8(d)) over the slowest rank's summed "
          "stage-A launches of a step (shell boxes + interior box)", "achieved": ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
          "frac": ach / FP64_PEAK_TFLOPS, "frac_of_measured_fma_loop": ach / FP64_MEASURED_TFLOPS, "traffic": None,
          "traffic_source": "not collected on N > 1 (the single-GPU line of the same kernel carries it)", "launch_ms": slow * 1e3,
          "launch_ms_per_rank": [x * 1e3 for x in per_rank], "launches_per_step": len(s.stage_a_events) // max(1, a.steps),
          "flop_per_launch": work["flop_a"], "scope": "per rank (slowest)"}
    if extra:
        rf.update(extra)
    return rf


def exchange_fields(a, torch, dist, s, world, local, ta, selfx, reserve_trial):
    """exchange_ms / pack_ms / overlap_frac / exposed_exchange_ms ... of a sharded run: the trace exchange (pack + RCCL send/recv) on the comm
    stream against the interior stage A on the compute stream, worst rank each."""
    out = {}
    ex, ov, xp, pk = [], [], [], []
    for ready, c0, c1, i0, i1, p1 in s.exchange_events:
        cs, ce, ps = ready.elapsed_time(c0), ready.elapsed_time(c1), ready.elapsed_time(p1)
        is_, ie = ready.elapsed_time(i0), ready.elapsed_time(i1)
        ex.append(ce - ps)                         # the RCCL send / recv group alone (start() .. finish())
        pk.append(ps - cs)                         # the pack copies in front of it
        ov.append(max(0.0, min(ce, ie) - max(cs, is_)) / max(ce - cs, 1e-9))
        xp.append(max(0.0, ce - ie))               # what the exchange adds to the step: its end past the end of the interior stage A
    mine = torch.tensor([sum(ex) / len(ex), sum(ov) / len(ov), ta * 1e3, sum(xp) / len(xp), sum(pk) / len(pk)], dtype=torch.float64,
                        device="cuda" if a.backend == "nccl" else "cpu")
    allv = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    names = [None] * world
    dist.all_gather_object(names, "%s:%d" % (torch.cuda.get_device_name(local), local))
    if selfx:
        out["rehearsal"] = "sharded step on one GPU: shell / interior boxes, packed faces, ncclSend / ncclRecv to self, stage B on ghost buffers"
    out["rccl_ranks"] = dist.get_world_size()
    out["backend"] = dist.get_backend()
    out["devices"] = names
    out["exchange_ms"] = max(float(v[0]) for v in allv)              # RCCL span only; the pack copies are `pack_ms`
    out["pack_ms"] = max(float(v[4]) for v in allv)
    out["high_priority_comm"] = {"comm_stream": True, "nccl_process_group_stream": bool(a.backend == "nccl" and os.environ.get("EXA_NCCL_STREAM_PRIORITY", "normal") == "high")}
    if reserve_trial is not None:
        out["reserve_cus_chosen"] = s.reserve_cus
        out["reserve_cus_trial"] = {str(k): v for k, v in reserve_trial.items()}
    out["overlap_frac"] = min(float(v[1]) for v in allv)
    out["stage_a_ms"] = max(float(v[2]) for v in allv)
    out["exposed_exchange_ms"] = max(float(v[3]) for v in allv)
    return out


def reserve_cus_trial(a, torch, dist, s, step, sync, world):
    """CUs for RCCL's transport kernels: the interior stage A is a persistent grid that fills every CU, and a resident workgroup is not
    preempted by a higher-priority stream.  Two untimed steps each with the grid 0 and 8 workgroups short of the chip; the faster
    setting (max over ranks) is kept for the timed steps, both are reported."""
    trial = {}
    for k in (0, 8):
        s.set_reserve_cus(k)
        step()                                                   # (settle)
        s.exchange_events = []
        sync()
        t0 = time.perf_counter()
        for _ in range(2):
            step()
        sync()
        el_k = (time.perf_counter() - t0) / 2
        xp_k = max(max(0.0, ev[0].elapsed_time(ev[2]) - ev[0].elapsed_time(ev[4])) for ev in s.exchange_events)
        tk = torch.tensor([el_k, xp_k], dtype=torch.float64, device="cuda" if a.backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(tk, op=dist.ReduceOp.MAX)
        trial[k] = {"ms_per_step": 1e3 * float(tk[0]), "exposed_exchange_ms": float(tk[1])}
    chosen = min(trial, key=lambda k: trial[k]["ms_per_step"])
    s.set_reserve_cus(chosen)
    step()
    sync()
    return trial


# ------------------------------------------------------------------------------------------------------
# configurations
# ------------------------------------------------------------------------------------------------------
def run_cfg2(a, torch, exa, world, rank, local, pde=None):
    """BASELINE configs[2] / configs[3]: 3-D Euler p=5, full predictor + corrector, 128^3 cells per GPU.  pde: id of a registered (generated)
    term set instead of the built-in exa::Euler."""
    import torch.distributed as dist
    selfx = world == 1 and a.self_exchange           # rehearsal: the sharded step on one GPU, periodic wrap through RCCL send/recv to self
    part = exa.CartesianPartition(world, rank, 3, exchange_self=(0, 1, 2) if selfx else ()) if (world > 1 or selfx) else None
    sharded = part is not None
    pdims = part.pdims if part else [1, 1, 1]
    coords = part.coords if part else [0, 0, 0]
    N = a.order + 1
    nc = [a.cells] * 3
    dx = [1.0 / (nc[d] * pdims[d]) for d in range(3)]
    s = exa.AderDgSolver(3, N, nc, pde=exa.PDE_EULER if pde is None else pde, n_vars=5, n_picard=-1, dx=dx, device=local, part=part,
                         backend_is_gloo=(a.backend != "nccl"))
    lam = synthetic_state(s, coords, pdims, seed=2 + rank)
    dt = 0.1 * min(dx) / ((2 * a.order + 1) * 3 * lam)
    work = s.work()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        s.step(dt)
    sync()
    reserve_trial = None
    if sharded and not a.no_reserve_trial:
        reserve_trial = reserve_cus_trial(a, torch, dist, s, lambda: s.step(dt), sync, world)
    s.stage_a_events = []                  # stage-A launch durations: events on the stream the kernels are launched on
    if sharded:
        s.exchange_events = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        s.step(dt)
    sync()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device="cuda" if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    finite = bool(torch.isfinite(s.u).all().item())
    dof_per_gpu = nc[0] * nc[1] * nc[2] * N ** 3 * 5
    out = {
        "metric": "DoF-updates/sec, 3D Euler p=%d fused STP+volume+Riemann, 1/2/4/8 MI355X" % a.order,
        "value": dof_per_gpu * world * a.steps / el, "unit": "DoF-updates/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * el / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "3D compressible Euler, ADER-DG p=%d, %d^3 cells per GPU (%dx%dx%d process grid), "
                               "%d Picard iterations + volume + Riemann + corrector" % (a.order, a.cells, pdims[0], pdims[1], pdims[2], N),
                   "cells_per_gpu": a.cells ** 3, "order": a.order, "n_vars": 5, "dt": dt,
                   "parallelism": "cartesian-%dx%dx%d" % tuple(pdims),
                   "term_set": "built-in exa::Euler (exa_pde.hpp)" if pde is None else "generated from SymPy expressions (pde_codegen.SympyPDE)"},
        "finite": finite,
    }
    # stage A: per step one launch (single GPU) or the shell boxes + the interior box (sharded); sum per step
    per_step = len(s.stage_a_events) // a.steps
    ta = sum(e0.elapsed_time(e1) for e0, e1 in s.stage_a_events) / a.steps * 1e-3
    if world == 1:
        ach = work["flop_a"] / ta / 1e12
        kname = s.stage_a_kernel_name()
        traffic, src = read_traffic("stage_a_traffic.json", cells=a.cells, order=a.order, kernel=kname)
        pmc = {}
        pf = os.path.join(ROOT, "profiles", "stage_a_pmc.json")          # SQ counters of the same kernel (48^3 cells): not measured in this run
        if os.path.exists(pf) and a.order == 5:
            rec = json.load(open(pf))
            if rec.get("kernel") == kname:
                pmc = {"valu_busy": rec["valu_issue_frac"], "lds_busy": rec["lds_array_busy_frac"], "mfma_busy": 0.0,
                       "busy_source": "profiles/stage_a_pmc.json (rocprofv3 --pmc SQ_* passes at 48^3 cells)"}
        out["roofline"] = {"kernel": s.stage_a_kernel_name(), "bound": "fp64-valu", "bound_basis": "algorithmic flops (SURVEY.md 8(d))", "achieved": ach,
                           "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS,
                           "frac_of_measured_fma_loop": ach / FP64_MEASURED_TFLOPS, "traffic": traffic,
                           "traffic_source": src, "launch_ms": ta * 1e3, "launches_per_step": per_step, "flop_per_launch": work["flop_a"],
                           "hbm_achieved_gbs": work["bytes_a"] / ta / 1e9, "hbm_frac": work["bytes_a"] / ta / 1e9 / HBM_PEAK_GBS,
                           "hbm_frac_of_measured_copy": work["bytes_a"] / ta / 1e9 / HBM_MEASURED_GBS,
                           "hbm_measured_gbs": (traffic / ta / 1e9) if traffic else None,
                           "note": "fp64-compute-bound (48 FLOP/B): priced against the 78.6 TFLOP/s fp64 vector peak; the kernel "
                                   "issues no MFMA (fp64 MFMA has the same peak and fills 28 % of a tile at N = 6)"}
        out["roofline"].update(pmc)
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_dg(3, N, N)
            out["cpu_reference_fv"] = cpu_reference_fv()
    if sharded:
        if "roofline" not in out:                                   # N > 1: the slowest rank's stage A against the per-GPU work
            out["roofline"] = sharded_roofline(a, torch, dist, s, work, ta, world, "fp64-valu",
                                               {"hbm_achieved_gbs": None, "note": "fp64-compute-bound (48 FLOP/B): priced against the 78.6 TFLOP/s fp64 vector peak"})
        out.update(exchange_fields(a, torch, dist, s, world, local, ta, selfx, reserve_trial))
    return out


def run_cfg1(a, torch, exa, local):
    """BASELINE configs[1]: 2-D Euler p=3, 512 x 512 cells, volume + Riemann + corrector only (single stage), one fused launch."""
    N, nc = 4, (512, 512)
    s = exa.AderDgSolver(2, N, nc, n_picard=0, fused_single_stage=True, device=local)
    lam = synthetic_state(s, [0, 0, 0], [1, 1, 1], seed=1)
    dt = 0.1 * min(s.dx) / ((2 * 3 + 1) * 2 * lam)
    steps, warm = max(a.steps, 20), max(a.warmup, 5)
    for _ in range(warm):
        s.step(dt)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        s.step(dt)
    e1.record()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    tk = e0.elapsed_time(e1) / steps * 1e-3
    dof = nc[0] * nc[1] * N * N * 5
    b_alg = 8 * 5 * (2 * N ** 2 + 8 * 2 * N) * nc[0] * nc[1]            # 3 840 B per cell
    traffic, src = read_traffic("traffic_cfg1.json")
    out = {"metric": "DoF-updates/sec, 2D Euler p=3 volume+Riemann+corrector (single stage), 1 MI355X", "value": dof * steps / el,
           "unit": "DoF-updates/s", "n_gpus": 1, "steps": steps, "warmup": warm, "ms_per_step": 1e3 * el / steps, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "BASELINE configs[1]: 2D compressible Euler, ADER-DG p=3, 512x512 cells, volume + Riemann + corrector "
                                  "(n_picard = 0), fused single launch", "cells_per_gpu": nc[0] * nc[1], "order": 3, "n_vars": 5, "dt": dt},
           "finite": bool(torch.isfinite(s.u).all().item()),
           "roofline": {"kernel": "dg_fused_single_kernel<4,Euler>", "bound": "hbm", "bound_basis": "algorithmic bytes (SURVEY.md 8(d): 3 840 B per cell); "
                        "the fused kernel keeps the traces on chip, so its measured HBM rate (hbm_measured_gbs) is lower than `achieved`",
                        "hbm_measured_gbs": (traffic / tk / 1e9) if traffic else None, "achieved": b_alg / tk / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": b_alg / tk / 1e9 / HBM_PEAK_GBS, "frac_of_measured_copy": b_alg / tk / 1e9 / HBM_MEASURED_GBS,
                        # against the fused kernel's OWN compulsory traffic -- u read once, written once: 16 NV N^2 = 1 280 B per cell -- the kernel
                        # is at a quarter of the HBM roof: it is LDS- / issue-bound, not memory-bound
                        "compulsory_bytes_per_launch": 16 * 5 * N * N * nc[0] * nc[1],
                        "frac_of_compulsory": 16 * 5 * N * N * nc[0] * nc[1] / tk / 1e9 / HBM_PEAK_GBS,
                        "traffic": traffic, "traffic_source": src, "launch_ms": tk * 1e3, "bytes_per_launch": b_alg}}
    if not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_dg(2, N, 0, seconds=6.0)
    return out


def run_cfg4(a, torch, exa, world, rank, local, pde=None):
    """BASELINE configs[4]: 3-D Euler p=7, 64^3 cells per GPU, FV subcell limiter with a Bernoulli(0.05) troubled mask (seed 4 + rank):
    troubled cells take the 15^3 FV Rusanov patch update instead of the DG result.  One GPU: the per-GPU shape alone.  N > 1 (or
    --self-exchange): the block of every rank is a shard of a Cartesian process grid -- the sharded limited step (`SubcellLimiter` on a
    partitioned `AderDgSolver`: the trace exchange overlapped with the interior predictor as in cfg 3, plus the limiter's two small exchanges,
    troubled flags and the subcell layers of the cells across a face from a troubled one); weak scaling, same fields as cfg 3's line."""
    import torch.distributed as dist
    selfx = world == 1 and a.self_exchange
    part = exa.CartesianPartition(world, rank, 3, exchange_self=(0, 1, 2) if selfx else ()) if (world > 1 or selfx) else None
    sharded = part is not None
    pdims = part.pdims if part else [1, 1, 1]
    coords = part.coords if part else [0, 0, 0]
    N, n = 8, a.cells if a.cells != 128 else 64
    dx = [1.0 / (n * max(pdims))] * 3                      # cells of one size in every direction (the FV patch update has one volume size): a box, not a cube, on uneven process grids
    kw = {} if pde is None else {"pde": pde, "n_vars": 5}
    if sharded:
        kw.update(part=part, backend_is_gloo=(a.backend != "nccl"))
    s = exa.AderDgSolver(3, N, (n,) * 3, dx=dx, device=local, **kw)
    lam = synthetic_state(s, coords, pdims, seed=4 + rank)
    dt = 0.1 * min(s.dx) / ((2 * 7 + 1) * 3 * lam)
    g = torch.Generator(device=s.dev)
    g.manual_seed(4 + rank)
    mask = torch.rand((n,) * 3, generator=g, device=s.dev) < 0.05
    lim = exa.SubcellLimiter(s, capacity=int(0.08 * n ** 3) + 16)
    steps, warm = a.steps, max(1, a.warmup)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(warm):
        cnt = lim.step(dt, mask)
    sync()
    lim.check(wait=True)
    reserve_trial = None
    if sharded and not a.no_reserve_trial:
        reserve_trial = reserve_cus_trial(a, torch, dist, s, lambda: lim.step(dt, mask), sync, world)
    s.stage_a_events = []
    if sharded:
        s.exchange_events = []
        lim.exchange_events = []
    t0 = time.perf_counter()
    for _ in range(steps):
        lim.step(dt, mask)
    sync()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device="cuda" if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    lim.check(wait=True)                                                   # (a step past the capacity would have kept the unlimited DG result)
    ta = sum(e0.elapsed_time(e1) for e0, e1 in s.stage_a_events) / steps * 1e-3
    work = s.work()
    dof = n ** 3 * N ** 3 * 5
    out = {"metric": "DoF-updates/sec, 3D Euler p=7 ADER-DG + FV subcell limiter, %s MI355X" % ("1" if world == 1 else "1/2/4/8"),
           "value": dof * world * steps / el, "unit": "DoF-updates/s",
           "n_gpus": world, "steps": steps, "warmup": warm, "ms_per_step": 1e3 * el / steps, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "BASELINE configs[4] per-GPU shape: 3D compressible Euler, ADER-DG p=7, %d^3 cells%s, 8 Picard iterations + "
                                  "volume + Riemann + corrector, FV subcell limiter (15^3 patches) on a Bernoulli(0.05) troubled mask"
                                  % (n, " per GPU (%dx%dx%d process grid)" % tuple(pdims) if sharded else ""),
                      "cells_per_gpu": n ** 3, "order": 7, "n_vars": 5, "dt": dt, "troubled_cells": int(cnt),
                      "parallelism": "cartesian-%dx%dx%d" % tuple(pdims),
                      "term_set": "built-in exa::Euler (exa_pde.hpp)" if pde is None else "generated from SymPy expressions (pde_codegen.SympyPDE)"},
           "finite": bool(torch.isfinite(s.u).all().item())}
    busy, busy_src = 0.126, "profiles/r03_pmc_mfma_n8.txt"
    pf = os.path.join(ROOT, "profiles", "m8_pmc.json")                    # SQ_VALU_MFMA_BUSY_CYCLES / SIMD cycles of the current kernel (scripts/pmc_mfma_n8.sh)
    if os.path.exists(pf):
        rec = json.load(open(pf))
        if rec.get("kernel") == s.stage_a_kernel_name():
            busy, busy_src = rec["mfma_busy"], "profiles/m8_pmc.json (%s)" % rec.get("source", "rocprofv3 --pmc")
    m8 = {"mfma_busy": busy, "busy_source": busy_src + " (SQ_VALU_MFMA_BUSY_CYCLES / SIMD cycles, 32^3 cells; not measured in this run)"}
    bound = "fp64 (valu+mfma)"
    basis = ("algorithmic flops (SURVEY.md 8(d)); the derivative contraction runs on v_mfma_f64_4x4x4_4b_f64, everything else as vector instructions -- the "
             "matrix instruction uses the vector ALU's fp64 multipliers (profiles/r05_valu_lds_overlap.txt: the two do not overlap), one 78.6 TFLOP/s peak for both")
    if world == 1 and not sharded:
        traffic, src = read_traffic("traffic_cfg4.json", cells=n, kernel=s.stage_a_kernel_name())
        out["roofline"] = {"kernel": s.stage_a_kernel_name(), "bound": bound, "bound_basis": basis,
                           "hbm_measured_gbs": (traffic / ta / 1e9) if traffic else None, "achieved": work["flop_a"] / ta / 1e12,
                           "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": work["flop_a"] / ta / 1e12 / FP64_PEAK_TFLOPS,
                           "frac_of_measured_fma_loop": work["flop_a"] / ta / 1e12 / FP64_MEASURED_TFLOPS, "traffic": traffic,
                           "traffic_source": src, "launch_ms": ta * 1e3, "flop_per_launch": work["flop_a"],
                           "algorithmic_bytes_per_launch": work["bytes_a"]}
        out["roofline"].update(m8)
    else:
        out["roofline"] = sharded_roofline(a, torch, dist, s, work, ta, world, bound, dict(m8, bound_basis_kernel=basis))
        out.update(exchange_fields(a, torch, dist, s, world, local, ta, selfx, reserve_trial))
        # the limiter's own two exchanges (troubled flags, then subcell layers): stream time from the first pack to the last landing, worst rank
        lx = [e0.elapsed_time(e1) for e0, e1 in lim.exchange_events] or [0.0]
        t = torch.tensor([sum(lx) / len(lx)], dtype=torch.float64, device="cuda" if a.backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out["limiter_exchange_ms"] = float(t.item())
    if not a.no_cpu_baseline and rank == 0 and world == 1:
        out["cpu_baseline"] = cpu_baseline_dg(3, N, N, seconds=10.0)
    return out


def run_fv_ref(a, torch, exa, local):
    """SURVEY.md 8(d) FV-parity row, batched: the reference's own configuration (2-D, P=4, H=1, 5+5 variables; Batched_stateless.py:9),
    2^20 patches, faithful mode (bit-exact with the compiled reference)."""
    P, H, m, aux, n = 4, 1, 5, 5, 1 << 20
    S, V = P + 2 * H, m + aux
    k = exa.FVRusanovKernel(2, P, H, m, aux, n, exa.PDE_EULER_REF2D, exa.FV_FAITHFUL, device=local)
    g = torch.Generator(device="cuda")
    g.manual_seed(0)
    # every step updates a FRESH array (0.75 GB each; 23 of them fit easily in 288 GB): the reference's update has no dt/h
    # scaling and is not constant-preserving at the patch edge, so applying it repeatedly to the same
    # array leaves the admissible states within a few steps
    steps, warm = max(a.steps, 20), max(a.warmup, 3)
    Qs = torch.rand((steps + warm, n, S, S, V), generator=g, device="cuda", dtype=torch.float64)
    Qs[..., 0] += 1.0
    Qs[..., 3] += 3.0
    for w in range(warm):
        k.time_step(Qs[w], 1e-4, 0.1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for w in range(warm, warm + steps):
        k.time_step(Qs[w], 1e-4, 0.1)
    e1.record()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    tk = e0.elapsed_time(e1) / steps * 1e-3
    Q = Qs
    vols = n * P * P
    b_alg = vols * (8 * V * (S / P) ** 2 + 8 * m)                       # patch + halo read once, n_real written once (220 B per volume)
    traffic, src = read_traffic("traffic_fv_ref.json")
    out = {"metric": "DoF-updates/sec, FV Rusanov patch update (reference configuration 2D P=4 H=1 5+5 vars), 1 MI355X",
           "value": vols * m * steps / el, "unit": "DoF-updates/s", "volume_updates_per_s": vols * steps / el, "n_gpus": 1, "steps": steps,
           "warmup": warm, "ms_per_step": 1e3 * el / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
           "data": "synthetic",
           "config": {"workload": "reference-native FV Rusanov time_step (Unit test/test.cpp shape): 2D, patch 4x4, halo 1, 5+5 variables, "
                                  "2^20 patches, faithful mode (bit-exact with the compiled reference)", "patches": n},
           "finite": bool(torch.isfinite(Q).all().item()),
           "roofline": {"kernel": "fv_rusanov_kernel<2,EulerRef2D,faithful,staged,4x4>", "bound": "hbm", "bound_basis": "algorithmic bytes (patch + halo "
                        "read once, n_real written once: 220 B per volume)", "hbm_measured_gbs": (traffic / tk / 1e9) if traffic else None,
                        "achieved": b_alg / tk / 1e9,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b_alg / tk / 1e9 / HBM_PEAK_GBS,
                        "frac_of_measured_copy": b_alg / tk / 1e9 / HBM_MEASURED_GBS, "traffic": traffic, "traffic_source": src,
                        "launch_ms": tk * 1e3, "bytes_per_launch": b_alg}}
    if not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_reference_fv()
    return out


def run_fv_grid(a, torch, exa, local):
    """SURVEY.md 8(f)-3, the steps either side of the FV kernel: a periodic Cartesian grid of patches advanced by FVPatchGrid -- halo states
    taken from the neighbours inside the patch kernel (exa_fv_grid_step_device), CFL scan on the device, one host read per step -- next to the
    BARE patch kernel on the same array (no halo fill, fixed dt), for the reference's patch shape and for cfg 4's limiter patch."""
    shapes = {"ref-4x4": dict(dim=2, grid=(1024, 1024), P=4, n_aux=5, pde=exa.PDE_EULER_REF2D),
              "limiter-15^3": dict(dim=3, grid=(16, 16, 32), P=15, n_aux=0, pde=exa.PDE_EULER)}
    steps, warm = max(a.steps, 10), max(a.warmup, 3)
    out_shapes = {}
    for name, c in shapes.items():
        dim, grid, P, V = c["dim"], c["grid"], c["P"], 5 + c["n_aux"]
        fv = exa.FVPatchGrid(dim, grid, P, 1, 5, c["n_aux"], c["pde"], exa.FV_RUSANOV, device=local)
        g = torch.Generator(device="cuda")
        g.manual_seed(3)
        U = fv.U
        U.copy_(torch.rand(U.shape, generator=g, device="cuda", dtype=torch.float64))
        U[..., 0] += 1.0
        U[..., 1:4] *= 0.2
        U[..., 3 if dim == 2 else 4] += 3.0
        fv.invalidate()
        n_patches = int(torch.tensor(grid).prod())
        vols = n_patches * P ** dim
        S = P + 2
        b_alg = vols * 16 * V                                        # every state read once, written once (halo-less arrays)
        b_bare = vols * (8 * V * (S / P) ** dim + 8 * 5)             # the bare kernel on the reference layout: patch + halo read, n_real written
        dt = 0.2 * fv.h / dim / fv.max_eigenvalue()

        def timed(fn, n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n
        # the bare patch kernel: in place on the reference layout (array with halo), whatever the halo layers hold, fixed dt
        two = exa.FVPatchGrid(dim, grid, P, 1, 5, c["n_aux"], c["pde"], exa.FV_RUSANOV, device=local, fused=False)
        two.set_interior(fv.U)
        two.fill_halos()
        bare = lambda: two.kernel.time_step(two.Q, 0.05 * dt, two.h)
        timed(bare, warm)
        t_bare = timed(bare, steps)
        two.set_interior(fv.U)
        timed(lambda: two.step(0.05 * dt), 2)
        t_two = timed(lambda: two.step(0.05 * dt), max(3, steps // 3))
        del two
        torch.cuda.empty_cache()
        timed(lambda: fv.step(0.05 * dt), warm)
        t_step = timed(lambda: fv.step(0.05 * dt), steps)
        t_cfl = timed(lambda: (fv.invalidate(), fv.max_eigenvalue_device()), steps)

        def cfl_step():
            lam = fv.max_eigenvalue()                                            # left on the device by the previous step + the step's one host read
            fv.step(min(0.05 * dt, 0.2 * fv.h / dim / lam))
        timed(cfl_step, 2)
        t_full = timed(cfl_step, steps)
        out_shapes[name] = {"patches": n_patches, "patch_size": P, "variables": V, "bare_kernel_ms": 1e3 * t_bare, "grid_step_ms": 1e3 * t_step,
                            "cfl_scan_ms": 1e3 * t_cfl, "step_with_cfl_ms": 1e3 * t_full, "step_with_cfl_over_bare": t_full / t_bare,
                            "grid_step_over_bare": t_step / t_bare, "two_pass_torch_fill_ms": 1e3 * t_two,
                            "volume_updates_per_s": vols / t_full, "algorithmic_gbs": b_alg / t_step / 1e9,
                            "frac_of_hbm_peak": b_alg / t_step / 1e9 / HBM_PEAK_GBS, "bare_kernel_algorithmic_gbs": b_bare / t_bare / 1e9,
                            "finite": bool(torch.isfinite(fv.interior_device()).all().item())}
        tr, src = read_traffic({"ref-4x4": "traffic_fv_grid_4x4.json", "limiter-15^3": "traffic_fv_grid_15.json"}[name])
        out_shapes[name].update({"traffic": tr, "traffic_source": src, "algorithmic_bytes_per_launch": b_alg,
                                 "traffic_over_algorithmic": (tr / b_alg) if tr else None})
        del fv, U
        torch.cuda.empty_cache()
    r = out_shapes["ref-4x4"]
    return {"metric": "DoF-updates/sec, FV patch GRID step (halo states from the neighbours inside the patch kernel + device CFL scan), 1 MI355X",
            "value": r["volume_updates_per_s"] * 5, "unit": "DoF-updates/s", "n_gpus": 1, "steps": steps, "warmup": warm,
            "ms_per_step": r["step_with_cfl_ms"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "periodic grid of FV Rusanov patches advanced by FVPatchGrid (SURVEY.md 8(f)-3): 2^20 patches 4x4 (5+5 variables) "
                                   "and 8192 patches 15^3 (5 variables); per step: CFL scan (device) + one host read + the fused halo/update launch"},
            "finite": all(v["finite"] for v in out_shapes.values()),
            "roofline": {"kernel": "fv_rusanov_kernel<..., GRID> (4x4) / fv_rusanov_slab_kernel<..., GRID> (15^3)", "bound": "hbm",
                         "bound_basis": "algorithmic bytes of the halo-less arrays (every state read once, written once: 16 V B per volume)", "achieved": r["algorithmic_gbs"],
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": r["frac_of_hbm_peak"], "traffic": r["traffic"], "traffic_source": r["traffic_source"],
                         "traffic_15cubed": out_shapes["limiter-15^3"]["traffic"], "traffic_over_algorithmic_15cubed": out_shapes["limiter-15^3"]["traffic_over_algorithmic"]},
            "shapes": out_shapes}


def nccl_options():
    """RCCL's transport kernels run on the process group's own stream.  The round-3 review asked for a high-priority one; measured on the
    RCCL-to-self rehearsal (profiles/r04_bench_cfg2_self_exchange*.json) it is WORSE: the send / recv group then completes only when the
    persistent interior launch retires (span 146.9 ms, 0.17 ms exposed) where the normal-priority stream behind the solver's high-priority comm
    stream is done after 0.31 ms (0.0 exposed).  Default: normal; EXA_NCCL_STREAM_PRIORITY=high for a node where it measures otherwise."""
    import torch.distributed as dist
    return dist.ProcessGroupNCCL.Options(is_high_priority_stream=os.environ.get("EXA_NCCL_STREAM_PRIORITY", "normal") == "high")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg1", "cfg4", "fv-ref", "fv-grid", "cfg2_sympy", "cfg4_sympy"])
    ap.add_argument("--cells", type=int, default=128, help="cells per axis per GPU (default: BASELINE configs[2])")
    ap.add_argument("--order", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="cfg2 on one GPU: do not append the short cfg1 / cfg4 / fv-ref runs (`other_configs`) to the line")
    ap.add_argument("--no-reserve-trial", action="store_true", help="sharded runs: skip the warm-up trial of reserve_cus 0 / 8")
    ap.add_argument("--backend", default="nccl", help="rehearsal only: 'gloo' runs the multi-rank path with host-staged exchange")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--self-exchange", action="store_true",
                    help="rehearsal only (--gpus 1, cfg2 / cfg4): run the SHARDED step on one GPU, the periodic wrap going through RCCL send/recv to self")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if a.config not in ("cfg2", "cfg2_sympy", "cfg4", "cfg4_sympy"):
            sys.exit("bench.py: --config %s is a single-GPU configuration" % a.config)
        spawn_ranks(a.gpus, sys.argv[1:])                      # does not return
    if a.gpus != world:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE = %d" % (a.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    # stdout carries ONE line: the JSON record.  Libraries write there too (RCCL prints a version table on rank 0 when its
    # communicator comes up), so file descriptor 1 is pointed at stderr for the run and the record goes to the saved one.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from exahype_amd import solvers as exa

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback in the product path)")
    if a.share_gpu:
        local = 0
    elif local >= torch.cuda.device_count():
        sys.exit("bench.py: rank %d has no GPU (%d visible); --share-gpu is for rehearsals only" % (rank, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            # RCCL's transport kernels run on the process group's OWN stream (not on the solver's comm stream, which only orders them): its
            # priority is a measured choice, see nccl_options()
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), pg_options=nccl_options())
        else:
            dist.init_process_group(a.backend)
    elif a.self_exchange:
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        dist.init_process_group("nccl", rank=0, world_size=1, init_method="tcp://127.0.0.1:%d" % port,
                                device_id=torch.device("cuda", local), pg_options=nccl_options())

    if a.config == "cfg2":
        out = run_cfg2(a, torch, exa, world, rank, local)
        if world == 1 and not a.self_exchange and not a.no_other_configs:
            # the other rows of SURVEY.md 8(d), briefly, in the same (driver-run) invocation: value, ms per step and roofline of each;
            # metric / value / config of the line stay those of cfg 2
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            b = argparse.Namespace(**vars(a))
            b.no_cpu_baseline = True
            b.steps, b.warmup = 3, 1
            others = {}
            b2 = argparse.Namespace(**vars(b))
            b2.warmup = 2

            def sympy_id():
                return sympy_euler().register()                 # (JIT-compiled by hipcc on first use; __graft_entry__.build() prebuilds it in-tree)
            runs = (("cfg2_sympy", lambda: run_cfg2(b2, torch, exa, 1, 0, local, pde=sympy_id())), ("cfg1", lambda: run_cfg1(b, torch, exa, local)),
                    ("cfg4", lambda: run_cfg4(b, torch, exa, 1, 0, local)), ("cfg4_sympy", lambda: run_cfg4(b, torch, exa, 1, 0, local, pde=sympy_id())),
                    ("fv-ref", lambda: run_fv_ref(b, torch, exa, local)), ("fv-grid", lambda: run_fv_grid(b, torch, exa, local)))
            for name, fn in runs:
                try:
                    r = fn()
                    others[name] = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "finite", "roofline")}
                    others[name]["workload"] = r["config"]["workload"]
                    others[name]["term_set"] = r["config"].get("term_set")
                    if "shapes" in r:
                        others[name]["shapes"] = r["shapes"]
                    base = out if name == "cfg2_sympy" else others.get("cfg4") if name == "cfg4_sympy" else None
                    if base and "value" in base:
                        others[name]["vs_builtin_term_set"] = r["value"] / base["value"]
                except Exception as e:                                     # a failing side line must not take the headline with it -- it says so
                    others[name] = {"error": "%s: %s" % (type(e).__name__, e)}
                gc.collect()
                torch.cuda.empty_cache()
            out["other_configs"] = others
    elif a.config == "cfg2_sympy":
        out = run_cfg2(a, torch, exa, world, rank, local, pde=sympy_euler().register())
    elif a.config == "cfg4_sympy":
        out = run_cfg4(a, torch, exa, world, rank, local, pde=sympy_euler().register())
    elif a.config == "fv-grid":
        out = run_fv_grid(a, torch, exa, local)
    elif a.config == "cfg1":
        out = run_cfg1(a, torch, exa, local)
    elif a.config == "cfg4":
        out = run_cfg4(a, torch, exa, world, rank, local)
    else:
        out = run_fv_ref(a, torch, exa, local)
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or a.self_exchange:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
