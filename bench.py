#!/usr/bin/env python3
"""Headline benchmark: DoF-updates/s of the 3-D Euler p=5 ADER-DG step
(space-time predictor Picard loop + volume integral + face Riemann solve +
corrector) on 1/2/4/8 MI355X.

    python bench.py --gpus N --steps K --warmup W

N = 1 runs BASELINE.json configs[2] (128^3 cells, one GPU).  N > 1 (launched by
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) keeps
128^3 cells per GPU (weak scaling; N = 8 is configs[3], 256^3 cells) on a
Cartesian process grid with the RCCL face-trace halo exchange overlapped with
the interior predictor work.  Synthetic data: smooth Euler
density wave + seeded 1e-3 noise, fixed dt.  One "step" = one full time step of
every cell.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6     # MI355X datasheet: fp64 vector == fp64 matrix (MFMA) peak; DESIGN.md "Roofs"
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_MEASURED_GBS = 6290.0   # ... and its measured float4-copy rate


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


def cpu_baseline(N, n_it, seconds=12.0):
    """The oracle (oracle/exa_oracle.c, OpenMP over cells) on the host cores, same kernel stack, bounded sample."""
    import numpy as np
    import oracle
    from oracle.dg_operators import operators
    from tests.util import euler_dg_state
    oracle.lib()
    nc = (12, 12, 12)
    ops = operators(N)
    u = euler_dg_state(nc + (N, N, N), seed=2, amp=0.1).reshape(-1)
    dx = [1.0 / c for c in nc]
    dt = 1e-4
    u = oracle.aderdg_step(u, dt, dx, ops, 3, N, 5, oracle.PDE_EULER, n_it, nc)     # warm-up
    t0, steps = time.perf_counter(), 0
    while True:
        u = oracle.aderdg_step(u, dt, dx, ops, 3, N, 5, oracle.PDE_EULER, n_it, nc)
        steps += 1
        el = time.perf_counter() - t0
        if el >= seconds or steps >= 400:
            break
    dof = int(np.prod(nc)) * N ** 3 * 5
    threads = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    return {"value": dof * steps / el, "unit": "DoF-updates/s", "cores": threads, "kind": "port",
            "sample": "%d steps of a %dx%dx%d-cell block, same p=%d Euler ADER-DG step, oracle/exa_oracle.c "
                      "(gcc -O3 -fopenmp), %.1f s" % (steps, nc[0], nc[1], nc[2], N - 1, el)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cells", type=int, default=128, help="cells per axis per GPU (default: BASELINE configs[2])")
    ap.add_argument("--order", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="rehearsal only: 'gloo' runs the multi-rank path with host-staged exchange")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from exahype_amd import solvers as exa

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (a.gpus, a.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback in the product path)")
    if a.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    part = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(a.backend)
        part = exa.CartesianPartition(world, rank, 3)
    pdims = part.pdims if part else [1, 1, 1]
    coords = part.coords if part else [0, 0, 0]

    N = a.order + 1
    nc = [a.cells] * 3
    dx = [1.0 / (nc[d] * pdims[d]) for d in range(3)]
    s = exa.AderDgSolver(3, N, nc, pde=exa.PDE_EULER, n_vars=5, n_picard=-1, dx=dx, device=local, part=part,
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
    # stage-A launch durations, measured live with events on the stream the kernels are launched on
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    t0 = time.perf_counter()
    if world == 1:
        for k in range(a.steps):
            ev[k][0].record()
            s.predictor_volume(dt)
            ev[k][1].record()
            s.riemann_corrector(dt)
    else:
        for k in range(a.steps):
            s.step(dt)
    sync()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device="cuda" if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    finite = bool(torch.isfinite(s.u).all().item())

    dof_per_gpu = nc[0] * nc[1] * nc[2] * N ** 3 * 5
    value = dof_per_gpu * world * a.steps / el
    out = {
        "metric": "DoF-updates/sec, 3D Euler p=%d fused STP+volume+Riemann, 1/2/4/8 MI355X" % a.order,
        "value": value, "unit": "DoF-updates/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * el / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "3D compressible Euler, ADER-DG p=%d, %d^3 cells per GPU (%dx%dx%d process grid), "
                               "%d Picard iterations + volume + Riemann + corrector" % (a.order, a.cells, pdims[0], pdims[1], pdims[2], N),
                   "cells_per_gpu": a.cells ** 3, "order": a.order, "n_vars": 5, "dt": dt,
                   "parallelism": "cartesian-%dx%dx%d" % tuple(pdims)},
        "finite": finite,
    }
    if world == 1:
        ta = sum(e0.elapsed_time(e1) for e0, e1 in ev) / a.steps * 1e-3       # s per stage-A launch
        ach = work["flop_a"] / ta / 1e12
        traffic = None
        tf = os.path.join(ROOT, "profiles", "stage_a_traffic.json")
        if os.path.exists(tf):
            rec = json.load(open(tf))
            if rec.get("cells") == a.cells and rec.get("order") == a.order:
                traffic = rec.get("hbm_bytes_per_launch")
        out["roofline"] = {"kernel": "dg_stage_a_kernel<3,%d,Euler>" % N, "bound": "mfma", "achieved": ach,
                           "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS, "traffic": traffic,
                           "launch_ms": ta * 1e3, "flop_per_launch": work["flop_a"],
                           "hbm_achieved_gbs": work["bytes_a"] / ta / 1e9, "hbm_frac": work["bytes_a"] / ta / 1e9 / HBM_PEAK_GBS,
                           "hbm_frac_of_measured_copy": work["bytes_a"] / ta / 1e9 / HBM_MEASURED_GBS,
                           "note": "fp64: MFMA peak == vector peak = 78.6 TFLOP/s on MI355X; the kernel is fp64-compute-bound (48 FLOP/B)"}
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, N)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
