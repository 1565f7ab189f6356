"""Stage A of the built-in Euler term set against the SAME system given as SymPy expressions (pde_codegen.SympyPDE), same cells,
same state (development aid).  usage: quick_bench_sympy.py N cells [reps]   (EXA_SYMPY_ONLY=1: only the generated term set)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exahype_amd import solvers as exa
from bench import sympy_euler as euler_sympy

N, nc = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
spde = euler_sympy()
if os.environ.get("EXA_USER_DG_FLAGS") is not None:              # e.g. "" (default scheduling) or "-mllvm -amdgpu-sched-strategy=max-memory-clause"
    spde.dg_flags = os.environ["EXA_USER_DG_FLAGS"].split()
pid = spde.register()
res = {}
order = (("built-in", exa.PDE_EULER), ("sympy", pid))
if os.environ.get("EXA_SYMPY_FIRST"): order = order[::-1]
for name, pde in order:
    if name == "built-in" and os.environ.get("EXA_SYMPY_ONLY"):
        continue
    s = exa.AderDgSolver(3, N, (nc,) * 3, pde=pde, n_vars=5)
    g = torch.Generator(device='cuda'); g.manual_seed(4)
    sh = s.u.shape[:-1]
    rho = 1 + 0.2 * torch.rand(sh, generator=g, device='cuda', dtype=torch.float64)
    s.u[..., 0] = rho
    for a in range(3): s.u[..., 1 + a] = rho * (0.4 * torch.rand(sh, generator=g, device='cuda', dtype=torch.float64) - 0.2)
    s.u[..., 4] = 2.6 + 0.5 * torch.rand(sh, generator=g, device='cuda', dtype=torch.float64)
    dt = 0.05 * s.dx[0] / (2 * N - 1) / 3 / 2.5
    u0 = s.u.clone()
    s.predictor_volume(dt); torch.cuda.synchronize()
    res[name] = (s.u.clone(), s.trace.clone() if hasattr(s, "trace") else None)
    s.u.copy_(u0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): s.predictor_volume(dt)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    w = s.work()
    print(f"{name:9s} {s.stage_a_kernel_name()} N={N} {nc}^3 cells: {t*1e3:.2f} ms/launch  {w['flop_a']/t/1e12:.2f} TFLOP/s  frac {w['flop_a']/t/78.6e12:.3f}", flush=True)
    del s
    torch.cuda.empty_cache()
if len(res) == 2:
    a, b = res["built-in"][0], res["sympy"][0]
    print("u* sympy vs built-in: max rel diff %.3e" % float((a - b).abs().max() / a.abs().max()))
