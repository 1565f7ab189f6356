"""Stage B (Riemann solve + corrector) alone on a 3-D block: ms per launch and the algorithmic HBM rate (development aid).
usage: quick_bench_stage_b.py N cells [reps]      (EXA_SB_SYMPY=1: the same system as SymPy expressions, bench.sympy_euler())"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exahype_amd import solvers as exa

N, nc = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
pde = exa.PDE_EULER
if os.environ.get("EXA_SB_SYMPY"):
    from bench import sympy_euler
    pde = sympy_euler().register()
s = exa.AderDgSolver(3, N, (nc,) * 3, pde=pde, n_vars=5)
g = torch.Generator(device='cuda'); g.manual_seed(4)
sh = s.u.shape[:-1]
rho = 1 + 0.2 * torch.rand(sh, generator=g, device='cuda', dtype=torch.float64)
s.u[..., 0] = rho
for a in range(3): s.u[..., 1 + a] = rho * (0.4 * torch.rand(sh, generator=g, device='cuda', dtype=torch.float64) - 0.2)
s.u[..., 4] = 2.6 + 0.5 * torch.rand(sh, generator=g, device='cuda', dtype=torch.float64)
dt = 0.05 * s.dx[0] / (2 * N - 1) / 3 / 2.5
s.predictor_volume(dt)
u0 = s.u.clone()
s.riemann_corrector(dt); torch.cuda.synchronize()
chk = float(s.u.double().sum())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): s.riemann_corrector(dt)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / reps * 1e-3
w = s.work()
print(f"stage B N={N} {nc}^3: {t*1e3:.3f} ms per launch, {w['bytes_b']/t/1e12:.2f} TB/s algorithmic (B_B); checksum after one launch {chk:.12e}", flush=True)
