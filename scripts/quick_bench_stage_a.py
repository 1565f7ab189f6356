"""Stage-A launch time and TFLOP/s for one (N, cells) (development aid).  usage: quick_bench_stage_a.py N cells [reps] [n_picard]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exahype_amd import solvers as exa

N, nc = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
npic = int(sys.argv[4]) if len(sys.argv) > 4 else -1
s = exa.AderDgSolver(3, N, (nc,) * 3, n_picard=npic)
g = torch.Generator(device='cuda'); g.manual_seed(4)
sh = s.u.shape[:-1]
rho = 1 + 0.2 * torch.rand(sh, generator=g, device='cuda', dtype=torch.float64)
s.u[..., 0] = rho
for a in range(3): s.u[..., 1 + a] = rho * (0.4 * torch.rand(sh, generator=g, device='cuda', dtype=torch.float64) - 0.2)
s.u[..., 4] = 2.6 + 0.5 * torch.rand(sh, generator=g, device='cuda', dtype=torch.float64)
dt = 0.05 * s.dx[0] / (2 * N - 1) / 3 / 2.5
s.predictor_volume(dt); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): s.predictor_volume(dt)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / reps * 1e-3
w = s.work()
print(f"stage A N={N} n_picard={npic} {nc}^3 cells: {t*1e3:.2f} ms/launch  {w['flop_a']/t/1e12:.2f} TFLOP/s  finite={bool(torch.isfinite(s.u).all())}")
