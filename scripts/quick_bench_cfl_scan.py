import sys, os, time
sys.path.insert(0, "/root/repo")
import torch
from exahype_amd import solvers as exa
N, nc = 6, 128
s = exa.AderDgSolver(3, N, (nc,) * 3)
g = torch.Generator(device='cuda'); g.manual_seed(4)
sh = s.u.shape[:-1]
rho = 1 + 0.2 * torch.rand(sh, generator=g, device='cuda', dtype=torch.float64)
s.u[..., 0] = rho
for a in range(3): s.u[..., 1 + a] = rho * (0.4 * torch.rand(sh, generator=g, device='cuda', dtype=torch.float64) - 0.2)
s.u[..., 4] = 2.6 + 0.5 * torch.rand(sh, generator=g, device='cuda', dtype=torch.float64)
lam = float(s.max_eigenvalue()[0])
dt = 0.3 * min(s.dx) / ((2 * N - 1) * 3 * lam)
def timed(fn, reps):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
s.predictor_volume(dt)
out = torch.zeros(1, dtype=torch.float64, device='cuda')
tb = timed(lambda: s.riemann_corrector(1e-9), 10)
tc = timed(lambda: s.riemann_corrector(1e-9, lam_out=out), 10)
ts = timed(lambda: s.max_eigenvalue(), 10)
print("stage B %.3f ms | stage B + scan in the launch %.3f ms | separate scan %.3f ms" % (tb * 1e3, tc * 1e3, ts * 1e3))
t_end = s.time + 6.5 * dt
t0 = time.perf_counter(); n = s.run(t_end, cfl=0.3); torch.cuda.synchronize(); el = time.perf_counter() - t0
print("run(): %d steps, %.1f ms per step (fused scan)" % (n, el / n * 1e3))
s._fused_backup = s.can_fuse_cfl_scan
s.can_fuse_cfl_scan = lambda: False
t_end = s.time + 6.5 * dt
t0 = time.perf_counter(); n = s.run(t_end, cfl=0.3); torch.cuda.synchronize(); el = time.perf_counter() - t0
print("run(): %d steps, %.1f ms per step (separate scan)" % (n, el / n * 1e3))
