"""BASELINE configs[4] on one GPU (development aid): 3-D Euler p=7 + FV subcell limiter, Bernoulli(0.05) troubled mask."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from exahype_amd import solvers as exa

def run(nc=32, N=8, frac=0.05, steps=3):
    s = exa.AderDgSolver(3, N, (nc,) * 3)
    g = torch.Generator(device='cuda'); g.manual_seed(4)
    sh = s.u.shape[:-1]
    rho = 1 + 0.2 * torch.rand(sh, generator=g, device='cuda', dtype=torch.float64)
    s.u[..., 0] = rho
    for a in range(3): s.u[..., 1 + a] = rho * (0.4 * torch.rand(sh, generator=g, device='cuda', dtype=torch.float64) - 0.2)
    s.u[..., 4] = 2.6 + 0.5 * torch.rand(sh, generator=g, device='cuda', dtype=torch.float64)
    mask = torch.rand((nc,) * 3, generator=g, device='cuda') < frac
    lim = exa.SubcellLimiter(s)
    dt = 0.05 * s.dx[0] / (2 * N - 1) / 3 / 2.5
    lim.step(dt, mask); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): n = lim.step(dt, mask)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / steps
    dof = nc ** 3 * N ** 3 * 5
    print(f"cfg4 shape: {nc}^3 cells p={N-1}, {n} troubled ({100*n/nc**3:.1f}%): {t*1e3:.1f} ms/step -> {dof/t/1e9:.3f} GDoF-updates/s  finite={bool(torch.isfinite(s.u).all())}")

if __name__ == "__main__":
    run()
