import sys, os
sys.path.insert(0, '/root/repo')
import torch
from exahype_amd import solvers as exa
N, nc = 6, 96
s = exa.AderDgSolver(3, N, (nc,) * 3, pde=exa.PDE_EULER, n_vars=5)
g = torch.Generator(device='cuda'); g.manual_seed(4)
sh = s.u.shape[:-1]
rho = 1 + 0.2 * torch.rand(sh, generator=g, device='cuda', dtype=torch.float64)
s.u[..., 0] = rho
for a in range(3): s.u[..., 1 + a] = rho * 0.1
s.u[..., 4] = 2.6
dt = 1e-5
s.step(dt); torch.cuda.synchronize()
def ev(): return torch.cuda.Event(enable_timing=True)
for prio in (0, -1):
    side = torch.cuda.Stream(priority=prio)
    for R in (0, 8, 16, 32):
        s.set_reserve_cus(R)
        cur = torch.cuda.current_stream()
        a0, a1, b0, b1 = ev(), ev(), ev(), ev()
        torch.cuda.synchronize()
        a0.record(cur)
        s.predictor_volume(dt)
        a1.record(cur)
        with torch.cuda.stream(side):
            b0.record(side)
            # stage B on half of the block, on the side stream, no dependency on A (timing only)
            s.riemann_corrector(dt, [0, 0, 0], [nc // 2, nc, nc])
            b1.record(side)
        torch.cuda.synchronize()
        print(f"prio {prio} reserve {R:2d}: A {a0.elapsed_time(a1):7.2f} ms | B starts {a0.elapsed_time(b0):7.2f} ends {a0.elapsed_time(b1):7.2f} (alone: see R=0 row of B after A)", flush=True)
# B alone
b0, b1 = ev(), ev()
b0.record(); s.riemann_corrector(dt, [0, 0, 0], [nc // 2, nc, nc]); b1.record(); torch.cuda.synchronize()
print("B (half block) alone:", b0.elapsed_time(b1), "ms")
