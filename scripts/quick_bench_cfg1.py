"""BASELINE configs[1] on one GPU (development aid): 2-D Euler p=3, 512x512 cells, single-stage step (volume + Riemann + corrector)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exahype_amd import solvers as exa

def run(N=4, nc=(512, 512), steps=20):
    s = exa.AderDgSolver(2, N, nc, n_picard=0, fused_single_stage=True)
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    sh = s.u.shape[:-1]
    rho = 1 + 0.2 * torch.rand(sh, generator=g, device='cuda', dtype=torch.float64)
    s.u[..., 0] = rho
    for a in range(3): s.u[..., 1 + a] = rho * (0.4 * torch.rand(sh, generator=g, device='cuda', dtype=torch.float64) - 0.2)
    s.u[..., 4] = 2.6 + 0.5 * torch.rand(sh, generator=g, device='cuda', dtype=torch.float64)
    dt = 0.05 * min(s.dx) / (2 * N - 1) / 2.5
    dof = nc[0] * nc[1] * N * N * 5
    def timed(fn):
        fn(); fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps
    t_fused = timed(lambda: s.step(dt))
    def two_kernel():
        s.predictor_volume(dt); s.riemann_corrector(dt)
    t_two = timed(two_kernel)
    b_io = 2 * dof * 8
    print(f"cfg1 {nc[0]}x{nc[1]} cells p={N-1}: fused step {t_fused:.3f} ms = {dof/t_fused/1e6:.1f} GDoF-updates/s "
          f"({b_io/t_fused/1e6:.0f} GB/s of u in+out) | two kernels {t_two:.3f} ms = {dof/t_two/1e6:.1f} GDoF-updates/s  finite={bool(torch.isfinite(s.u).all())}")

if __name__ == "__main__":
    run()
    run(N=4, nc=(2048, 2048), steps=5)
