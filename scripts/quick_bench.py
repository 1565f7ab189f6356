"""Quick per-kernel timing of the ADER-DG stages (development aid, not the bench contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from exahype_amd import solvers as exa
from tests.util import euler_dg_state

def run(dim, N, nc, n_it=-1, steps=5):
    s = exa.AderDgSolver(dim, N, nc, n_picard=n_it)
    # smooth-ish admissible state, generated on device to save time
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    sh = s.u.shape[:-1]
    rho = 1 + 0.2 * torch.rand(sh, generator=g, device='cuda', dtype=torch.float64)
    s.u[..., 0] = rho
    for a in range(3): s.u[..., 1 + a] = rho * (0.4 * torch.rand(sh, generator=g, device='cuda', dtype=torch.float64) - 0.2)
    s.u[..., 4] = 2.5 + 0.5 * torch.rand(sh, generator=g, device='cuda', dtype=torch.float64) + 0.1
    dt = 0.05 * min(s.dx) / (2 * N - 1) / 2.5
    w = s.work()
    ea = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    s.step(dt); torch.cuda.synchronize()
    ta = tb = 0.0
    for _ in range(steps):
        ea[0].record(); s.predictor_volume(dt); ea[1].record(); s.riemann_corrector(dt); ea[2].record()
        torch.cuda.synchronize()
        ta += ea[0].elapsed_time(ea[1]); tb += ea[1].elapsed_time(ea[2])
    ta /= steps; tb /= steps
    ncell = int(np.prod(nc)); dof = ncell * N ** dim * 5
    print(f"dim {dim} N {N} cells {nc} n_it {n_it}: A {ta:.3f} ms ({w['flop_a']/ta/1e9:.2f} TFLOP/s, {w['bytes_a']/ta/1e6:.1f} GB/s)  "
          f"B {tb:.3f} ms ({w['bytes_b']/tb/1e6:.1f} GB/s)  -> {dof/(ta+tb)/1e-3/1e9:.3f} GDoF/s  finite={bool(torch.isfinite(s.u).all())}", flush=True)

if __name__ == "__main__":
    run(3, 6, (32, 32, 32))
    run(3, 6, (64, 64, 64))
    run(3, 4, (64, 64, 64))
    run(2, 4, (512, 512), n_it=0)
    run(2, 4, (512, 512))
