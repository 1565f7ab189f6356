"""Quick timing of the fused FV Rusanov patch kernel (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from exahype_amd import solvers as exa

def run(dim, P, H, n_real, n_aux, n_patches, pde, mode, steps=10):
    S, V = P + 2 * H, n_real + n_aux
    k = exa.FVRusanovKernel(dim, P, H, n_real, n_aux, n_patches, pde, mode)
    g = torch.Generator(device='cuda'); g.manual_seed(0)
    Q = torch.rand((n_patches,) + (S,) * dim + (V,), generator=g, device='cuda', dtype=torch.float64)
    Q[..., 0] += 1.0
    Q[..., (3 if pde == exa.PDE_EULER_REF2D else 4)] += 3.0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    k.time_step(Q, 1e-4, 0.1); torch.cuda.synchronize()
    e0.record()
    for _ in range(steps): k.time_step(Q, 1e-4, 0.1)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / steps * 1e-3
    vols = n_patches * P ** dim
    alg = vols * (8 * V * (S / P) ** dim + 8 * n_real)        # read patch+halo once, write n_real once
    print(f"FV dim {dim} P {P} H {H} vars {n_real}+{n_aux} patches {n_patches} mode {mode}: {t*1e3:.3f} ms  "
          f"{vols*n_real/t/1e9:.2f} GDoF-upd/s  {alg/t/1e9:.0f} GB/s algorithmic ({alg/t/8e12*100:.1f}% of 8 TB/s)  finite={bool(torch.isfinite(Q).all())}", flush=True)

if __name__ == "__main__":
    run(2, 4, 1, 5, 5, 1 << 20, exa.PDE_EULER_REF2D, exa.FV_FAITHFUL)
    if len(sys.argv) > 1 and sys.argv[1] == "ref":
        sys.exit(0)
    run(2, 4, 1, 5, 5, 1 << 20, exa.PDE_EULER_REF2D, exa.FV_RUSANOV)
    run(3, 15, 1, 5, 0, 8192, exa.PDE_EULER, exa.FV_RUSANOV)
    run(3, 15, 1, 5, 0, 8192, exa.PDE_EULER, exa.FV_FAITHFUL)
    run(2, 16, 1, 5, 0, 1 << 16, exa.PDE_EULER, exa.FV_RUSANOV)
