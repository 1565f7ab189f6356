"""Grid step of 2^20 4x4 patches (5+5 variables) and 8192 15^3 patches: ms per launch (development aid).  usage: quick_bench_fv_grid.py [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exahype_amd import solvers as exa
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for name, dim, grid, P, n_aux, pde in (("4x4", 2, (1024, 1024), 4, 5, exa.PDE_EULER_REF2D), ("15^3", 3, (16, 16, 32), 15, 0, exa.PDE_EULER)):
    fv = exa.FVPatchGrid(dim, grid, P, 1, 5, n_aux, pde, exa.FV_RUSANOV)
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    U = fv.U
    U.copy_(torch.rand(U.shape, generator=g, device="cuda", dtype=torch.float64))
    U[..., 0] += 1.0; U[..., 1:4] *= 0.2; U[..., 3 if dim == 2 else 4] += 3.0
    fv.invalidate()
    dt = 0.01 * fv.h / dim / 3.0
    for _ in range(3): fv.step(dt)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fv.step(dt)
    e1.record(); torch.cuda.synchronize()
    print(f"grid step {name}: {e0.elapsed_time(e1)/reps:.4f} ms", flush=True)
