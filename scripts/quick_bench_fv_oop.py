"""Timing of the reference-configuration FV update: in place vs out of place (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exahype_amd import solvers as exa

P, H, m, aux, n = 4, 1, 5, 5, 1 << 20
S, V = P + 2 * H, m + aux
for mode in (exa.FV_FAITHFUL, exa.FV_RUSANOV):
    k = exa.FVRusanovKernel(2, P, H, m, aux, n, exa.PDE_EULER_REF2D, mode)
    g = torch.Generator(device='cuda'); g.manual_seed(0)
    Q = torch.rand((n, S, S, V), generator=g, device='cuda', dtype=torch.float64)
    Q[..., 0] += 1.0; Q[..., 3] += 3.0
    out = torch.empty((n, P, P, V), device='cuda', dtype=torch.float64)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, fn in (("oop", lambda: k.time_step_oop(Q, 1e-4, 0.1, out=out)), ("in place", lambda: k.time_step(Q, 1e-4, 0.1))):
        fn(); torch.cuda.synchronize()
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"mode {mode} {name}: {e0.elapsed_time(e1)/10:.3f} ms", flush=True)
