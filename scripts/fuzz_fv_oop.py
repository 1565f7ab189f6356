"""Random shapes through the out-of-place FV update against the in-place one (development aid): both modes, with / without auxiliary variables."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from exahype_amd import solvers as exa
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for it in range(60):
    dim = int(rng.integers(2, 4))
    P = int(rng.integers(2, 17 if dim == 2 else 9))
    H = int(rng.integers(1, 3))
    aux = int(rng.integers(0, 4))
    n = int(rng.integers(1, 40))
    mode = exa.FV_RUSANOV if rng.random() < 0.5 else exa.FV_FAITHFUL
    pde = exa.PDE_EULER
    S, V = P + 2 * H, 5 + aux
    k = exa.FVRusanovKernel(dim, P, H, 5, aux, n, pde, mode)
    Q = torch.rand((n,) + (S,) * dim + (V,), device="cuda", dtype=torch.float64)
    Q[..., 0] += 1.0; Q[..., 4] += 3.0
    out = k.time_step_oop(Q, 1e-4, 0.1)
    Q2 = Q.clone()
    k.time_step(Q2, 1e-4, 0.1)
    core = (slice(None),) + (slice(H, H + P),) * dim
    d = (out - Q2[core]).abs().max().item()
    ok = d < 1e-12 and bool(torch.isfinite(out).all())
    bad += 0 if ok else 1
    if not ok: print("MISMATCH", dim, P, H, aux, n, mode, d)
print("fuzz done, mismatches:", bad)
