"""Random blocks and boxes through the two stage-A kernels of N = 6 (register-resident / LDS-resident) and the one- / two-kernel step (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from exahype_amd import solvers as exa
from tests.util import euler_dg_state, rel_err
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N, bad = 6, 0
for it in range(25):
    nc = tuple(int(x) for x in rng.integers(1, 12, 3))
    u = euler_dg_state(nc + (N,) * 3, seed=int(rng.integers(1 << 30)))
    n_it = int(rng.integers(1, 7))
    dt = 1e-3 * float(rng.random() + 0.5)
    lo = [int(rng.integers(0, c)) for c in nc]
    hi = [int(rng.integers(l + 1, c + 1)) for l, c in zip(lo, nc)]
    outs = []
    for variant in ("reg", "lds"):
        s = exa.AderDgSolver(3, N, nc, n_picard=n_it, stage_a=variant)
        s.upload(u)
        s.predictor_volume(dt, lo, hi)
        outs.append((s.download().copy(), s.trace.cpu().numpy().copy()))
    e1, e2 = rel_err(outs[0][0], outs[1][0]), rel_err(outs[0][1], outs[1][1])
    one = exa.AderDgSolver(3, N, nc, n_picard=n_it, one_kernel_step=True)
    two = exa.AderDgSolver(3, N, nc, n_picard=n_it)
    one.upload(u); two.upload(u)
    for k in range(3):
        one.step(dt * (1 + 0.2 * k)); two.step(dt * (1 + 0.2 * k))
    e3 = rel_err(one.download(), two.download())
    ok = e1 < 1e-12 and e2 < 1e-12 and e3 < 1e-12
    bad += 0 if ok else 1
    if not ok: print("MISMATCH", nc, lo, hi, n_it, e1, e2, e3)
print("fuzz done, mismatches:", bad)
