"""Step time at N = 6: two kernels (stage A, stage B) against the one-kernel step -- development aid.  usage: quick_bench_step.py [cells] [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exahype_amd import solvers as exa
from tests.util import euler_dg_state
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
N = 6
for mode in (False, True, False, True):
    s = exa.AderDgSolver(3, N, (n, n, n), one_kernel_step=mode)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    u = torch.rand(s._u.shape, generator=g, device="cuda", dtype=torch.float64) * 0.1
    u[..., 0] += 1.0; u[..., 4] += 2.5
    s._u.copy_(u)
    dt = 1e-5
    for _ in range(2): s.step(dt)
    s.flush(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): s.step(dt)
    s.flush(); torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    print(f"{n}^3 one_kernel={mode}: {el*1e3:.2f} ms/step  finite={bool(torch.isfinite(s.u).all())}", flush=True)
    del s; torch.cuda.empty_cache()
