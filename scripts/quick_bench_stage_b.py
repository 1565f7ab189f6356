"""Stage B (Riemann + corrector) launch time -- development aid.  usage: quick_bench_stage_b.py N cells [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exahype_amd import solvers as exa
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
s = exa.AderDgSolver(3, N, (n, n, n))
g = torch.Generator(device="cuda"); g.manual_seed(1)
u = torch.rand(s._u.shape, generator=g, device="cuda", dtype=torch.float64) * 0.1
u[..., 0] += 1.0; u[..., 4] += 2.5
s._u.copy_(u)
dt = 1e-6
s.predictor_volume(dt)
s.riemann_corrector(dt); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): s.riemann_corrector(dt)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / reps
w = s.work()
print(f"stage B N={N} {n}^3: {t:.3f} ms/launch  {w['bytes_b']/t/1e6:.0f} GB/s algorithmic  finite={bool(torch.isfinite(s._u).all())}", flush=True)
