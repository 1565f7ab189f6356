"""Stage A for term sets with position- / time-dependent terms or a non-conservative product (development aid): ms per launch on 3-D cells,
Euler with a source that depends on x and t (`xt`) and, optionally, the pressure gradient moved into a non-conservative product (`ncp`).
usage: quick_bench_plain.py N cells [xt|ncp] [build]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, sympy
from exahype_amd import solvers as exa
from exahype_amd.pde_codegen import SympyPDE

N, n = int(sys.argv[1]), int(sys.argv[2])
kind = sys.argv[3] if len(sys.argv) > 3 else "xt"
G4 = sympy.Float(0.4)


def pressure(q):
    return G4 * (q[4] - (q[1] ** 2 + q[2] ** 2 + q[3] ** 2) / (2 * q[0]))


def flux(q, x, t, d):
    p, un = pressure(q), q[d + 1] / q[0]
    f = [un * q[0], un * q[1], un * q[2], un * q[3], un * (q[4] + p)]
    if kind != "ncp":
        f[d + 1] += p
    return f


def eig(q, x, t, d):
    return sympy.Abs(q[d + 1] / q[0]) + sympy.sqrt(sympy.Float(1.4) * pressure(q) / q[0])


def source(q, x, t):
    g = [0.1 * sympy.sin(x[0] + t), 0.05 * x[1], -0.1 * sympy.cos(t)]
    return [0, q[0] * g[0], q[0] * g[1], q[0] * g[2], q[1] * g[0] + q[2] * g[1] + q[3] * g[2]]


def ncp(q, dq, d):
    # grad p = dp/dq . grad q  (the pressure term of the momentum flux as a non-conservative product)
    P = pressure(q)
    gp = sum(sympy.diff(P, q[v]) * dq[v] for v in range(5))
    out = [0, 0, 0, 0, 0]
    out[d + 1] = gp
    return out


p = SympyPDE(5, flux, eig, max_dim=3, name="euler_" + kind, source=source, ncp=ncp if kind == "ncp" else None)
if len(sys.argv) > 4 and sys.argv[4] == "build":                  # (no GPU needed: compile the term set's library in-tree so that it travels, print its path)
    print(p.build())
    sys.exit(0)
s = exa.AderDgSolver(3, N, (n, n, n), pde=p.register(), n_vars=5, origin=[0.1, 0.2, 0.3], time=0.5)
g = torch.Generator(device='cuda'); g.manual_seed(4)
sh = s.u.shape[:-1]
rho = 1 + 0.2 * torch.rand(sh, generator=g, device='cuda', dtype=torch.float64)
s.u[..., 0] = rho
for a in range(3): s.u[..., 1 + a] = rho * (0.4 * torch.rand(sh, generator=g, device='cuda', dtype=torch.float64) - 0.2)
s.u[..., 4] = 2.6 + 0.5 * torch.rand(sh, generator=g, device='cuda', dtype=torch.float64)
dt = 0.05 * s.dx[0] / (2 * N - 1) / 3 / 2.5
s.predictor_volume(dt); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): s.predictor_volume(dt)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 3 * 1e-3
w = s.work()
print(f"{kind}: {s.stage_a_kernel_name()} N={N} {n}^3 cells: {t*1e3:.3f} ms per launch = {w['flop_a']/t/1e12:.2f} TFLOP/s of the conservative scheme's flops ({w['flop_a']/t/78.6e12:.3f} of 78.6)  finite={bool(torch.isfinite(s.u).all())}")
