"""Which part of the generated term set costs the N = 6 stage A its few per cent against exa::Euler: the generated header with single members
replaced by the hand-written bodies (development aid).  usage: quick_bench_sympy_variants.py cells reps variant..."""
import sys, os, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exahype_amd import solvers as exa
import bench

HAND = {
    "dir": ("flux_scaled_dir(const double* q, const double* a, const Dir& c, double* F) {", """
        const double coeff = a[0] * fma(q[3], c.n[2], fma(q[2], c.n[1], q[1] * c.n[0]));
        F[0] = coeff * q[0];
        F[1] = fma(a[1], c.n[0], coeff * q[1]);
        F[2] = fma(a[1], c.n[1], coeff * q[2]);
        F[3] = fma(a[1], c.n[2], coeff * q[3]);
        F[4] = coeff * (q[4] + a[1]);
    }"""),
    "aux": ("aux_fast(const double* q, double* a) {", """
        const double irho = exa::fast_rcp(q[0]);
        a[0] = irho;
        a[1] = (1.4 - 1) * (q[4] - 0.5 * irho * (q[1] * q[1] + q[2] * q[2] + q[3] * q[3]));
    }"""),
}


def variant(names):
    p = bench.sympy_euler()
    src = p.source()
    for n in names:
        if n == "none":
            continue
        head, body = HAND[n]
        i = src.index(head) + len(head)
        j = src.index("\n    }", i) + len("\n    }")
        src = src[:i] + body + src[j:]
    p.source = lambda: src
    return p


cells, reps = int(sys.argv[1]), int(sys.argv[2])
for v in sys.argv[3:]:
    pid = exa.PDE_EULER if v == "builtin" else variant(v.split("+")).register()
    s = exa.AderDgSolver(3, 6, (cells,) * 3, pde=pid, n_vars=5)
    g = torch.Generator(device='cuda'); g.manual_seed(4)
    sh = s.u.shape[:-1]
    rho = 1 + 0.2 * torch.rand(sh, generator=g, device='cuda', dtype=torch.float64)
    s.u[..., 0] = rho
    for a in range(3): s.u[..., 1 + a] = rho * (0.4 * torch.rand(sh, generator=g, device='cuda', dtype=torch.float64) - 0.2)
    s.u[..., 4] = 2.6 + 0.5 * torch.rand(sh, generator=g, device='cuda', dtype=torch.float64)
    dt = 0.05 * s.dx[0] / 11 / 3 / 2.5
    s.predictor_volume(dt); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): s.predictor_volume(dt)
    e1.record(); torch.cuda.synchronize()
    print("%-12s %.2f ms" % (v, e0.elapsed_time(e1) / reps), flush=True)
    del s
