"""Plain stage A (term sets with position- / time-dependent terms or an ncp) -- development aid: ms per launch on 3-D N = 6."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from exahype_amd import solvers as exa
from tests.test_user_pde import coupled_xt_ncp_system
p = coupled_xt_ncp_system(max_dim=3)
N, n = 6, 16
s = exa.AderDgSolver(3, N, (n, n, n), pde=p.register(), n_vars=3)
s.upload(1.0 + 0.1 * np.random.default_rng(0).random((n, n, n, N, N, N, 3)))
s.predictor_volume(1e-5); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): s.predictor_volume(1e-5)
e1.record(); torch.cuda.synchronize()
print(f"{s.stage_a_kernel_name()}: {e0.elapsed_time(e1)/5:.3f} ms per {n}^3 launch  finite={bool(torch.isfinite(s.u).all())}")
