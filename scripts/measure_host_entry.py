"""Rate of the host-buffer entry points (the reference's own boundary: `time_step(double* Q, double dt)` on host memory): staging through HBM
over PCIe included.  usage: measure_host_entry.py [log2 patches = 18]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from exahype_amd import solvers as exa

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 18
n = 1 << lg
k = exa.FVRusanovKernel(2, 4, 1, 5, 5, n, exa.PDE_EULER_REF2D, exa.FV_FAITHFUL)
Q = np.empty((n, 6, 6, 10))
Q[:] = 1.0 + 0.1 * np.sin(np.arange(360).reshape(6, 6, 10))
Q[..., 4] = 3.0
k.time_step(Q, 1e-3)                       # numpy array: exa_fv_time_step_host (copy in, kernel, copy out)
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    k.time_step(Q, 1e-3)
t = (time.perf_counter() - t0) / reps
print(f"FV host entry, pageable numpy, {n} patches ({Q.nbytes / 1e6:.0f} MB each way): {t * 1e3:.1f} ms per call = {n * 16 * 5 / t:.3e} DoF-updates/s, "
      f"{2 * Q.nbytes / t / 1e9:.1f} GB/s over the link", flush=True)
# pinned host memory handed over as a torch tensor -> .numpy() view
Qp = torch.from_numpy(Q).pin_memory()
Qv = Qp.numpy()
k.time_step(Qv, 1e-3)
t0 = time.perf_counter()
for _ in range(reps):
    k.time_step(Qv, 1e-3)
t = (time.perf_counter() - t0) / reps
print(f"FV host entry, pinned buffer: {t * 1e3:.1f} ms per call = {n * 16 * 5 / t:.3e} DoF-updates/s, {2 * Q.nbytes / t / 1e9:.1f} GB/s over the link", flush=True)
# device-resident for comparison
Qd = torch.from_numpy(Q).cuda()
k.time_step(Qd, 1e-3); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    k.time_step(Qd, 1e-3)
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / 10
print(f"FV device entry: {t * 1e3:.3f} ms per call = {n * 16 * 5 / t:.3e} DoF-updates/s", flush=True)
