"""HBM rates of this chip for plain streams (torch kernels): read-only (sum), write-only (fill), copy -- context for the HBM-bound kernels' rooflines."""
import torch, time
n = 1 << 29                                        # 4 GiB of float64
x = torch.rand(n, device="cuda", dtype=torch.float64)
y = torch.empty_like(x)
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
b = n * 8
t = timed(lambda: x.sum());            print(f"read  (sum of 4 GiB)        : {b / t / 1e12:.2f} TB/s")
t = timed(lambda: y.fill_(1.0));       print(f"write (fill of 4 GiB)       : {b / t / 1e12:.2f} TB/s")
t = timed(lambda: y.copy_(x));         print(f"copy  (4 GiB -> 4 GiB)      : {2 * b / t / 1e12:.2f} TB/s (read + write)")
t = timed(lambda: torch.add(x, 1.0, out=y)); print(f"add   (read 4 GiB, write 4) : {2 * b / t / 1e12:.2f} TB/s (read + write)")
