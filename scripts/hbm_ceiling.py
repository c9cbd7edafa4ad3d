"""Practical HBM ceilings of the box ("use the measured figure alongside the spec figure"):
pure-write (fill), copy (read+write) and pure-read (sum) rates with plain torch kernels on 8 GiB buffers."""
import time, torch
n = 1 << 30   # doubles -> 8 GiB
a = torch.empty(n, dtype=torch.float64, device="cuda"); b = torch.empty_like(a)
def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps
tw = timed(lambda: a.fill_(1.5)); tc = timed(lambda: b.copy_(a)); tr = timed(lambda: a.sum())
gb = n * 8 / 1e9
print("fill  (write only) %.0f GB/s" % (gb / tw))
print("copy  (read+write) %.0f GB/s total, %.0f GB/s written" % (2 * gb / tc, gb / tc))
print("sum   (read only)  %.0f GB/s" % (gb / tr))
