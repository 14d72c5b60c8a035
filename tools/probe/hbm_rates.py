"""Achievable HBM rates on this box for pure write / read / copy streams (torch kernels), for the roofline context."""
import torch, time
n = 1 << 27   # 1 GiB of doubles
a = torch.empty(n, dtype=torch.float64, device="cuda")
b = torch.empty(n, dtype=torch.float64, device="cuda")
def t(f, reps=20):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
gb = n * 8 / 1e9
print("write  %.0f GB/s" % (gb / t(lambda: a.fill_(1.0))))
print("read   %.0f GB/s" % (gb / t(lambda: a.sum())))
print("copy   %.0f GB/s (read+write bytes)" % (2 * gb / t(lambda: b.copy_(a))))
