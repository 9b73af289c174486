"""What a plain streaming read / copy reaches on this GPU (torch kernels), to put the narrow SpMM's rate next to it."""
import time, torch
x = torch.empty(600_000_000 // 4, dtype=torch.float32, device="cuda").normal_()
y = torch.empty_like(x)
def t(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ms = t(lambda: x.sum())
print("sum   600 MB: %.4f ms  %.2f TB/s read" % (ms, 0.6 / ms))
ms = t(lambda: y.copy_(x))
print("copy  600 MB: %.4f ms  %.2f TB/s read+write" % (ms, 1.2 / ms))
ms = t(lambda: torch.max(x))
print("max   600 MB: %.4f ms  %.2f TB/s read" % (ms, 0.6 / ms))
z = torch.empty(2_400_000_000 // 4, dtype=torch.float32, device="cuda").normal_()
ms = t(lambda: z.sum())
print("sum  2400 MB: %.4f ms  %.2f TB/s read" % (ms, 2.4 / ms))
