"""Achievable HBM bandwidth of plain torch streaming kernels on this box (calibration for the memory-bound conv layers)."""
import torch, time
dev = "cuda:0"
def bench(fn, nbytes, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    return us, nbytes / us / 1e6
for mb in (54, 215, 430, 860):
    n = mb * 1000 * 1000 // 2
    x = torch.randn(n, device=dev, dtype=torch.float16); y = torch.empty_like(x); z = torch.empty_like(x)
    print("size %4d MB: fill %6.1f us %5.2f TB/s | copy %6.1f us %5.2f TB/s (r+w) | sum %6.1f us %5.2f TB/s | add3 %6.1f us %5.2f TB/s (2r+w) | relu_ %6.1f us %5.2f TB/s (r+w in place)" % (
        (mb,) + bench(lambda: y.fill_(1.0), n * 2) + bench(lambda: y.copy_(x), n * 4) + bench(lambda: x.sum(), n * 2)
        + bench(lambda: torch.add(x, y, out=z), n * 6) + bench(lambda: x.relu_(), n * 4)))
