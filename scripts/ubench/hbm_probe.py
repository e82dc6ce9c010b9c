#!/usr/bin/env python3
"""HBM write / copy rates on the box (torch fill_ and copy_), to place the head kernel's 40.9 GB of stores."""
import time, torch
dev = "cuda:0"
n = 10_000_000_000   # 40 GB of fp32
a = torch.empty(n, dtype=torch.float32, device=dev)
for name, fn, nbytes in (("fill 40GB", lambda: a.fill_(1.0), 4 * n),
                         ("fill 4GB", lambda: a[: n // 10].fill_(1.0), 4 * n // 10),
                         ("copy 20GB->20GB", lambda: a[: n // 2].copy_(a[n // 2:]), 4 * n),
                         ("sum 40GB (read)", lambda: a.sum(), 4 * n)):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    print("%-18s %.2f ms  %.2f TB/s" % (name, dt * 1e3, nbytes / dt / 1e12))
