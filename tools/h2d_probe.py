#!/usr/bin/env python3
"""Host->device copy bandwidth on this box: pinned vs pageable, by size."""
import time
import torch

for mb in (0.0528, 1, 6.7, 64):
    n = int(mb * 1e6)
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    for kind in ("pinned", "pageable"):
        h = torch.empty(n, dtype=torch.uint8)
        h.fill_(3)
        if kind == "pinned":
            h = h.pin_memory()
        for _ in range(3):
            d.copy_(h, non_blocking=True)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            d.copy_(h, non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 10
        print(f"{mb:8.3f} MB {kind:9s} {dt * 1e3:8.3f} ms  {n / dt / 1e9:6.2f} GB/s", flush=True)
    back = torch.empty(n, dtype=torch.uint8).pin_memory()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        back.copy_(d, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 10
    print(f"{mb:8.3f} MB D2H pinned {dt * 1e3:8.3f} ms  {n / dt / 1e9:6.2f} GB/s", flush=True)
