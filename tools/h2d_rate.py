#!/usr/bin/env python3
"""host -> device copy rate of the box (pinned memory, 1 and 3 concurrent streams): the ceiling of the work-list hand-over"""
import time
import torch
n = 256 << 20
host = [torch.empty(n, dtype=torch.uint8, pin_memory=True) for _ in range(3)]
dev = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(3)]
for k in (1, 3):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k):
            with torch.cuda.stream(streams[i]):
                for _ in range(4):
                    dev[i].copy_(host[i], non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{k} stream(s): {k * 4 * n / dt / 1e9:.1f} GB/s host->device (pinned, 256 MiB copies)")
small = torch.empty(5 << 20, dtype=torch.uint8, pin_memory=True)
dsm = torch.empty(5 << 20, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    dsm.copy_(small, non_blocking=True)
torch.cuda.synchronize()
print(f"5 MiB copies back to back: {200 * (5 << 20) / (time.perf_counter() - t0) / 1e9:.1f} GB/s")
