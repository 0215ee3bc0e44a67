#!/usr/bin/env python3
"""host -> device copy rate of the box (pinned memory): the ceiling of the work-list hand-over.  Copies of the size of one 4K work
list (4 MiB) against copies of a batch of 32 of them (128 MiB), from 1 and 4 host threads / streams, with distinct source buffers
per copy as the hand-over has them."""
import threading
import time

import torch


def run(size, threads, total=2 << 30):
    n_copies = max(total // size // threads, 1)
    n_buf = min(n_copies, max(1, (512 << 20) // size // threads))
    host = [[torch.empty(size, dtype=torch.uint8, pin_memory=True) for _ in range(n_buf)] for _ in range(threads)]
    dev = [[torch.empty(size, dtype=torch.uint8, device="cuda") for _ in range(n_buf)] for _ in range(threads)]
    streams = [torch.cuda.Stream() for _ in range(threads)]

    def work(i):
        with torch.cuda.stream(streams[i]):
            for k in range(n_copies):
                dev[i][k % n_buf].copy_(host[i][k % n_buf], non_blocking=True)
    best = 0.0
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ths = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        torch.cuda.synchronize()
        best = max(best, threads * n_copies * size / (time.perf_counter() - t0) / 1e9)
    return best


for size in (1 << 20, 4 << 20, 16 << 20, 128 << 20):
    for threads in (1, 4):
        print(f"{size >> 20:4d} MiB copies, {threads} thread(s)/stream(s): {run(size, threads):6.1f} GB/s host->device", flush=True)
