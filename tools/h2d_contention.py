#!/usr/bin/env python3
"""host -> device copy rate while other host threads copy pageable -> pinned memory (what the hand-over does): is the DMA slowed by
the host's own memory traffic?"""
import threading
import time

import numpy as np
import torch

SIZE = 4 << 20


def run(copy_streams, memcpy_threads, seconds=2.0):
    host = [[torch.empty(SIZE, dtype=torch.uint8, pin_memory=True) for _ in range(16)] for _ in range(copy_streams)]
    dev = [[torch.empty(SIZE, dtype=torch.uint8, device="cuda") for _ in range(16)] for _ in range(copy_streams)]
    streams = [torch.cuda.Stream() for _ in range(copy_streams)]
    src = [np.random.randint(0, 255, SIZE, dtype=np.uint8) for _ in range(memcpy_threads)]
    dst = [torch.empty(SIZE, dtype=torch.uint8, pin_memory=True).numpy() for _ in range(memcpy_threads)]
    stop = threading.Event()
    copied = [0] * memcpy_threads
    moved = [0] * copy_streams

    def memcpy_worker(i):
        while not stop.is_set():
            np.copyto(dst[i], src[i])
            copied[i] += SIZE

    def dma_worker(i):
        with torch.cuda.stream(streams[i]):
            k = 0
            while not stop.is_set():
                for _ in range(8):
                    dev[i][k % 16].copy_(host[i][k % 16], non_blocking=True)
                    k += 1
                streams[i].synchronize()
                moved[i] += 8 * SIZE
    ths = [threading.Thread(target=memcpy_worker, args=(i,)) for i in range(memcpy_threads)] + [threading.Thread(target=dma_worker, args=(i,)) for i in range(copy_streams)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    time.sleep(seconds)
    stop.set()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    return sum(moved) / dt / 1e9, sum(copied) / dt / 1e9


for cs, mt in ((4, 0), (4, 2), (4, 4), (4, 8), (1, 0), (1, 4)):
    dma, mc = run(cs, mt)
    print(f"{cs} copy stream(s) of 4 MiB copies + {mt} host memcpy thread(s): DMA {dma:5.1f} GB/s, host memcpy {mc:5.1f} GB/s", flush=True)
