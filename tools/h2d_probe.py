#!/usr/bin/env python3
"""Runs on the GPU box: H2D copy rate of a pinned 4 MB uint8 buffer on a side stream, as Model._upload issues it --
alone, and while a stream of kernels keeps the GPU busy on the default stream."""
import time
import numpy as np
import torch

dev = torch.device("cuda:0")
n = 4 * 1024 * 1024
host = torch.from_numpy(np.empty(n, np.uint8)).pin_memory()
print("pinned:", host.is_pinned())
d = torch.empty(n, dtype=torch.uint8, device=dev)
s = torch.cuda.Stream(device=dev)
for busy in (False, True):
    a = torch.randn(4096, 4096, device=dev)
    torch.cuda.synchronize()
    ts = []
    for it in range(20):
        if busy:
            for _ in range(4):
                a = a @ a * 1e-4
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        with torch.cuda.stream(s):
            e0.record(s); d.copy_(host, non_blocking=True); e1.record(s)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        ts.append((e0.elapsed_time(e1), (t1 - t0) * 1e3))
    ev = sorted(t[0] for t in ts)[len(ts) // 2]; hostms = sorted(t[1] for t in ts)[len(ts) // 2]
    print(f"busy={busy}: copy {ev:.3f} ms on the stream ({n / ev / 1e6:.1f} GB/s), host call {hostms:.3f} ms")
