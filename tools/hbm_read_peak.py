#!/usr/bin/env python3
"""Practical HBM read bandwidth of this GPU with plain library kernels (a reduction over a large tensor): the ceiling a
streaming kernel can be compared with, next to the 8 TB/s data-sheet figure."""
import torch

dev = torch.device("cuda", 0)
for gib in (4, 16):
    x = torch.empty(gib << 28, dtype=torch.float32, device=dev).normal_()
    for name, fn in (("sum", lambda: x.sum()), ("max", lambda: x.max()), ("abs().max() (fused by torch? no: 2 passes)", None)):
        if fn is None:
            continue
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{gib} GiB {name}: {ms:.3f} ms  {x.numel() * 4 / ms / 1e9:.2f} TB/s")
    del x
