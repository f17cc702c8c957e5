#!/usr/bin/env python3
"""Diagnostic: what the per-step events of bench.py cost.  Times the prepared fused launch with an event in front of every
step and with events around the whole run only, alternating."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cann-ops-adv-vllm_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import custom_ops  # noqa: E402

a = bench.parse()
dev = torch.device("cuda", 0)
wl = bench.build_workload(a, dev, a.batch, a.seed, custom_ops)
B, Hq, Hkv, D, bs = a.batch, a.heads, a.kv_heads, a.head_dim, a.block_size
pos = torch.empty((B, Hq, 256), dtype=torch.int32, device=dev)
plen = torch.empty((B, Hq, 8), dtype=torch.int32, device=dev)
pmax = torch.empty((B, 8), dtype=torch.int64, device=dev)
plan = custom_ops.PreparedFusion(wl["q"], wl["key"], wl["val"], wl["table"], wl["l1"], wl["block_ids"], wl["seq"],
                                 pos, plen, pmax, Hq, D ** -0.5, Hkv, bs, k=a.topk, antiquant_scale=wl["a_scale"],
                                 antiquant_offset=wl["a_off"])
for _ in range(5):
    plan.run()
torch.cuda.synchronize()
n = a.steps
for rep in range(4):
    for per_step in (True, False):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        evs[0].record()
        for i in range(n):
            if per_step and i:
                evs[i].record()
            plan.run()
        evs[n].record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / n * 1e3
        print(f"per-step events {per_step}: {evs[0].elapsed_time(evs[n]) / n:.4f} ms per launch by events, {wall:.4f} ms wall")
