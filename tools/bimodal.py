#!/usr/bin/env python3
"""Diagnostic: does the launch time of the bench workload depend on where the small buffers (workspace, outputs) or the
KV pool happen to be placed?  Re-allocates them inside one process and times the same launch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cann-ops-adv-vllm_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import custom_ops  # noqa: E402

a = bench.parse()
dev = torch.device("cuda", 0)
B, Hq, Hkv, D, bs = a.batch, a.heads, a.kv_heads, a.head_dim, a.block_size


def timed(wl, tag, hold):
    pos = torch.empty((B, Hq, 256), dtype=torch.int32, device=dev)
    plen = torch.empty((B, Hq, 8), dtype=torch.int32, device=dev)
    pmax = torch.empty((B, 8), dtype=torch.int64, device=dev)
    plan = custom_ops.PreparedFusion(wl["q"], wl["key"], wl["val"], wl["table"], wl["l1"], wl["block_ids"], wl["seq"],
                                     pos, plen, pmax, Hq, D ** -0.5, Hkv, bs, k=a.topk)
    for _ in range(5):
        plan.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        plan.run()
    e1.record()
    torch.cuda.synchronize()
    print(f"{tag}: {e0.elapsed_time(e1) / 40:.4f} ms  ws@{plan._ws.data_ptr():#x} key@{wl['key'].data_ptr():#x} "
          f"val@{wl['val'].data_ptr():#x}", flush=True)
    hold.append((pos, plen, pmax, plan))   # keep them alive: the next round gets different addresses


hold = []
wl = bench.build_workload(a, dev, B, a.seed, custom_ops)
for i in range(4):
    timed(wl, f"same pool, new small buffers #{i}", hold)
for j in range(3):
    pad = torch.empty(((j + 1) * 37) << 20, dtype=torch.uint8, device=dev)   # shifts the next big allocations
    hold.append(pad)
    del wl
    torch.cuda.empty_cache()
    wl = bench.build_workload(a, dev, B, a.seed, custom_ops)
    for i in range(2):
        timed(wl, f"pool #{j + 1}, run {i}", hold)
