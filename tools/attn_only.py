#!/usr/bin/env python3
"""The standalone attention op on the page lists the fused op selects for the bench shape (no selection in the launch):
what the page stream alone costs next to the fused launch.  Usage: tools/attn_only.py [bench.py shape flags]"""
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
wl = bench.build_workload(a, dev, a.batch, a.seed, custom_ops)
B, Hq, Hkv, D, bs = a.batch, a.heads, a.kv_heads, a.head_dim, a.block_size
pos = torch.empty((B, Hq, 256), dtype=torch.int32, device=dev)
plen = torch.empty((B, Hq, 8), dtype=torch.int32, device=dev)
pmax = torch.empty((B, 8), dtype=torch.int64, device=dev)
plan = custom_ops.PreparedFusion(wl["q"], wl["key"], wl["val"], wl["table"], wl["l1"], wl["block_ids"], wl["seq"],
                                 pos, plen, pmax, Hq, D ** -0.5, Hkv, bs, k=a.topk, antiquant_scale=wl["a_scale"],
                                 antiquant_offset=wl["a_off"])
plan.run()
torch.cuda.synchronize()
e = torch.empty(0, dtype=wl["q"].dtype, device=dev)
ei = torch.empty(0, dtype=torch.int64, device=dev)
asc = wl["a_scale"] if wl["a_scale"] is not None else e
aof = wl["a_off"] if wl["a_off"] is not None else e
act = pmax[:, 0].contiguous()
pages = int(plen[:, :, 0].sum().item()) if Hkv == Hq else None
if Hkv != Hq:
    # a group streams the UNION of its heads' lists in the fused op: hand the standalone op that union as one list per kv head
    G = Hq // Hkv
    SENT = 0x7FFFFFFF
    flat = pos.view(B, Hkv, G * 256).to(torch.int64)
    srt = flat.sort(dim=2).values
    dup = torch.zeros_like(srt, dtype=torch.bool)
    dup[:, :, 1:] = srt[:, :, 1:] == srt[:, :, :-1]
    srt[dup] = SENT
    srt = srt.sort(dim=2).values
    n = int((srt != SENT).sum(dim=2).max().item())
    pos = srt[:, :, :max(n, 1)].to(torch.int32).contiguous()
    pages = int((pos != SENT).sum().item())
    act = torch.full((B,), pos.shape[2] * bs, dtype=torch.int64, device=dev)


def attn():
    return custom_ops.sparse_paged_attention(wl["q"], wl["key"], wl["val"], e, e, act, e, e, e, e, e, asc, aof, wl["table"], ei,
                                             pos, Hq, D ** -0.5, "BNSD", Hkv, bs, 1)


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


t_attn, t_fused = timed(attn), timed(plan.run)
line = f"attention on given lists {t_attn:.4f} ms, fused {t_fused:.4f} ms (launch included, back to back)"
if pages is not None:
    nbytes = pages * bs * D * (1 if a.kv == "int8" else 2) * 2
    line += f"; {nbytes / t_attn / 1e6:.0f} / {nbytes / t_fused / 1e6:.0f} GB/s"
print(line)
