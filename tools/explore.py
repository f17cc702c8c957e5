#!/usr/bin/env python3
"""Kernel exploration on the bench workload: fused vs its parts, KV layouts, batch sizes.
Not part of the product or of the driver's contract; prints one line per variant."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cann-ops-adv-vllm_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import custom_ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    sys.argv = [sys.argv[0]] + sys.argv[1:]
    a = bench.parse()
    dev = torch.device("cuda", 0)
    wl = bench.build_workload(a, dev, a.batch, a.seed, custom_ops)
    B, Hq, Hkv, D, bs = a.batch, a.heads, a.kv_heads, a.head_dim, a.block_size
    pos = torch.empty((B, Hq, 256), dtype=torch.int32, device=dev)
    plen = torch.empty((B, Hq, 8), dtype=torch.int32, device=dev)
    pmax = torch.empty((B, 8), dtype=torch.int64, device=dev)
    plan = custom_ops.PreparedFusion(wl["q"], wl["key"], wl["val"], wl["table"], wl["l1"], wl["block_ids"], wl["seq"],
                                     pos, plen, pmax, Hq, D ** -0.5, Hkv, bs, antiquant_scale=wl["a_scale"],
                                     antiquant_offset=wl["a_off"], k=a.topk)
    t_fused = timeit(plan.run)
    kv_elem = 1 if a.kv == "int8" else 2
    pages, gbytes = bench.gather_bytes(pos, plen, Hkv, Hq // Hkv, wl["page_len"], bs, D, kv_elem)
    print(f"fused            {t_fused*1e3:8.1f} us  {gbytes/t_fused/1e6:8.1f} GB/s  pages={pages}")
    t_cs = timeit(lambda: custom_ops.cent_select(wl["q"], wl["l1"], wl["block_ids"], wl["table"], wl["seq"], k=a.topk))
    print(f"cent_select      {t_cs*1e3:8.1f} us")
    e = torch.empty(0, dtype=wl["q"].dtype, device=dev)
    ei = torch.empty(0, dtype=torch.int64, device=dev)
    asc = wl["a_scale"] if wl["a_scale"] is not None else e
    aof = wl["a_off"] if wl["a_off"] is not None else e
    act = pmax[:, 0].contiguous()

    def attn(k, v):
        return custom_ops.sparse_paged_attention(wl["q"], k, v, e, e, act, e, e, e, e, e, asc, aof, wl["table"], ei,
                                                 pos, Hq, D ** -0.5, "BNSD", Hkv, bs, 1)
    try:
        t_at = timeit(lambda: attn(wl["key"], wl["val"]))
        print(f"attention BSH    {t_at*1e3:8.1f} us  {gbytes/t_at/1e6:8.1f} GB/s")
    except RuntimeError as ex:   # exploration builds hold one instantiation only
        print("attention BSH    n/a:", str(ex)[:60])
    G = Hq // Hkv
    if G > 1:
        # the group's union list as a shared [B,Hkv,P] block_position: the streaming floor of the fused op
        Pn = 256 * G
        valid = torch.arange(256, device=dev)[None, None, :] < plen[:, :, :1]
        hit = torch.zeros((B, Hkv, wl["page_len"] + 1), dtype=torch.bool, device=dev)
        idx = torch.where(valid, pos.long(), torch.full_like(pos, wl["page_len"]).long()).reshape(B, Hkv, G * 256)
        hit.scatter_(2, idx, True)
        hit = hit[:, :, :wl["page_len"]]
        order = torch.argsort((~hit).to(torch.int8), dim=2, stable=True)[:, :, :Pn].to(torch.int32)
        n = hit.sum(dim=2, keepdim=True)
        upos = torch.where(torch.arange(order.shape[2], device=dev)[None, None, :] < n, order,
                           torch.full_like(order, 0x7fffffff)).contiguous()
        act_u = (n.max(dim=1).values.reshape(B).long() * bs).contiguous()
        t_u = timeit(lambda: custom_ops.sparse_paged_attention(wl["q"], wl["key"], wl["val"], e, e, act_u, e, e, e, e, e,
                                                               asc, aof, wl["table"], ei, upos, Hq, D ** -0.5, "BNSD",
                                                               Hkv, bs, 1))
        print(f"attention union  {t_u*1e3:8.1f} us  {gbytes/t_u/1e6:8.1f} GB/s")
    if os.environ.get("EXPLORE_BNSD", "1") == "1":
        kb = wl["key"].permute(0, 2, 1, 3).contiguous()
        vb = wl["val"].permute(0, 2, 1, 3).contiguous()
        t_bn = timeit(lambda: attn(kb, vb))
        print(f"attention BNSD   {t_bn*1e3:8.1f} us  {gbytes/t_bn/1e6:8.1f} GB/s")


if __name__ == "__main__":
    main()
