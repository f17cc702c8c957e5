#!/usr/bin/env python3
"""Token-level lists on the bench shape: KV bytes/s of sparse_paged_attention(token_level=True) at 1/8 token sparsity for
three selection patterns (isolated random tokens, runs of 16 tokens, whole pages) next to the page-level op on the same
whole pages.  Usage: tools/token_bench.py [bench.py shape flags]; TOKEN_ROWS=kv gives every kv head one list that its
group of q heads shares (the bytes counted are then those of one pass over the list per kv head)."""
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
B, Hq, Hkv, D, bs, S = a.batch, a.heads, a.kv_heads, a.head_dim, a.block_size, a.seqlen
n_tok = S // 8
ROWS = Hkv if os.environ.get("TOKEN_ROWS") == "kv" else Hq
g = torch.Generator(device=dev)
g.manual_seed(5)
e = torch.empty(0, dtype=wl["q"].dtype, device=dev)
ei = torch.empty(0, dtype=torch.int64, device=dev)
asc = wl["a_scale"] if wl["a_scale"] is not None else e
aof = wl["a_off"] if wl["a_off"] is not None else e
kv_elem = 1 if a.kv == "int8" else 2


def pick(units, unit):
    """[B, Hq, n_tok] sorted token indices: a random 1/8 of the `units`-token groups of every (sequence, head)."""
    n_units = S // unit
    sel = torch.rand((B, ROWS, n_units), generator=g, device=dev).argsort(dim=2)[:, :, :n_units // 8].sort(dim=2).values
    tok = (sel.unsqueeze(-1) * unit + torch.arange(unit, device=dev)).reshape(B, ROWS, -1)
    return tok.to(torch.int32).contiguous(), sel.to(torch.int32).contiguous()


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


gbytes = B * ROWS * n_tok * D * kv_elem * 2
for name, unit in (("isolated tokens", 1), ("runs of 16 tokens", 16), ("whole pages", bs)):
    tok, sel = pick(None, unit)
    ms = timed(lambda: custom_ops.sparse_paged_attention(wl["q"], wl["key"], wl["val"], e, e, None, e, e, e, e, e, asc, aof,
                                                         wl["table"], ei, tok, Hq, D ** -0.5, "BNSD", Hkv, bs, 1,
                                                         token_level=True))
    print(f"token-level, {name:18s}: {ms:.3f} ms  {gbytes / ms / 1e6:7.0f} GB/s  ({gbytes / ms / 1e6 / 8000:.3f} of 8 TB/s)")
    if unit == bs:
        pages = torch.full((B, ROWS, 256), 0x7fffffff, dtype=torch.int32, device=dev)
        pages[:, :, :sel.shape[2]] = sel
        ms = timed(lambda: custom_ops.sparse_paged_attention(wl["q"], wl["key"], wl["val"], e, e, None, e, e, e, e, e, asc,
                                                             aof, wl["table"], ei, pages, Hq, D ** -0.5, "BNSD", Hkv, bs, 1))
        print(f"page-level,  the same pages   : {ms:.3f} ms  {gbytes / ms / 1e6:7.0f} GB/s  ({gbytes / ms / 1e6 / 8000:.3f} of 8 TB/s)")
