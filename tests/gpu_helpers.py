"""Helpers for the GPU parity tests: numpy <-> device tensors and op call shims."""
import numpy as np
import torch

import custom_ops

DEV = "cuda:0"


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def empty(dtype=torch.float16):
    return torch.empty(0, dtype=dtype, device=DEV)


def f32(t):
    return t.detach().float().cpu().numpy()


def _opt(t, e):
    return t if t is not None else e


def call_attention(q, k, v, block_table, block_position, actual_seq, Hq, Hkv, scale, bs, layout="BNSD",
                   antiq_scale=None, antiq_offset=None, out_dtype=None, pse=None, mask=None, quant_scale2=None,
                   quant_offset2=None, **extra):
    e = empty(q.dtype)
    return custom_ops.sparse_paged_attention(
        q, k, v, _opt(pse, e), _opt(mask, e), actual_seq, e, e, e, _opt(quant_scale2, e), _opt(quant_offset2, e),
        _opt(antiq_scale, e), _opt(antiq_offset, e),
        block_table, empty(torch.int64), block_position, Hq, scale, layout, Hkv, bs, 1, out_dtype=out_dtype, **extra)


def call_fusion(q, k, v, block_table, l1, block_ids, total_seq_len, Hq, Hkv, scale, bs, topk, P=256,
                antiq_scale=None, antiq_offset=None, out_dtype=None, sink_recent=False, max_pages_cap=False,
                pse=None, mask=None, quant_scale2=None, quant_offset2=None, **extra):
    e = empty(q.dtype)
    B = total_seq_len.shape[0]
    pos = torch.empty((B, Hq, P), dtype=torch.int32, device=DEV)
    plen = torch.empty((B, Hq, 8), dtype=torch.int32, device=DEV)
    pmax = torch.empty((B, 8), dtype=torch.int64, device=DEV)
    out, pos2, pmax2 = custom_ops.sparse_paged_fusion_attention(
        q, k, v, _opt(pse, e), _opt(mask, e), [1], e, e, e, _opt(quant_scale2, e), _opt(quant_offset2, e),
        _opt(antiq_scale, e), _opt(antiq_offset, e),
        block_table, empty(torch.int64), l1, block_ids, total_seq_len, pos, plen, pmax,
        Hq, scale, "BNSD", Hkv, bs, 1, k=topk, out_dtype=out_dtype, sink_recent=sink_recent,
        max_pages_cap=max_pages_cap, **extra)
    assert pos2 is pos and pmax2 is pmax
    return out, pos, plen, pmax
