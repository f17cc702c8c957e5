"""ctypes wrapper of oracle/liboracle.so (the OpenMP C restatement, oracle.c).

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
16-bit float arrays are passed as their bit patterns: np.float16 arrays directly, bfloat16 as
np.uint16 bits with dt='bf16'.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_long, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None
_DT = {"f16": 0, "bf16": 1, "i8": 2}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise ImportError(f"{_PATH} missing: run `make -C oracle` (or __graft_entry__.build())")
        _lib = ctypes.CDLL(_PATH)
        _lib.spfa_oracle_threads.restype = c_int
        _lib.spfa_oracle_cent_select.restype = c_int
        _lib.spfa_oracle_cent_select.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p] + \
            [c_int] * 12 + [c_void_p] * 4
        _lib.spfa_oracle_attention.restype = c_int
        _lib.spfa_oracle_attention.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_int, c_long, c_long, c_long,
                                               c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                               c_float, c_void_p]
    return _lib


def threads() -> int:
    return int(lib().spfa_oracle_threads())


def _p(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


def _c(a, dtype=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a) if dtype is None else np.ascontiguousarray(a, dtype=dtype)
    return a


def cent_select(q, l1, block_ids, block_table, seq_len, k=64, block_size=128, max_page_num=256,
                sink_recent=False, max_pages_cap=False, dt="f16", want_indices=True):
    q, l1 = _c(q), _c(l1)
    block_ids, block_table, seq_len = _c(block_ids, np.int32), _c(block_table, np.int32), _c(seq_len, np.int32)
    B = seq_len.shape[0]
    Hq, D = q.shape[1], q.shape[2]
    Hkv, C = l1.shape[0], l1.shape[1]
    P = int(max_page_num)
    idx = np.empty((B, Hq, k), np.int32) if want_indices else None
    pos = np.empty((B, Hq, P), np.int32)
    len8 = np.zeros((B, Hq, 8), np.int32)
    max8 = np.zeros((B, 8), np.int64)
    rc = lib().spfa_oracle_cent_select(_p(q), _p(l1), _DT[dt], _p(block_ids), _p(block_table), _p(seq_len), B, Hq,
                                       Hkv, D, C, int(k), block_ids.shape[1], block_table.shape[1], int(block_size),
                                       P, int(sink_recent), int(max_pages_cap), _p(idx), _p(pos), _p(len8), _p(max8))
    assert rc == 0
    return pos, len8, max8, idx


def attention(q, kcache, vcache, block_table, pos, Hq, Hkv, scale, block_size, kv_layout="BSH", dt="f16",
              kv_dt=None, row_len=None, act_seq=None, token_limit=None, antiq_scale=None, antiq_offset=None):
    q, kcache, vcache = _c(q), _c(kcache), _c(vcache)
    block_table, pos = _c(block_table, np.int32), _c(pos, np.int32)
    kv_dt = kv_dt or dt
    es = 1 if kv_dt == "i8" else 2
    B, D = pos.shape[0], q.shape[-1]
    if kv_layout == "BSH":
        blk, tok, head = block_size * Hkv * D * es, Hkv * D * es, D * es
    else:
        blk, tok, head = Hkv * block_size * D * es, D * es, block_size * D * es
    row_len = _c(row_len, np.int32)
    act_seq = _c(act_seq, np.int64)
    token_limit = _c(token_limit, np.int32)
    antiq_scale, antiq_offset = _c(antiq_scale), _c(antiq_offset)
    out = np.zeros((B, Hq, D), np.float32)
    rc = lib().spfa_oracle_attention(_p(q), _DT[dt], _p(kcache), _p(vcache), _DT[kv_dt], blk, tok, head,
                                     kcache.shape[0], _p(block_table), block_table.shape[1], _p(pos), pos.shape[1],
                                     pos.shape[2], _p(row_len), _p(act_seq), _p(token_limit), _p(antiq_scale),
                                     _p(antiq_offset), B, Hq, Hkv, D, int(block_size), float(scale), _p(out))
    assert rc == 0
    return out


def fusion(q, kcache, vcache, block_table, l1, block_ids, total_seq_len, Hq, Hkv, scale, block_size=128, k=64,
           max_page_num=256, sink_recent=False, max_pages_cap=False, kv_layout="BSH", dt="f16", kv_dt=None,
           antiq_scale=None, antiq_offset=None):
    """cent_select -> attention over each q head's own list (see oracle/ref.py:sparse_paged_fusion_attention)."""
    pos, len8, max8, _ = cent_select(q, l1, block_ids, block_table, total_seq_len, k, block_size, max_page_num,
                                     sink_recent, max_pages_cap, dt, want_indices=False)
    out = attention(q, kcache, vcache, block_table, pos, Hq, Hkv, scale, block_size, kv_layout, dt, kv_dt,
                    row_len=len8[:, :, 0], token_limit=total_seq_len, antiq_scale=antiq_scale,
                    antiq_offset=antiq_offset)
    return out, pos, len8, max8
