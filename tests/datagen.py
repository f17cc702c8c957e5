"""Seeded synthetic inputs shared by the golden generator, the tests and bench.py.

Everything is numpy (PCG64 streams are stable across platforms) so the same seed
gives the same bytes in the build container and on the GPU box; every golden
fixture also stores a sha256 of its inputs to detect drift.

Two families, following the reference's own tests:
  * exact-grid fp16 (values n/256, n in [0,256)): every fp32 summation order of a
    D<=128 dot product is exact, so top-k results are reproducible bit for bit on
    any implementation.
  * reference distributions: q, K, V = rand*(1-1e-3)+1e-3, centroids = randn
    (src/transformer/sparse_paged_fusion_attention/test/fused_ifa_v5_case.py:53-61),
    aliasing block_table = randint(0, kvPageLen) and block_ids = randint(0, C)
    (tests/select_position.py:42-45).
"""
from __future__ import annotations

import hashlib

import numpy as np

SENTINEL = 0x7FFFFFFF


def rng_for(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(seed))


def grid_f16(rng, shape) -> np.ndarray:
    return (rng.integers(0, 256, size=shape).astype(np.float32) / 256.0).astype(np.float16)


def ref_rand_f16(rng, shape) -> np.ndarray:
    eps = 1e-3
    return (rng.random(size=shape, dtype=np.float32) * (1.0 - eps) + eps).astype(np.float16)


def randn_f16(rng, shape) -> np.ndarray:
    return rng.standard_normal(size=shape, dtype=np.float32).astype(np.float16)


def sha256_of(*arrays) -> str:
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode())
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def cent_case(seed: int, B: int, Hq: int, Hkv: int, D: int, C: int, grid: bool = True):
    """q [B,Hq,D] f16 and l1_cent [Hkv,C,D] f16 (tests/compute_cent.py:16-50 shapes)."""
    rng = rng_for(seed)
    if grid:
        return grid_f16(rng, (B, Hq, D)), grid_f16(rng, (Hkv, C, D))
    return ref_rand_f16(rng, (B, Hq, D)), randn_f16(rng, (Hkv, C, D))


def page_case(seed: int, Hkv: int, C: int, kv_page_len: int, max_batch: int, max_page: int,
              seq_lens, aliasing: bool = True):
    """block_ids [Hkv,kvPageLen] i32, block_table [maxBatch,maxPage] i32, seq_len [B] i32.

    aliasing=True : block_table = randint(0, kvPageLen) (reference style, pages repeat).
    aliasing=False: every (batch, logical page) gets its own physical page.
    """
    rng = rng_for(seed + 7919)
    block_ids = rng.integers(0, C, size=(Hkv, kv_page_len)).astype(np.int32)
    if aliasing:
        block_table = rng.integers(0, kv_page_len, size=(max_batch, max_page)).astype(np.int32)
    else:
        assert max_batch * max_page <= kv_page_len
        block_table = rng.permutation(kv_page_len)[: max_batch * max_page].reshape(
            max_batch, max_page).astype(np.int32)
    seq_len = np.asarray(seq_lens, dtype=np.int32)
    return block_ids, block_table, seq_len


def indices_case(seed: int, B: int, Hq: int, k: int, C: int) -> np.ndarray:
    """Random top-k ids (with repeats), tests/select_position.py:45."""
    rng = rng_for(seed + 104729)
    return rng.integers(0, C, size=(B, Hq, k)).astype(np.int32)


def kv_case(seed: int, n_blocks: int, block_size: int, Hkv: int, D: int, layout: str = "BSH",
            dist: str = "ref", dtype: str = "f16"):
    """K and V caches.  BSH = [blocks, bs, Hkv, D], BNSD = [blocks, Hkv, bs, D]."""
    rng = rng_for(seed + 15485863)
    shape = (n_blocks, block_size, Hkv, D) if layout == "BSH" else (n_blocks, Hkv, block_size, D)
    if dtype == "i8":
        k = rng.integers(-127, 128, size=shape).astype(np.int8)
        v = rng.integers(-127, 128, size=shape).astype(np.int8)
        return k, v
    if dist == "ref":
        return ref_rand_f16(rng, shape), ref_rand_f16(rng, shape)
    return randn_f16(rng, shape), randn_f16(rng, shape)


def antiquant_case(seed: int, Hkv: int, D: int, with_offset: bool = True):
    """antiquant_scale / antiquant_offset [2,Hkv,D] f16 (index 0 key, 1 value)."""
    rng = rng_for(seed + 32452843)
    scale = (rng.random(size=(2, Hkv, D), dtype=np.float32) * 0.02 + 0.001).astype(np.float16)
    off = randn_f16(rng, (2, Hkv, D)) if with_offset else None
    return scale, off


def boundary_gap(scores: np.ndarray, k: int) -> float:
    """Smallest gap between the k-th and (k+1)-th largest score over all rows."""
    s = np.sort(scores.astype(np.float64), axis=-1)[..., ::-1]
    if k >= s.shape[-1]:
        return float("inf")
    return float((s[..., k - 1] - s[..., k]).min())


def has_tie_in_top(scores: np.ndarray, k: int) -> np.ndarray:
    """Per row: does any pair among the k+1 largest scores tie exactly?"""
    s = np.sort(scores.astype(np.float64), axis=-1)[..., ::-1]
    top = s[..., : min(k + 1, s.shape[-1])]
    return (np.diff(top, axis=-1) == 0).any(axis=-1)
