"""CPU oracle (numpy) for the sparse paged fusion attention chain.

TEST INFRASTRUCTURE ONLY.  Nothing under ``cann-ops-adv-vllm_amd/`` imports this
module; it is the checker used by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.

Every function restates the algorithm of one reference function and cites it.
The reference's only CPU implementation is ``tests/utils.py`` (torch); the
integer/index stages below are pinned against golden vectors produced by
importing that file (``tests/golden/gen_golden.py``).  The reference has no
CPU attention at all, so ``sparse_paged_attention`` / the fused op restate the
device kernel's arithmetic; beyond the reference's constant-V known-answer
tests their numerics are "parity unpinned".

Pinned:   compute_cent, select_position, cent_select (golden vectors from
          reference tests/utils.py:4-123).
Unpinned: attention numerics (no reference CPU path exists); pinned only by
          the reference's known-answer cases (tests/ifa_v5_case.py:584-765)
          and the fused == separate identity
          (src/transformer/sparse_paged_fusion_attention/test/fused_ifa_v5_case.py:215-276).
"""
from __future__ import annotations

import numpy as np

SENTINEL = 0x7FFFFFFF  # reference pad value, cent_select.cpp:406 / tests/utils.py:32


# --------------------------------------------------------------------------- #
# small dtype helpers (numpy has no bfloat16)
# --------------------------------------------------------------------------- #
def round_to_bf16(x: np.ndarray) -> np.ndarray:
    """Round fp32 values to the nearest bf16 (ties to even), returned as fp32."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    nan = np.isnan(x)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    out = (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)
    out = np.where(nan, np.float32(np.nan), out)
    return out.reshape(np.shape(x))


def round_to_f16(x: np.ndarray) -> np.ndarray:
    return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)


def bf16_bits_to_f32(bits: np.ndarray) -> np.ndarray:
    return (np.asarray(bits, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    return (round_to_bf16(x).view(np.uint32) >> 16).astype(np.uint16)


# --------------------------------------------------------------------------- #
# compute_cent  (reference: tests/utils.py:87-123; kernel compute_cent.cpp:143-245)
# --------------------------------------------------------------------------- #
def cent_scores(q: np.ndarray, l1_cent: np.ndarray) -> np.ndarray:
    """fp32 scores[b, h, c] = q[b, h, :] . l1_cent[h // G, c, :].

    The reference casts both operands to float and matmuls (utils.py:117); the
    device kernel multiplies fp16 pairs and accumulates in fp32
    (compute_cent.cpp:279-340).  Accumulation here is float64 rounded once to
    fp32: on the exact-grid parity data every summation order gives the same
    value, on other data this is the correctly rounded score.
    """
    q = np.asarray(q)
    l1 = np.asarray(l1_cent)
    B, Hq, D = q.shape
    Hkv, C, D2 = l1.shape
    assert D == D2 and Hq % Hkv == 0
    G = Hq // Hkv
    qf = q.astype(np.float64).reshape(B, Hkv, G, D)
    lf = l1.astype(np.float64)
    s = np.einsum("bngd,ncd->bngc", qf, lf, optimize=True)
    return s.reshape(B, Hq, C).astype(np.float32)


def topk_desc(scores: np.ndarray, k: int) -> np.ndarray:
    """Indices of the k largest entries along the last axis, descending; equal
    scores are ordered by ascending index (the HIP kernel's defined tie-break;
    torch.topk, utils.py:121, leaves tie order unspecified)."""
    order = np.argsort(-scores.astype(np.float64), axis=-1, kind="stable")
    return order[..., :k].astype(np.int32)


def compute_cent(q, l1_cent, k: int = 64) -> np.ndarray:
    """reference: tests/utils.py:87-123 (k fixed to 64 there, :101)."""
    return topk_desc(cent_scores(q, l1_cent), k)


# --------------------------------------------------------------------------- #
# select_position  (reference: tests/utils.py:4-84; kernel select_position.cpp:122-183,
#                   cent_select.cpp:340-425 for the sink/recent and cap variants)
# --------------------------------------------------------------------------- #
def select_position(block_ids, block_table, seq_len, indices, block_size: int = 128,
                    max_page_num: int | None = None, sink_recent: bool = False,
                    max_pages_cap: bool = False):
    """Returns (page_position [B,Hq,P], page_position_length [B,Hq], gathered ids [B,Hq,maxPL]).

    P = max_page_num if given (device convention, add_custom.cpp:958) else the
    batch max page count (utils.py:45-48).  Positions beyond P are dropped and
    the length is clipped to P.

    sink_recent=True adds cent_select.cpp:350-362: every logical page whose
    block_table VALUE equals that of logical page 0, pageLen-2 or pageLen-1.
    max_pages_cap=True clips the length to pageLen // 8 (cent_select.cpp:218,410-414);
    the positions themselves are not truncated by the cap.
    """
    block_ids = np.asarray(block_ids)
    block_table = np.asarray(block_table)
    seq_len = np.asarray(seq_len)
    indices = np.asarray(indices)
    B = seq_len.shape[0]                      # utils.py:34 (batch comes from seq_len)
    Hkv, kv_page_len = block_ids.shape
    Hq, K = indices.shape[1], indices.shape[2]
    assert Hq % Hkv == 0
    G = Hq // Hkv
    page_lens = (seq_len.astype(np.int64) + (block_size - 1)) // block_size   # utils.py:44
    max_pl = int(page_lens.max()) if B > 0 else 0
    P = max_pl if max_page_num is None else int(max_page_num)
    pos = np.full((B, Hq, P), SENTINEL, dtype=np.int32)
    length = np.zeros((B, Hq), dtype=np.int32)
    c_full = np.zeros((B, Hq, max_pl), dtype=np.int32)
    for b in range(B):
        pl = int(page_lens[b])
        if pl <= 0:
            continue
        bt = block_table[b, :pl].astype(np.int64)
        page_idx = np.clip(bt, 0, kv_page_len - 1)                           # utils.py:62
        c_kv = block_ids[:, page_idx]                                        # [Hkv, pl]
        extra = None
        if sink_recent:
            extra = (bt == bt[0]) | (bt == bt[pl - 1])
            if pl >= 2:
                extra |= (bt == bt[pl - 2])
        for h in range(Hq):
            c = c_kv[h // G]
            c_full[b, h, :pl] = c
            mask = np.isin(c, indices[b, h])                                 # utils.py:70
            if extra is not None:
                mask = mask | extra
            hit = np.nonzero(mask)[0]
            n = int(hit.size)
            w = min(n, P)
            pos[b, h, :w] = hit[:w]
            if max_pages_cap:
                n = min(n, pl // 8)
            length[b, h] = min(n, P)
    return pos, length, c_full


def cent_select(q, l1_cent, block_ids, block_table, seq_len, k: int = 64, block_size: int = 128,
                max_page_num: int = 256, sink_recent: bool = False, max_pages_cap: bool = False):
    """compute_cent -> select_position -> per-batch max (reference composition:
    tests/cent_select.py:74-98; kernel cent_select.cpp:101-161).

    Returns (page_position [B,Hq,P] i32, page_position_length [B,Hq,8] i32 replicated,
    max_page_position_length [B,8] i64 replicated = max_h(len) * block_size).
    """
    seq_len = np.asarray(seq_len)
    B = seq_len.shape[0]
    idx = compute_cent(np.asarray(q)[:B], l1_cent, k)
    pos, length, _ = select_position(block_ids, block_table, seq_len, idx, block_size,
                                     max_page_num, sink_recent, max_pages_cap)
    len8 = np.repeat(length[:, :, None], 8, axis=2).astype(np.int32)          # cent_select.cpp:418
    mx = length.max(axis=1).astype(np.int64) * block_size if length.size else np.zeros((B,), np.int64)
    max8 = np.repeat(mx[:, None], 8, axis=1).astype(np.int64)                 # cent_select.cpp:143-151
    return pos, len8, max8, idx


# --------------------------------------------------------------------------- #
# sparse paged attention (restates the device kernel; no reference CPU path)
# --------------------------------------------------------------------------- #
def _kv_rows(cache: np.ndarray, blk: int, n2: int, layout: str, Hkv: int, D: int) -> np.ndarray:
    """[block_size, D] rows of kv-head n2 in physical block blk.

    BSH : [blocks, bs, Hkv*D] or [blocks, bs, Hkv, D]  (offset blk*bs*Hkv*D + n2*D,
          sparse_paged_fusion_attention_split_Bbn2s2_Us2.h:262,391-393)
    BNSD: [blocks, Hkv, bs, D]
    """
    if layout == "BSH":
        c = cache.reshape(cache.shape[0], cache.shape[1], Hkv, D)
        return c[blk, :, n2, :]
    c = cache.reshape(cache.shape[0], Hkv, -1, D)
    return c[blk, n2, :, :]


def sparse_paged_attention(q, key_cache, value_cache, block_table, block_position,
                           actual_seq_lengths, num_heads: int, num_kv_heads: int,
                           scale: float, block_size: int, kv_layout: str = "BSH",
                           antiquant_scale=None, antiquant_offset=None,
                           token_limit=None, p_round: str | None = None,
                           exact: bool = True, pse_shift=None, atten_mask=None,
                           quant_scale2=None, quant_offset2=None, token_level: bool = False,
                           key_shared_prefix=None, value_shared_prefix=None, shared_prefix_len=None,
                           return_lse: bool = False, antiquant_mode: int = 0):
    """Decode attention over a virtual sequence of selected pages.  Returns fp32 [B,Hq,D].

    Virtual token t of batch b / head row r lives in virtual page j = t // bs; its
    logical page is p = block_position[b, r, j] and its physical block
    block_table[b, p] (…Us2.h:256-294).  block_position may have Hq rows (each q head
    its own list) or Hkv rows (shared by the group).  A head's list ends at its first
    0x7fffffff entry (the kernel `break`s there, …Us2.h:308-311); sentinel slots
    contribute nothing.  actual_seq_lengths[b] bounds the virtual sequence (None: whole list).

    token_limit[b] (optional): tokens whose LOGICAL index p*bs + i >= token_limit[b] are
    masked (used by the fused op for a partial last page).

    Arithmetic (…Us2.h:3363 scale, :3408-3419 online softmax, :3567-3568 P cast,
    :3987-3999 rescale / divide):
      exact=True : float64 two-pass softmax (the mathematically exact answer).
      exact=False: per-page online softmax in fp32, running max starting at -2e38
                   (:859,4472), P rounded to p_round ('f16' | 'bf16' | None) before P.V.
    int8 KV (…Us2.h:2647,2651,4160-4174): q' = q * scale_k (per channel), key offset
    dropped (softmax is shift invariant), out = (P.V_i8 / l + offset_v) * scale_v.
    antiquant_scale/offset: [2, Hkv*D] (0 = key, 1 = value; …Us2.h:2035-2043), or [2]: one
    key and one value parameter for the whole tensor (:2036-2038).

    pse_shift [B | 1, Hq, 1, S] is added to the scaled scores and atten_mask [B, S]
    (non zero = masked) then replaces them by -1e12 (…Us2.h:3363-3404, :860); both are
    indexed by the COLUMN of the gathered sequence, j * block_size + i for token i of the
    j-th list slot (:2192-2194, 2239-2244).

    token_level=True: block_position holds logical TOKEN indices (the variant sketched in the
    reference's tests/ifa_v5_case_tkp.py:53-54): entry j < actual_seq_lengths[b] before the first
    0x7fffffff is token t, row t % bs of page block_table[b, t // bs].

    antiquant_mode=1 (per token, …Us2.h:2039-2043, 3711-3822): antiquant_scale / antiquant_offset are float32
    [2, B, S] over the columns of the gathered sequence (as pse_shift / atten_mask): K_t = (k_t + off[0, b, c]) *
    scale[0, b, c], V_t = (v_t + off[1, b, c]) * scale[1, b, c].  exact mode.
    antiquant_mode=2 (per tensor and head, …_tiling_check.cc:488-497): [2, Hkv] in the query dtype, one key and one value
    parameter per kv head.  antiquant_mode=3 (per token and head, :466-487, …Us2.h:2203-2206): float32 [2, B, Hkv, S].

    key_shared_prefix / value_shared_prefix ([1, S, Hkv * D] | [1, S, Hkv, D] for a BSH cache, [1, Hkv, S, D] for
    BNSD; the KV dtype, int8 with the same antiquant parameters): every row attends the first shared_prefix_len
    (default S) prefix tokens in front of its gathered pages; pse_shift / atten_mask columns then start with the
    prefix tokens (…Us2.h:1418-1439, 2263-2266, 2283-2285).

    return_lse=True: also returns the rows' softmax log-sum-exp [B, Hq] float32, ln(sum_t exp(score_t)) with the
    scores as the formula has them — for int8 KV including the key offset the kernel drops from its products and
    adds back here (…Us2.h:2651-2659, 3130-3159); -3.4e38 for a row that attends nothing (:1289-1300).  exact mode.

    quant_scale2 / quant_offset2 (scalar or [Hq * D]): int8 output,
    out * scale2 + offset2 in fp32, cast to f16 and then to int8, both with ties away from
    zero (…Us2.h:4016-4111); the result is returned as float32 holding the int8 values.
    """
    q = np.asarray(q, dtype=np.float32)
    B, Hq, D = q.shape
    Hkv = num_kv_heads
    G = Hq // Hkv
    bp = np.asarray(block_position)
    rows = bp.shape[1]
    assert rows in (Hq, Hkv)
    P = bp.shape[2]
    bt = np.asarray(block_table)
    quant = antiquant_scale is not None
    per_token = quant and antiquant_mode in (1, 3)
    assert not per_token or exact
    if per_token:
        # [2, B, S], or with a row per kv head (mode 3) [2, B, Hkv, S]: both as [2, B, heads, S]
        heads = Hkv if antiquant_mode == 3 else 1
        pt_s = np.asarray(antiquant_scale, dtype=np.float64).reshape(2, B, heads, -1)
        pt_o = None if antiquant_offset is None else np.asarray(antiquant_offset, dtype=np.float64).reshape(2, B, heads, -1)
        a_o = None
    elif quant:
        def _params(t):
            t = np.asarray(t, dtype=np.float32)
            if antiquant_mode == 2:     # one value per kv head
                return np.broadcast_to(t.reshape(2, Hkv, 1), (2, Hkv, D))
            if t.size == 2:
                return np.broadcast_to(t.reshape(2, 1, 1), (2, Hkv, D))
            return t.reshape(2, Hkv, D)
        a_s = _params(antiquant_scale)
        a_o = _params(antiquant_offset) if antiquant_offset is not None else None
    pse = None if pse_shift is None else np.asarray(pse_shift, dtype=np.float32)
    amask = None if atten_mask is None else np.asarray(atten_mask).reshape(B, -1)
    assert (pse is None and amask is None) or exact, "pse_shift / atten_mask: exact mode only"
    out = np.zeros((B, Hq, D), dtype=np.float32)
    lse = np.full((B, Hq), np.float32(-3.40e38), dtype=np.float32)
    assert not return_lse or exact, "return_lse: exact mode only"
    facc = np.float64 if exact else np.float32
    n_pre = 0
    if key_shared_prefix is not None:
        kp, vp = np.asarray(key_shared_prefix), np.asarray(value_shared_prefix)
        if kv_layout == "BSH":
            kp, vp = kp.reshape(-1, Hkv, D), vp.reshape(-1, Hkv, D)                  # [S, Hkv, D]
        else:
            kp, vp = kp.reshape(Hkv, -1, D).transpose(1, 0, 2), vp.reshape(Hkv, -1, D).transpose(1, 0, 2)
        n_pre = kp.shape[0] if shared_prefix_len is None else int(shared_prefix_len)
        assert 0 <= n_pre <= kp.shape[0] and not token_level
    for b in range(B):
        for h in range(Hq):
            n2 = h // G
            row = bp[b, h if rows == Hq else n2]
            sent = np.nonzero(row == SENTINEL)[0]
            n_pages = int(sent[0]) if sent.size else P
            if actual_seq_lengths is not None:
                n_tok = int(actual_seq_lengths[b])
                n_pages = min(n_pages, (n_tok + block_size - 1) // block_size)
            else:
                n_tok = n_pages * block_size
            qv = q[b, h].astype(facc)
            if quant and not per_token:
                qv = qv * a_s[0, n2].astype(facc)
            ks, vs, cols = [], [], []
            if n_pre:
                ks.append(kp[:n_pre, n2])
                vs.append(vp[:n_pre, n2])
                cols.append(np.arange(n_pre))
            if token_level:
                n_tok = P if actual_seq_lengths is None else int(actual_seq_lengths[b])
                n_ent = min(int(sent[0]) if sent.size else P, max(0, n_tok))
                toks = row[:n_ent].astype(np.int64)
                bad = np.nonzero((toks < 0) | (toks >= bt.shape[1] * block_size))[0]
                if bad.size:
                    toks = toks[:int(bad[0])]
                if toks.size:
                    blk = np.clip(bt[b, toks // block_size], 0, key_cache.shape[0] - 1)
                    kc = key_cache.reshape(key_cache.shape[0], -1, Hkv, D) if kv_layout == "BSH" else \
                        key_cache.reshape(key_cache.shape[0], Hkv, -1, D).transpose(0, 2, 1, 3)
                    vc = value_cache.reshape(value_cache.shape[0], -1, Hkv, D) if kv_layout == "BSH" else \
                        value_cache.reshape(value_cache.shape[0], Hkv, -1, D).transpose(0, 2, 1, 3)
                    ks.append(kc[blk, toks % block_size, n2])
                    vs.append(vc[blk, toks % block_size, n2])
                    cols.append(np.arange(toks.size))
                n_pages = 0
            for j in range(n_pages):
                p = int(row[j])
                blk = int(bt[b, p])
                valid = min(block_size, n_tok - j * block_size)
                if token_limit is not None:
                    valid = min(valid, int(token_limit[b]) - p * block_size)
                if valid <= 0:
                    continue
                ks.append(_kv_rows(key_cache, blk, n2, kv_layout, Hkv, D)[:valid])
                vs.append(_kv_rows(value_cache, blk, n2, kv_layout, Hkv, D)[:valid])
                cols.append(n_pre + j * block_size + np.arange(valid))
            if not ks:
                continue
            if exact:
                K = np.concatenate(ks).astype(np.float64)
                V = np.concatenate(vs).astype(np.float64)
                col = np.concatenate(cols)
                if per_token:
                    ph = n2 if antiquant_mode == 3 else 0
                    if pt_o is not None:
                        K = K + pt_o[0, b, ph, col][:, None]
                        V = V + pt_o[1, b, ph, col][:, None]
                    K = K * pt_s[0, b, ph, col][:, None]
                    V = V * pt_s[1, b, ph, col][:, None]
                s = (K @ qv) * np.float64(scale)
                if pse is not None:
                    s = s + pse[b if pse.shape[0] > 1 else 0, h, 0, col].astype(np.float64)
                if amask is not None:
                    s = np.where(amask[b, col] != 0, np.float64(-1e12), s)
                p_ = np.exp(s - s.max())
                o = (p_ @ V) / p_.sum()
                row_const = 0.0
                if quant and a_o is not None:   # the key offset's share of every score of the row
                    row_const = float(scale) * float((qv * a_o[0, n2].astype(np.float64)).sum())
                lse[b, h] = np.float32(np.log(p_.sum()) + s.max() + row_const)
            else:
                m = np.float32(-2e38)
                l = np.float32(0)
                o = np.zeros((D,), np.float32)
                for Kp, Vp in zip(ks, vs):
                    s = (Kp.astype(np.float32) @ qv).astype(np.float32) * np.float32(scale)
                    m_new = np.maximum(m, s.max())
                    alpha = np.exp(m - m_new, dtype=np.float32)
                    p_ = np.exp(s - m_new, dtype=np.float32)
                    l = l * alpha + p_.sum(dtype=np.float32)
                    if p_round == "f16":
                        p_ = round_to_f16(p_)
                    elif p_round == "bf16":
                        p_ = round_to_bf16(p_)
                    o = o * alpha + (p_ @ Vp.astype(np.float32)).astype(np.float32)
                    m = m_new
                o = o / l
            if quant and not per_token:
                if a_o is not None:
                    o = o + a_o[1, n2].astype(facc)
                o = o * a_s[1, n2].astype(facc)
            out[b, h] = o.astype(np.float32)
    if quant_scale2 is not None:
        out = post_quant(out, quant_scale2, quant_offset2)
    return (out, lse) if return_lse else out


def _round_half_away(x: np.ndarray) -> np.ndarray:
    return np.sign(x) * np.floor(np.abs(x) + 0.5)


def f32_to_f16_half_away(x: np.ndarray) -> np.ndarray:
    """float32 -> float16 rounding to nearest with ties away from zero (the accelerator's CAST_ROUND)."""
    x = np.asarray(x, dtype=np.float32)
    ax = np.minimum(np.abs(x), np.float32(65504))
    near = ax.astype(np.float16)                         # ties to even
    below = np.where(near.astype(np.float32) > ax, np.nextafter(near, np.float16(0)), near)
    with np.errstate(over="ignore"):
        above = np.nextafter(below, np.float16(np.inf))
    lo, hi = below.astype(np.float64), above.astype(np.float64)
    up = np.isfinite(hi) & ((ax.astype(np.float64) - lo) * 2 >= (hi - lo))
    return np.where(up, above, below).astype(np.float16) * np.sign(x).astype(np.float16)


def post_quant(out: np.ndarray, quant_scale2, quant_offset2=None) -> np.ndarray:
    """int8 output quantisation of attention_out [B, Hq, D] (…Us2.h:4016-4111); float32 holding int8 values."""
    B, Hq, D = out.shape
    sc = np.asarray(quant_scale2, dtype=np.float32).reshape(-1)
    sc = sc.reshape(1, Hq, D) if sc.size > 1 else sc.reshape(1, 1, 1)
    x = out.astype(np.float32) * sc
    if quant_offset2 is not None:
        of = np.asarray(quant_offset2, dtype=np.float32).reshape(-1)
        x = x + (of.reshape(1, Hq, D) if of.size > 1 else of.reshape(1, 1, 1))
    h = f32_to_f16_half_away(x.astype(np.float32)).astype(np.float32)
    return np.clip(_round_half_away(h), -128, 127).astype(np.float32)


def sparse_paged_fusion_attention(q, key_cache, value_cache, block_table, l1_cent, block_ids,
                                  total_seq_len, num_heads: int, num_kv_heads: int, scale: float,
                                  block_size: int = 128, k: int = 64, max_page_num: int = 256,
                                  sink_recent: bool = False, max_pages_cap: bool = False,
                                  kv_layout: str = "BSH", antiquant_scale=None,
                                  antiquant_offset=None, p_round=None, exact: bool = True,
                                  pse_shift=None, atten_mask=None, quant_scale2=None, quant_offset2=None,
                                  key_shared_prefix=None, value_shared_prefix=None, shared_prefix_len=None,
                                  return_lse: bool = False, antiquant_mode: int = 0):
    """cent_select then attention over each q head's own page list
    (sparse_paged_fusion_attention.cpp:56-76; attention runs with
    actualSeqLengths := max_page_position_length and blockPosition := the rows just
    written, .cpp:68).  A head attends over page_position_length[b,h] pages; tokens of a
    selected page at logical index >= total_seq_len[b] are masked (identical to the
    reference whenever total_seq_len is a multiple of block_size, as in all its tests).

    Returns (attention_out fp32 [B,Hq,D], page_position, page_position_length[B,Hq,8],
             max_page_position_length [B,8]); with return_lse=True attention_out is the pair (out, lse).
    """
    pos, len8, max8, _ = cent_select(q, l1_cent, block_ids, block_table, total_seq_len, k,
                                     block_size, max_page_num, sink_recent, max_pages_cap)
    B = np.asarray(total_seq_len).shape[0]
    # honour the (possibly capped) length: slots past it behave as sentinels
    eff = pos.copy()
    cols = np.arange(eff.shape[2])[None, None, :]
    eff[cols >= len8[:, :, :1]] = SENTINEL
    out = sparse_paged_attention(np.asarray(q)[:B], key_cache, value_cache, block_table, eff,
                                 None, num_heads, num_kv_heads, scale, block_size, kv_layout,
                                 antiquant_scale, antiquant_offset,
                                 token_limit=np.asarray(total_seq_len), p_round=p_round, exact=exact,
                                 pse_shift=pse_shift, atten_mask=atten_mask, quant_scale2=quant_scale2,
                                 quant_offset2=quant_offset2, key_shared_prefix=key_shared_prefix,
                                 value_shared_prefix=value_shared_prefix, shared_prefix_len=shared_prefix_len,
                                 return_lse=return_lse, antiquant_mode=antiquant_mode)
    return out, pos, len8, max8


# --------------------------------------------------------------------------- #
# assign_block_ids: cluster ids of newly filled pages.  The reference has NO producer of
# block_ids (README.md:33 only describes the tensor), so there is nothing to pin this
# against: "parity unpinned".  The definition below is the one the HIP kernel implements.
# --------------------------------------------------------------------------- #
def page_means(key_cache, block_table, seq_len, first_new_page, num_kv_heads: int, head_dim: int,
               block_size: int = 128, kv_layout: str = "BSH", max_new_pages: int = 0,
               antiquant_scale=None, antiquant_offset=None, dt: str = "f16"):
    """[(b, page, phys, mean[Hkv, D] rounded to dt)] for every new page, in (b, page) order.

    mean = fp32 sum over the page's live tokens times fp32 (1 / tokens); int8 K:
    (mean + offset_k) * scale_k per channel (key parameters = index 0, …Us2.h:2035-2043)."""
    bt = np.asarray(block_table)
    seq_len = np.asarray(seq_len)
    first = np.asarray(first_new_page)
    Hkv, D = num_kv_heads, head_dim
    max_page = bt.shape[1]
    max_new = max_page if max_new_pages <= 0 else min(int(max_new_pages), max_page)
    n_blocks = key_cache.shape[0]
    rnd = round_to_bf16 if dt == "bf16" else round_to_f16
    quant = antiquant_scale is not None
    if quant:
        def _params(t):
            t = np.asarray(t, dtype=np.float32)
            return np.broadcast_to(t.reshape(2, 1, 1), (2, Hkv, D)) if t.size == 2 else t.reshape(2, Hkv, D)
        a_s = _params(antiquant_scale)
        a_o = _params(antiquant_offset) if antiquant_offset is not None else None
    out = []
    for b in range(seq_len.shape[0]):
        page_len = max(0, min((int(seq_len[b]) + block_size - 1) // block_size, max_page))
        f0 = max(0, int(first[b]))
        for page in range(f0, min(page_len, f0 + max_new)):
            phys = int(bt[b, page])
            if phys < 0 or phys >= n_blocks:     # an unassigned slot names no page: skipped
                continue
            valid = min(block_size, int(seq_len[b]) - page * block_size)
            m = np.zeros((Hkv, D), np.float32)
            for n2 in range(Hkv):
                rows = _kv_rows(key_cache, phys, n2, kv_layout, Hkv, D)[:valid].astype(np.float64)
                ssum = rows.sum(axis=0).astype(np.float32)
                m[n2] = ssum * (np.float32(1.0) / np.float32(valid))
                if quant:
                    if a_o is not None:
                        m[n2] = m[n2] + a_o[0, n2]
                    m[n2] = m[n2] * a_s[0, n2]
            out.append((b, page, phys, rnd(m)))
    return out


def assign_block_ids(key_cache, block_table, seq_len, first_new_page, l1_cent, block_ids,
                     block_size: int = 128, kv_layout: str = "BSH", max_new_pages: int = 0,
                     antiquant_scale=None, antiquant_offset=None, dt: str = "f16"):
    """block_ids[n, phys] = argmax_c mean'(page, n) . l1_cent[n, c, :] for every new page (ties: smallest c).
    Returns (new block_ids, scores) where scores[(b, page)] = float64 [Hkv, C] of that page."""
    l1 = np.asarray(l1_cent, dtype=np.float64)
    Hkv, C, D = l1.shape
    ids = np.array(block_ids, copy=True)
    scores = {}
    for b, page, phys, m in page_means(key_cache, block_table, seq_len, first_new_page, Hkv, D, block_size,
                                       kv_layout, max_new_pages, antiquant_scale, antiquant_offset, dt):
        sc = np.einsum("nd,ncd->nc", m.astype(np.float64), l1)
        scores[(b, page)] = sc
        if phys < ids.shape[1]:
            ids[:, phys] = np.argmax(sc, axis=1).astype(np.int32)     # first maximum = smallest id
    return ids, scores
