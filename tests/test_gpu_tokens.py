"""Token-level positions on sparse_paged_attention (token_level=True): the list of a head holds logical TOKEN indices
instead of page indices.  The reference only sketches this variant in a test (tests/ifa_v5_case_tkp.py: token_position
[B, Hkv, max_tokens] :53-54, the sparse pattern :96-135, the identity mapping :370-379) against an op its package no
longer exports, so the pins are its two known-answer constructions; numerics beyond them are checked against the
oracle's restatement (parity unpinned).  Tolerance: 1e-3 relative, elements below 1 % of the largest output measured
against that floor."""
import numpy as np
import pytest
import torch

import datagen
from oracle import ref
from parity import rel_err

pytestmark = pytest.mark.gpu
SENT = 0x7FFFFFFF


@pytest.fixture(scope="module")
def H():
    import gpu_helpers
    return gpu_helpers


def _pool(seed, n_blocks, bs, Hkv, D, layout, kvdt):
    if kvdt == "i8":
        k, v = datagen.kv_case(seed, n_blocks, bs, Hkv, D, layout, dtype="i8")
        sc, off = datagen.antiquant_case(seed, Hkv, D)
        return k, v, sc, off
    k, v = datagen.kv_case(seed, n_blocks, bs, Hkv, D, layout, dist="randn")
    if kvdt == "bf16":
        k, v = ref.round_to_bf16(k.astype(np.float32)), ref.round_to_bf16(v.astype(np.float32))
    return k, v, None, None


def _to_dev(a, kvdt, H):
    if a is None:
        return None
    if a.dtype == np.int8 or a.dtype == np.int32:
        return H.dev(a)
    if kvdt == "bf16":
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(H.DEV).to(torch.bfloat16)
    return H.dev(a.astype(np.float16))


def _random_lists(rng, B, rows, max_tokens, seq_tokens, counts):
    """Sorted random token subsets per (b, row), sentinel padded."""
    tp = np.full((B, rows, max_tokens), SENT, dtype=np.int32)
    for b in range(B):
        for r in range(rows):
            n = counts[b][r]
            tp[b, r, :n] = np.sort(rng.choice(seq_tokens[b], size=n, replace=False))
    return tp


@pytest.mark.parametrize("B,Hq,Hkv,D,bs,kvdt,layout,rows_hq,max_tokens", [
    (3, 8, 8, 128, 128, "f16", "BSH", True, 700),
    (3, 8, 2, 128, 128, "f16", "BSH", False, 700),       # lists shared by a group
    (2, 8, 2, 128, 128, "f16", "BNSD", True, 300),       # own list per q head of a group
    (3, 4, 4, 64, 16, "f16", "BSH", True, 260),
    (3, 8, 8, 128, 128, "bf16", "BSH", True, 500),
    (3, 8, 4, 128, 128, "i8", "BSH", False, 640),
    (2, 4, 4, 64, 32, "i8", "BNSD", True, 200),
    (40, 32, 32, 128, 128, "f16", "BSH", True, 384),      # 1280 items: more than the resident workgroups
    # lists shared by a group are streamed once for all its q heads (the matrix-core streamer, token-addressed)
    (3, 32, 4, 128, 128, "f16", "BSH", False, 700),      # groups of 8
    (3, 16, 4, 128, 128, "bf16", "BNSD", False, 500),
    (3, 8, 2, 64, 16, "f16", "BSH", False, 260),
    (2, 16, 4, 64, 32, "i8", "BNSD", False, 333),
    (2, 64, 2, 128, 128, "f16", "BSH", False, 450),      # groups of 32: two passes of 16 heads
    (1, 32, 8, 128, 128, "f16", "BSH", False, 1400),     # one sequence: the list is dealt to several workgroups
    (80, 32, 8, 128, 128, "i8", "BSH", False, 300),      # 640 items: more than the resident workgroups
])
def test_token_lists_vs_oracle(H, B, Hq, Hkv, D, bs, kvdt, layout, rows_hq, max_tokens):
    _run_case(H, B, Hq, Hkv, D, bs, kvdt, layout, rows_hq, max_tokens, 12)


@pytest.mark.parametrize("B,Hq,Hkv,D,bs,kvdt,layout,rows_hq,max_tokens,max_page", [
    (2, 8, 8, 128, 128, "f16", "BSH", True, 5000, 48),       # three windows, one q head per pass
    (2, 8, 2, 128, 128, "f16", "BSH", False, 10000, 100),    # group form, five windows
    (2, 16, 4, 128, 128, "i8", "BSH", False, 9000, 80),
    (1, 8, 8, 64, 16, "bf16", "BNSD", True, 2500, 200),      # one sequence: windows of a list dealt to several workgroups
    (1, 32, 8, 128, 128, "f16", "BSH", False, 6000, 50),
    (24, 32, 32, 128, 128, "f16", "BSH", True, 2300, 24),    # 768 items, two windows each, through the queue
])
@pytest.mark.parametrize("window", [None, 512])
def test_long_token_lists_are_streamed_in_windows(H, monkeypatch, B, Hq, Hkv, D, bs, kvdt, layout, rows_hq, max_tokens,
                                                  max_page, window):
    """The workgroup holds at most 4096 entries of a list at a time (fewer where a wide pass leaves less room; 512 when
    the test asks for it); longer lists go through as windows whose states are merged.  Rows of exactly one and two
    windows, a row that ends one entry into a window, and random lengths."""
    if window:
        monkeypatch.setenv("SPFA_TOKEN_WINDOW", str(window))
    w = window or 4096
    _run_case(H, B, Hq, Hkv, D, bs, kvdt, layout, rows_hq, max_tokens, max_page, exact=(w, 2 * w, w + 1))


def _run_case(H, B, Hq, Hkv, D, bs, kvdt, layout, rows_hq, max_tokens, max_page, exact=()):
    rng = datagen.rng_for(7 * B + Hq + max_tokens)
    n_blocks = min(B * max_page, 160)
    k, v, sc, off = _pool(11, n_blocks, bs, Hkv, D, layout, kvdt)
    table = rng.integers(0, n_blocks, size=(B, max_page)).astype(np.int32)
    seq_tokens = [int(x) for x in rng.integers(max_tokens // 3, max_page * bs, size=B)]
    rows = Hq if rows_hq else Hkv
    counts = [[int(rng.integers(1, min(max_tokens, seq_tokens[b]) + 1)) for _ in range(rows)] for b in range(B)]
    counts[0][0] = min(max_tokens, seq_tokens[0])                         # a full row (no sentinel)
    counts[B - 1][rows - 1] = 1                                           # a single token
    for i, n in enumerate(exact):
        b, r = (i + 1) % B, (i + 1) % rows
        if n <= min(max_tokens, seq_tokens[b]) and (b, r) not in ((0, 0), (B - 1, rows - 1)):
            counts[b][r] = n
    tp = _random_lists(rng, B, rows, max_tokens, seq_tokens, counts)
    q = datagen.ref_rand_f16(rng, (B, Hq, D)).astype(np.float32)
    if kvdt == "bf16":
        q = ref.round_to_bf16(q)
    qd, kd, vd = _to_dev(q, kvdt if kvdt != "i8" else "f16", H), _to_dev(k, kvdt, H), _to_dev(v, kvdt, H)
    kw = dict(antiq_scale=_to_dev(sc, "f16", H), antiq_offset=_to_dev(off, "f16", H)) if kvdt == "i8" else {}
    import custom_ops
    e = H.empty(qd.dtype)
    got = custom_ops.sparse_paged_attention(qd, kd, vd, e, e, None, e, e, e, e, e, kw.get("antiq_scale", e),
                                            kw.get("antiq_offset", e), H.dev(table), H.empty(torch.int64), H.dev(tp), Hq,
                                            D ** -0.5, "BNSD", Hkv, bs, 1, out_dtype=torch.float32, token_level=True)
    want = ref.sparse_paged_attention(q, k, v, table, tp, None, Hq, Hkv, D ** -0.5, bs, layout, sc, off, token_level=True)
    assert rel_err(H.f32(got), want).max() <= 1e-3
    # actual_seq_lengths bound the lists (host list, then a device tensor)
    lens = [int(x) for x in rng.integers(1, max_tokens + 1, size=B)]
    want2 = ref.sparse_paged_attention(q, k, v, table, tp, lens, Hq, Hkv, D ** -0.5, bs, layout, sc, off, token_level=True)
    for arg in (lens, torch.tensor(lens, dtype=torch.int64, device=H.DEV)):
        got2 = custom_ops.sparse_paged_attention(qd, kd, vd, e, e, arg, e, e, e, e, e, kw.get("antiq_scale", e),
                                                 kw.get("antiq_offset", e), H.dev(table), H.empty(torch.int64), H.dev(tp),
                                                 Hq, D ** -0.5, "BNSD", Hkv, bs, 1, out_dtype=torch.float32,
                                                 token_level=True)
        assert rel_err(H.f32(got2), want2).max() <= 1e-3


def test_identity_token_list_equals_page_level_attention(H):
    """tests/ifa_v5_case_tkp.py:370-379: token_position = 0 .. n-1 over an identity block table is ordinary paged
    attention; it must agree with the page-level op given block_position = 0 .. pages-1 and the same lengths."""
    B, Hq, Hkv, D, bs, pages = 3, 8, 2, 128, 128, 9
    rng = datagen.rng_for(3)
    k, v, _, _ = _pool(3, B * pages, bs, Hkv, D, "BSH", "f16")
    table = np.arange(B * pages, dtype=np.int32).reshape(B, pages)
    lens = [pages * bs, 5 * bs + 17, 1]
    tp = np.tile(np.arange(pages * bs, dtype=np.int32), (B, Hkv, 1))
    bp = np.tile(np.arange(pages, dtype=np.int32), (B, Hkv, 1))
    q = datagen.ref_rand_f16(rng, (B, Hq, D))
    args = (H.dev(q), H.dev(k), H.dev(v), H.dev(table))
    import custom_ops
    e = H.empty()
    tok = custom_ops.sparse_paged_attention(args[0], args[1], args[2], e, e, lens, e, e, e, e, e, e, e, args[3],
                                            H.empty(torch.int64), H.dev(tp), Hq, D ** -0.5, "BNSD", Hkv, bs, 1,
                                            out_dtype=torch.float32, token_level=True)
    page = H.call_attention(*args, H.dev(bp), lens, Hq, Hkv, D ** -0.5, bs, out_dtype=torch.float32)
    assert float((tok - page).abs().max()) <= 1e-5 * float(page.abs().max())
    want = ref.sparse_paged_attention(q, k, v, table, tp, lens, Hq, Hkv, D ** -0.5, bs, "BSH", token_level=True)
    assert rel_err(H.f32(tok), want).max() <= 1e-3


def test_reference_sparse_token_pattern_known_answer(H):
    """The construction of tests/ifa_v5_case_tkp.py:53-59,96-135,200-225: per kv head a list with a contiguous front, a
    stride-4 middle and a stride-7 back; K = V = 0.1 + 0.001 h on the listed tokens and 9.9 everywhere else, q = 0.1.
    V is constant over the attended set, so the exact answer is that constant whatever the softmax weights."""
    B, Hq, Hkv, D, bs = 2, 8, 4, 128, 16
    total = 256 * bs
    per_head = 1024
    lists = []
    for h in range(Hkv):
        base = (h * 17) % total
        front, middle = min(512, per_head // 3), min(1024, per_head // 3)
        toks = [(base + i) % total for i in range(front)]
        cur = base + front
        toks += [(cur + 4 * i) % total for i in range(middle)]
        cur += 4 * middle
        toks += [(cur + 7 * i + 13 * h) % total for i in range(per_head - front - middle)]
        lists.append(sorted(set(toks))[:per_head])
    n_blocks = total // bs
    k = np.full((n_blocks, bs, Hkv, D), 9.9, dtype=np.float16)
    v = np.full((n_blocks, bs, Hkv, D), 9.9, dtype=np.float16)
    tp = np.full((B, Hkv, per_head), SENT, dtype=np.int32)
    for h, toks in enumerate(lists):
        val = np.float16(0.1 + 0.001 * h)
        t = np.asarray(toks)
        k[t // bs, t % bs, h, :] = val
        v[t // bs, t % bs, h, :] = val
        tp[:, h, :len(toks)] = t
    table = np.tile(np.arange(n_blocks, dtype=np.int32), (B, 1))            # identity mapping, as in the reference
    q = np.full((B, Hq, D), 0.1, dtype=np.float16)
    import custom_ops
    e = H.empty()
    out = custom_ops.sparse_paged_attention(H.dev(q), H.dev(k), H.dev(v), e, e, [per_head] * B, e, e, e, e, e, e, e,
                                            H.dev(table), H.empty(torch.int64), H.dev(tp), Hq, D ** -0.5, "BNSD", Hkv, bs,
                                            0, out_dtype=torch.float32, token_level=True)
    got = H.f32(out)
    G = Hq // Hkv
    for h in range(Hq):
        want = float(np.float16(0.1 + 0.001 * (h // G)))
        assert np.abs(got[:, h, :] - want).max() <= 1e-3 * want, (h, got[:, h, :4])
    assert np.abs(got - 9.9).min() > 1.0                                    # the reference's own pass criterion (:177-190)


def test_token_level_rejects_what_it_does_not_support(H):
    import custom_ops
    from custom_ops import _abi
    B, Hq, D, bs = 2, 4, 256, 128
    q = torch.zeros((B, Hq, D), dtype=torch.float16, device=H.DEV)
    kv = torch.zeros((8, bs, Hq, D), dtype=torch.float16, device=H.DEV)
    table = torch.zeros((B, 4), dtype=torch.int32, device=H.DEV)
    tp = torch.zeros((B, Hq, 64), dtype=torch.int32, device=H.DEV)
    e = H.empty()
    with pytest.raises(_abi.AclError, match="head dims 64 and 128"):
        custom_ops.sparse_paged_attention(q, kv, kv, e, e, None, e, e, e, e, e, e, e, table, H.empty(torch.int64), tp, Hq,
                                          0.1, "BNSD", Hq, bs, 1, token_level=True)
    # the setting is one-shot: the same call without the keyword treats the tensor as page indices again
    out = custom_ops.sparse_paged_attention(q, kv, kv, e, e, None, e, e, e, e, e, e, e, table, H.empty(torch.int64), tp, Hq,
                                            0.1, "BNSD", Hq, bs, 1)
    assert out.shape == q.shape
