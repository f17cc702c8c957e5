"""GPU parity (through the C-ABI) of sparse_paged_attention and sparse_paged_fusion_attention.

The reference has no CPU attention; the oracle restates the kernel arithmetic (oracle/ref.py) and is
pinned here by the reference's constant-V known-answer construction (tests/ifa_v5_case.py:584-765),
the dense-equivalence identity and the fused == separate identity
(src/transformer/sparse_paged_fusion_attention/test/fused_ifa_v5_case.py:215-276).
Tolerance: 1e-3 relative on attention_out (BASELINE north_star), measured on the float32 output of
the kernel; the fp16 / bf16 outputs must equal that float32 result rounded once.
"""
import numpy as np
import pytest
import torch

import datagen
import parity
from oracle import ref

pytestmark = pytest.mark.gpu

SENT = 0x7FFFFFFF
TOL = 1e-3


@pytest.fixture(scope="module")
def H():
    import gpu_helpers
    return gpu_helpers


def _kat(H, layout, bs, Hq, Hkv, D=128, pages_per_head=6, n_blocks=64, head_delta=0.001, B=2):
    """Selected pages hold head-specific constants, everything else 9.9: the exact answer is the
    head's constant whatever the softmax weights are (tests/ifa_v5_case.py:634-691)."""
    shape = (n_blocks, bs, Hkv, D) if layout == "BSH" else (n_blocks, Hkv, bs, D)
    k = np.full(shape, 9.9, np.float16)
    v = np.full(shape, 9.9, np.float16)
    table = np.full((B, Hkv * pages_per_head), SENT, np.int32)
    pos = np.full((B, Hkv, pages_per_head + 3), SENT, np.int32)
    vals = []
    slot = 0
    for h in range(Hkv):
        blocks = [((h * 3) % n_blocks + 2 * i) % n_blocks for i in range(pages_per_head)]
        val = np.float16(0.1 + h * head_delta)
        vals.append(float(val))
        for i, blk in enumerate(blocks):
            if layout == "BSH":
                k[blk, :, h, :] = val
                v[blk, :, h, :] = val
            else:
                k[blk, h] = val
                v[blk, h] = val
            table[:, slot] = blk
            pos[:, h, i] = slot
            slot += 1
    q = np.full((B, Hq, D), 0.1, np.float16)
    out = H.call_attention(H.dev(q), H.dev(k), H.dev(v), H.dev(table), H.dev(pos),
                           torch.tensor([pages_per_head * bs], dtype=torch.int64), Hq, Hkv, 1.0, bs)
    out = H.f32(out)
    G = Hq // Hkv
    for h in range(Hq):
        assert np.allclose(out[:, h], vals[h // G], rtol=1e-3, atol=0), (h, out[0, h, :4], vals[h // G])
    assert not (np.abs(out - 9.9) < 1.0).any()          # tests/ifa_v5_case.py:751


@pytest.mark.parametrize("layout,bs,Hq,Hkv", [("BSH", 128, 8, 8), ("BNSD", 128, 8, 8), ("BSH", 16, 8, 8),
                                              ("BNSD", 16, 4, 4), ("BSH", 128, 8, 2), ("BSH", 64, 16, 4)])
def test_known_answer_constant_v(H, layout, bs, Hq, Hkv):
    _kat(H, layout, bs, Hq, Hkv)


def _random_lists(rng, B, rows, max_page, P, lo, hi):
    pos = np.full((B, rows, P), SENT, np.int32)
    for b in range(B):
        for r in range(rows):
            n = int(rng.integers(lo, hi + 1))
            pos[b, r, :n] = np.sort(rng.choice(max_page, size=n, replace=False))
    return pos


@pytest.mark.parametrize("dtype,layout,Hq,Hkv,rows,D,bs,seq_mode", [
    ("f16", "BSH", 8, 8, "kv", 128, 128, "none"),
    ("f16", "BSH", 8, 2, "kv", 128, 128, "uniform"),
    ("f16", "BSH", 8, 2, "q", 128, 128, "ragged"),
    ("f16", "BNSD", 4, 4, "q", 128, 16, "ragged"),
    ("bf16", "BSH", 8, 2, "kv", 128, 128, "ragged"),
    ("bf16", "BNSD", 16, 2, "kv", 64, 32, "uniform"),
    ("f16", "BSH", 4, 4, "kv", 64, 128, "device"),
    ("f16", "BSH", 4, 1, "kv", 256, 64, "none"),
    ("f16", "BSH", 32, 2, "kv", 128, 128, "ragged"),     # G = 16: one pass of 16 heads
    ("f16", "BSH", 6, 2, "kv", 128, 48, "ragged"),       # G = 3, pages of one and a half 32-token tiles
    ("bf16", "BSH", 32, 1, "kv", 128, 128, "device"),    # G = 32: two passes of 16 heads
    ("f16", "BNSD", 4, 2, "kv", 32, 16, "ragged"),       # G = 2, D = 32
    ("bf16", "BSH", 10, 2, "kv", 256, 32, "none"),       # G = 5, D = 256
    ("f16", "BSH", 4, 2, "kv", 512, 32, "ragged"),       # D = 512: 1 KiB head rows, one q head per pass
    ("bf16", "BNSD", 2, 2, "q", 512, 16, "device"),
])
def test_attention_vs_oracle(H, dtype, layout, Hq, Hkv, rows, D, bs, seq_mode):
    rng = datagen.rng_for(1234 + Hq * 7 + D)
    B, n_blocks, max_page, P = 3, 40, 24, 12
    tdt = torch.float16 if dtype == "f16" else torch.bfloat16
    q = datagen.ref_rand_f16(rng, (B, Hq, D))
    k, v = datagen.kv_case(5, n_blocks, bs, Hkv, D, layout, dist="randn")
    if dtype == "bf16":   # make every value exactly representable in bf16
        q, k, v = (ref.round_to_bf16(x.astype(np.float32)) for x in (q, k, v))
    table = rng.integers(0, n_blocks, size=(B + 2, max_page)).astype(np.int32)
    R = Hq if rows == "q" else Hkv
    pos = _random_lists(rng, B, R, max_page, P, 1, P - 2)
    pos[0, 0, :] = SENT                                     # a head with an empty list -> zeros
    act_host, act_arg = None, torch.empty(0, dtype=torch.int64)
    if seq_mode == "uniform":
        act_host = [5 * bs - 3] * B
        act_arg = [5 * bs - 3]
    elif seq_mode in ("ragged", "device"):
        act_host = [int(x) for x in rng.integers(1, P * bs, size=B)]
        act_arg = torch.tensor(act_host, dtype=torch.int64)
        if seq_mode == "device":
            act_arg = act_arg.to(H.DEV)
    scale = 1.0 / np.sqrt(D)
    want = ref.sparse_paged_attention(q, k, v, table, pos, act_host, Hq, Hkv, scale, bs, layout)
    qd, kd, vd = H.dev(q, tdt), H.dev(k, tdt), H.dev(v, tdt)
    out32 = H.f32(H.call_attention(qd, kd, vd, H.dev(table), H.dev(pos), act_arg, Hq, Hkv, scale, bs,
                                   out_dtype=torch.float32))
    err = float(parity.rel_err(out32, want).max())
    assert err <= TOL, f"max rel err {err:.3e}"
    G = Hq // Hkv
    assert (out32[0, :(G if rows == "kv" else 1)] == 0).all()           # empty list -> zeros
    out = H.call_attention(qd, kd, vd, H.dev(table), H.dev(pos), act_arg, Hq, Hkv, scale, bs)
    assert out.dtype == tdt and out.shape == qd.shape
    assert torch.equal(out.cpu(), torch.from_numpy(out32).to(tdt))      # one rounding of the fp32 result
    # reference-numerics mode of the oracle (online softmax, P rounded to the KV dtype) agrees too
    want2 = ref.sparse_paged_attention(q, k, v, table, pos, act_host, Hq, Hkv, scale, bs, layout,
                                       exact=False, p_round=dtype)
    # (P rounding costs up to 2^-11 / 2^-8 per term; with zero-mean V compare against the row scale)
    tol2 = 2e-3 if dtype == "f16" else 1.5e-2
    dev2 = float(np.abs(want2 - want).max() / np.abs(want).max())
    assert dev2 <= tol2, dev2
    # ... and so does the kernel against it: the reference rounds P to the KV dtype before P.V (…Us2.h:3567-3568), the
    # kernel keeps P in fp32 (or a high + low pair), so the two differ by that rounding, measured against the largest
    # output of the tensor (V is zero-mean here, element-wise relative errors are meaningless near the zeros)
    dev3 = float(np.abs(out32 - want2).max() / np.abs(want2).max())
    assert dev3 <= tol2, dev3


def test_dense_equivalence_vs_torch_sdpa(H):
    """block_position[b,h,j] = j turns the op into ordinary paged attention
    (tests/ifa_v5_case_tkp.py:370-379 uses the identity mapping)."""
    rng = datagen.rng_for(99)
    B, Hq, Hkv, D, bs, n_pages = 2, 8, 2, 128, 128, 5
    q = datagen.ref_rand_f16(rng, (B, Hq, D))
    k, v = datagen.kv_case(9, B * n_pages, bs, Hkv, D, "BSH", dist="randn")
    table = np.arange(B * n_pages, dtype=np.int32).reshape(B, n_pages)
    pos = np.tile(np.arange(n_pages, dtype=np.int32), (B, Hkv, 1))
    seq = n_pages * bs - 37
    out = H.f32(H.call_attention(H.dev(q), H.dev(k), H.dev(v), H.dev(table), H.dev(pos), [seq], Hq, Hkv,
                                 D ** -0.5, bs, out_dtype=torch.float32))
    kk = torch.from_numpy(k.astype(np.float32)).reshape(B, n_pages * bs, Hkv, D)[:, :seq]
    vv = torch.from_numpy(v.astype(np.float32)).reshape(B, n_pages * bs, Hkv, D)[:, :seq]
    kk = kk.permute(0, 2, 1, 3).repeat_interleave(Hq // Hkv, dim=1)
    vv = vv.permute(0, 2, 1, 3).repeat_interleave(Hq // Hkv, dim=1)
    qq = torch.from_numpy(q.astype(np.float32)).unsqueeze(2)
    want = torch.nn.functional.scaled_dot_product_attention(qq, kk, vv).squeeze(2).numpy()
    assert float(parity.rel_err(out, want).max()) <= TOL


def _fusion_case(seed, B, Hq, Hkv, D, C, seq_lens, kv_page_len, max_page, bs, kv="f16", q_extra=0):
    q, l1 = datagen.cent_case(seed, B + q_extra, Hq, Hkv, D, C, grid=True)
    ids, table, seq = datagen.page_case(seed, Hkv, C, kv_page_len, B + 1, max_page, seq_lens)
    k, v = datagen.kv_case(seed, kv_page_len, bs, Hkv, D, "BSH", dist="randn", dtype="i8" if kv == "i8" else "f16")
    return q, l1, ids, table, seq, k, v


@pytest.mark.parametrize("Hq,Hkv,D,C,topk,seq_lens,kvdt", [
    (8, 2, 128, 256, 8, [4000, 32 * 128, 129, 77], "f16"),
    (4, 4, 128, 512, 16, [9000, 5 * 128, 128 * 60], "f16"),
    (8, 8, 64, 128, 8, [3000, 1, 2500], "f16"),
    (8, 2, 128, 256, 8, [4000, 32 * 128, 129], "bf16"),
    (16, 1, 128, 256, 4, [6000, 999], "f16"),             # G = 16
])
def test_fusion_vs_oracle_and_separate(H, Hq, Hkv, D, C, topk, seq_lens, kvdt):
    import custom_ops
    B, bs, kv_page_len, max_page = len(seq_lens), 128, 96, 80
    q, l1, ids, table, seq, k, v = _fusion_case(41 + Hq, B, Hq, Hkv, D, C, seq_lens, kv_page_len, max_page, bs, q_extra=1)
    tdt = torch.float16 if kvdt == "f16" else torch.bfloat16
    if kvdt == "bf16":
        k, v = (ref.round_to_bf16(x.astype(np.float32)) for x in (k, v))
    scale = D ** -0.5
    want, wpos, wlen, wmax = ref.sparse_paged_fusion_attention(q, k, v, table, l1, ids, seq, Hq, Hkv, scale, bs, topk, 256)
    qd, kd, vd, l1d = H.dev(q, tdt), H.dev(k, tdt), H.dev(v, tdt), H.dev(l1, tdt)
    td, idsd, seqd = H.dev(table), H.dev(ids), H.dev(seq)
    out, pos, plen, pmax = H.call_fusion(qd, kd, vd, td, l1d, idsd, seqd, Hq, Hkv, scale, bs, topk,
                                         out_dtype=torch.float32)
    assert np.array_equal(pos.cpu().numpy(), wpos)                     # bit-exact page_position
    assert np.array_equal(plen.cpu().numpy(), wlen)
    assert np.array_equal(pmax.cpu().numpy(), wmax)
    out = H.f32(out)
    err = float(parity.rel_err(out[:B], want).max())
    assert err <= TOL, f"max rel err {err:.3e}"
    # fused == separate (fused_ifa_v5_case.py:215-276)
    spos, slen, smax = custom_ops.cent_select(qd, l1d, idsd, td, seqd, k=topk)
    assert torch.equal(spos, pos) and torch.equal(slen, plen) and torch.equal(smax, pmax)
    if all(s % bs == 0 for s in seq_lens):
        sep = H.f32(H.call_attention(qd[:B], kd, vd, td, spos, smax[:, 0].contiguous(), Hq, Hkv, scale, bs,
                                     out_dtype=torch.float32))
        assert float(parity.rel_err(out[:B], sep).max()) <= TOL
    # drop-in dtype
    out16, _, _, _ = H.call_fusion(qd, kd, vd, td, l1d, idsd, seqd, Hq, Hkv, scale, bs, topk)
    assert out16.dtype == tdt
    assert torch.equal(out16[:B].cpu(), torch.from_numpy(out[:B]).to(tdt))


def test_fusion_page_multiple_equals_separate(H):
    """All lengths multiples of the page size: fused output == cent_select + sparse_paged_attention."""
    import custom_ops
    Hq, Hkv, D, C, topk, bs = 8, 2, 128, 256, 8, 128
    seq_lens = [40 * 128, 64 * 128]
    q, l1, ids, table, seq, k, v = _fusion_case(7, 2, Hq, Hkv, D, C, seq_lens, 96, 80, bs)
    qd, kd, vd, l1d, td, idsd, seqd = (H.dev(x) for x in (q, k, v, l1, table, ids, seq))
    out, pos, plen, pmax = H.call_fusion(qd, kd, vd, td, l1d, idsd, seqd, Hq, Hkv, 1.0, bs, topk,
                                         out_dtype=torch.float32)
    spos, slen, smax = custom_ops.cent_select(qd, l1d, idsd, td, seqd, k=topk)
    assert torch.equal(spos, pos)
    sep = H.call_attention(qd, kd, vd, td, spos, smax[:, 0].contiguous(), Hq, Hkv, 1.0, bs, out_dtype=torch.float32)
    assert float(parity.rel_err(H.f32(out), H.f32(sep)).max()) <= TOL


@pytest.mark.parametrize("Hq,Hkv,kvdt", [(8, 8, "f16"), (8, 2, "f16"), (8, 2, "bf16"), (16, 2, "f16")])
def test_dead_tokens_of_the_last_page_are_never_used(H, Hq, Hkv, kvdt):
    """Token slots past total_seq_len hold NaN bit patterns (a cache page is not cleared before reuse): the result
    must not depend on them."""
    D, C, topk, bs, pages = 128, 128, 128, 128, 8    # k = C: every page of a sequence is selected
    seq_lens = [3 * bs + 5, 7 * bs + 127, 1]
    B = len(seq_lens)
    q, l1 = datagen.cent_case(91, B, Hq, Hkv, D, C, grid=True)
    ids, table, seq = datagen.page_case(91, Hkv, C, B * pages, B, pages, seq_lens, aliasing=False)
    k, v = datagen.kv_case(91, B * pages, bs, Hkv, D, "BSH", dist="randn")
    tdt = torch.float16 if kvdt == "f16" else torch.bfloat16
    if kvdt == "bf16":
        k, v = (ref.round_to_bf16(x.astype(np.float32)) for x in (k, v))
    scale = D ** -0.5
    want, wpos, wlen, wmax = ref.sparse_paged_fusion_attention(q, k, v, table, l1, ids, seq, Hq, Hkv, scale, bs, topk, 256)
    assert [int(x) for x in wlen[:, 0, 0]] == [4, 8, 1]
    kd, vd = H.dev(k, tdt), H.dev(v, tdt)
    for b, n in enumerate(seq_lens):      # poison every token slot at or past the sequence end
        for j in range(pages):
            first_dead = min(max(n - j * bs, 0), bs)
            kd[int(table[b, j]), first_dead:] = float("nan")
            vd[int(table[b, j]), first_dead:] = float("nan")
    qd, l1d, td, idsd, seqd = H.dev(q, tdt), H.dev(l1, tdt), H.dev(table), H.dev(ids), H.dev(seq)
    out, pos, plen, pmax = H.call_fusion(qd, kd, vd, td, l1d, idsd, seqd, Hq, Hkv, scale, bs, topk,
                                         out_dtype=torch.float32)
    assert np.array_equal(pos.cpu().numpy(), wpos)
    assert bool(torch.isfinite(out).all())
    assert float(parity.rel_err(H.f32(out), want).max()) <= TOL


@pytest.mark.parametrize("Hq,Hkv,with_offset", [(8, 2, True), (4, 4, False), (8, 1, True)])
def test_int8_kv_vs_oracle(H, Hq, Hkv, with_offset):
    """antiquant path: K_f = (K_i8 + off_k) * scale_k, V_f = (V_i8 + off_v) * scale_v, per channel."""
    D, C, topk, bs = 128, 256, 8, 128
    seq_lens = [4000, 20 * 128, 300]
    B = len(seq_lens)
    q, l1, ids, table, seq, k, v = _fusion_case(3 + Hq, B, Hq, Hkv, D, C, seq_lens, 96, 80, bs, kv="i8")
    a_s, a_o = datagen.antiquant_case(5, Hkv, D, with_offset)
    scale = D ** -0.5
    want, wpos, wlen, wmax = ref.sparse_paged_fusion_attention(
        q, k, v, table, l1, ids, seq, Hq, Hkv, scale, bs, topk, 256,
        antiquant_scale=a_s.astype(np.float32), antiquant_offset=None if a_o is None else a_o.astype(np.float32))
    qd, l1d, td, idsd, seqd = (H.dev(x) for x in (q, l1, table, ids, seq))
    kd, vd = H.dev(k), H.dev(v)
    asd = H.dev(a_s)
    aod = H.dev(a_o) if a_o is not None else None
    out, pos, plen, pmax = H.call_fusion(qd, kd, vd, td, l1d, idsd, seqd, Hq, Hkv, scale, bs, topk,
                                         antiq_scale=asd, antiq_offset=aod, out_dtype=torch.float32)
    assert np.array_equal(pos.cpu().numpy(), wpos)
    err = float(parity.rel_err(H.f32(out), want).max())
    assert err <= TOL, f"max rel err {err:.3e}"
    # standalone op on the same lists
    sep = H.call_attention(qd, kd, vd, td, pos, pmax[:, 0].contiguous() * 0 + 10 ** 9, Hq, Hkv, scale, bs,
                           antiq_scale=asd, antiq_offset=aod, out_dtype=torch.float32)
    full = all(s % bs == 0 for s in seq_lens)
    if full:
        assert float(parity.rel_err(H.f32(sep), want).max()) <= TOL


def test_attention_abi_errors(H):
    from custom_ops._abi import AclError
    q = H.dev(np.zeros((1, 4, 128), np.float16))
    kv = H.dev(np.zeros((4, 128, 4, 128), np.float16))
    table = H.dev(np.zeros((1, 4), np.int32))
    pos = H.dev(np.zeros((1, 4, 4), np.int32))
    with pytest.raises(AclError, match="blockSize"):
        H.call_attention(q, kv, kv, table, pos, [128], 4, 4, 1.0, 100)
    with pytest.raises(AclError, match="antiquant"):
        H.call_attention(q, kv, kv, table, pos, [128], 4, 4, 1.0, 128, antiq_scale=H.dev(np.ones((2, 4, 128), np.float16)))
    with pytest.raises(AclError, match="int8 KV cache needs"):
        H.call_attention(q, kv.to(torch.int8), kv.to(torch.int8), table, pos, [128], 4, 4, 1.0, 128)
    with pytest.raises(AclError, match="blockposition"):
        H.call_attention(q, kv, kv, table, H.dev(np.zeros((1, 3, 4), np.int32)), [128], 4, 4, 1.0, 128)


@pytest.mark.parametrize("Hq,Hkv,kvdt", [(8, 8, "f16"), (8, 2, "bf16"), (8, 2, "i8")])
def test_page_list_split_equals_single_workgroup(H, Hq, Hkv, kvdt):
    """A small batch is dealt to several workgroups per item; with the split turned off the same launch runs one
    workgroup per item.  Integer outputs must be identical, the attention equal up to the merge order."""
    from custom_ops import _abi
    D, C, topk, bs = 128, 256, 8, 128
    seq_lens = [70 * bs + 9, 3 * bs, 1]
    B = len(seq_lens)
    q, l1, ids, table, seq, k, v = _fusion_case(77, B, Hq, Hkv, D, C, seq_lens, 96, 80, bs,
                                                kv="i8" if kvdt == "i8" else "f16")
    tdt = torch.bfloat16 if kvdt == "bf16" else torch.float16
    kd, vd = (H.dev(k), H.dev(v)) if kvdt == "i8" else (H.dev(k, tdt), H.dev(v, tdt))
    qd, l1d, td, idsd, seqd = H.dev(q, tdt), H.dev(l1, tdt), H.dev(table), H.dev(ids), H.dev(seq)
    kw = {}
    if kvdt == "i8":
        a_s, a_o = datagen.antiquant_case(5, Hkv, D, True)
        kw = dict(antiq_scale=H.dev(a_s), antiq_offset=H.dev(a_o))
    res = {}
    for max_split in (1, 16, 3):
        _abi.check(_abi.spfaSetMaxSplit(max_split), "spfaSetMaxSplit")
        try:
            res[max_split] = H.call_fusion(qd, kd, vd, td, l1d, idsd, seqd, Hq, Hkv, D ** -0.5, bs, topk,
                                           out_dtype=torch.float32, **kw)
            spos = res[max_split][1]
            sep = H.call_attention(qd, kd, vd, td, spos, torch.tensor([10 ** 9] * B), Hq, Hkv, D ** -0.5, bs,
                                   out_dtype=torch.float32, **kw)
            res[("sep", max_split)] = sep
        finally:
            _abi.spfaSetMaxSplit(16)
    # the split workgroups of an item score its rows in one shared pass; without that every one scores alone
    _abi.check(_abi.spfaSetSharedScoring(0), "spfaSetSharedScoring")
    try:
        alone = H.call_fusion(qd, kd, vd, td, l1d, idsd, seqd, Hq, Hkv, D ** -0.5, bs, topk, out_dtype=torch.float32, **kw)
    finally:
        _abi.spfaSetSharedScoring(1)
    out1, pos1, len1, max1 = res[1]
    scale = float(out1.abs().max())
    # integer outputs bit for bit; the attention up to the order in which the waves claimed their pages
    assert all(torch.equal(x, y) for x, y in zip(alone[1:], res[16][1:]))
    assert float((alone[0] - res[16][0]).abs().max()) <= 2e-6 * scale
    for ms in (16, 3):
        out, pos, plen, pmax = res[ms]
        assert torch.equal(pos, pos1) and torch.equal(plen, len1) and torch.equal(pmax, max1)
        assert float((out - out1).abs().max()) <= 2e-6 * scale
        assert float((res[("sep", ms)] - res[("sep", 1)]).abs().max()) <= 2e-6 * scale


@pytest.mark.parametrize("B,kvdt", [(32, "f16"), (48, "i8"), (64, "f16")])
def test_items_scored_ahead_equal_items_scored_one_by_one(H, B, kvdt):
    """A launch of exactly J items per resident workgroup (32 kv heads, one q head per pass: 512 workgroups on a 256-CU
    part, so B = 32 / 48 / 64 gives J = 2 / 3 / 4) scores the J rows of a workgroup in one pass over the head's
    centroids before the first item is streamed.  Ragged lengths; compared with the oracle and with the same launch
    under spfaSetSharedScoring(0), which scores every item when its turn comes: integer outputs bit for bit, the
    attention up to the order in which waves claimed pages.  (On a part with another workgroup count the shapes simply
    take the ordinary path and the test still holds.)"""
    from custom_ops import _abi
    Hq = Hkv = 32
    D, C, topk, bs, kv_page_len, max_page = 128, 128, 8, 32, 96, 40
    rng = datagen.rng_for(900 + B)
    seq_lens = [int(x) for x in rng.integers(1, max_page * bs, size=B)]
    seq_lens[0], seq_lens[B - 1] = max_page * bs, 1
    q, l1, ids, table, seq, k, v = _fusion_case(300 + B, B, Hq, Hkv, D, C, seq_lens, kv_page_len, max_page, bs,
                                                kv="i8" if kvdt == "i8" else "f16")
    kd, vd = (H.dev(k), H.dev(v)) if kvdt == "i8" else (H.dev(k, torch.float16), H.dev(v, torch.float16))
    qd, l1d, td, idsd, seqd = H.dev(q, torch.float16), H.dev(l1, torch.float16), H.dev(table), H.dev(ids), H.dev(seq)
    kw, okw = {}, {}
    if kvdt == "i8":
        a_s, a_o = datagen.antiquant_case(5, Hkv, D, True)
        kw = dict(antiq_scale=H.dev(a_s), antiq_offset=H.dev(a_o))
        okw = dict(antiquant_scale=a_s.astype(np.float32), antiquant_offset=a_o.astype(np.float32))
    scale = D ** -0.5
    ahead = H.call_fusion(qd, kd, vd, td, l1d, idsd, seqd, Hq, Hkv, scale, bs, topk, out_dtype=torch.float32, **kw)
    _abi.check(_abi.spfaSetSharedScoring(0), "spfaSetSharedScoring")
    try:
        one_by_one = H.call_fusion(qd, kd, vd, td, l1d, idsd, seqd, Hq, Hkv, scale, bs, topk, out_dtype=torch.float32, **kw)
    finally:
        _abi.spfaSetSharedScoring(1)
    assert all(torch.equal(x, y) for x, y in zip(ahead[1:], one_by_one[1:]))
    assert float((ahead[0] - one_by_one[0]).abs().max()) <= 2e-6 * float(one_by_one[0].abs().max())
    want, wpos, wlen, wmax = ref.sparse_paged_fusion_attention(q, k, v, table, l1, ids, seq, Hq, Hkv, scale, bs, topk, 256, **okw)
    out, pos, plen, pmax = ahead
    assert np.array_equal(pos.cpu().numpy(), wpos)
    assert np.array_equal(plen.cpu().numpy(), wlen)
    assert np.array_equal(pmax.cpu().numpy(), wmax)
    assert float(parity.rel_err(H.f32(out), want).max()) <= TOL


@pytest.mark.parametrize("B,kvdt", [(8, "f16"), (40, "f16"), (64, "i8"), (96, "f16")])
def test_wide_and_narrow_workgroups_agree(H, B, kvdt):
    """The one-q-head kernels exist as one workgroup of eight streamer waves per CU and as two of four; the planner
    picks by launch shape (B = 96 x 32 heads is more than eight items per wide workgroup: narrow).  Both forced in turn,
    with and without the items scored ahead: integer outputs bit for bit, the attention to fp32 rounding; the planner's
    own choice against the oracle."""
    from custom_ops import _abi
    Hq = Hkv = 32
    D, C, topk, bs, kv_page_len, max_page = 128, 128, 8, 32, 96, 40
    rng = datagen.rng_for(950 + B)
    seq_lens = [int(x) for x in rng.integers(1, max_page * bs, size=B)]
    seq_lens[0], seq_lens[B - 1] = max_page * bs, 1
    q, l1, ids, table, seq, k, v = _fusion_case(350 + B, B, Hq, Hkv, D, C, seq_lens, kv_page_len, max_page, bs,
                                                kv="i8" if kvdt == "i8" else "f16")
    kd, vd = (H.dev(k), H.dev(v)) if kvdt == "i8" else (H.dev(k, torch.float16), H.dev(v, torch.float16))
    qd, l1d, td, idsd, seqd = H.dev(q, torch.float16), H.dev(l1, torch.float16), H.dev(table), H.dev(ids), H.dev(seq)
    kw, okw = {}, {}
    if kvdt == "i8":
        a_s, a_o = datagen.antiquant_case(5, Hkv, D, True)
        kw = dict(antiq_scale=H.dev(a_s), antiq_offset=H.dev(a_o))
        okw = dict(antiquant_scale=a_s.astype(np.float32), antiquant_offset=a_o.astype(np.float32))
    scale = D ** -0.5
    auto = H.call_fusion(qd, kd, vd, td, l1d, idsd, seqd, Hq, Hkv, scale, bs, topk, out_dtype=torch.float32, **kw)
    res = {}
    try:
        for geo in (1, 2):
            for shared in (1, 0):
                _abi.check(_abi.spfaSetWorkgroupGeometry(geo), "spfaSetWorkgroupGeometry")
                _abi.check(_abi.spfaSetSharedScoring(shared), "spfaSetSharedScoring")
                res[(geo, shared)] = H.call_fusion(qd, kd, vd, td, l1d, idsd, seqd, Hq, Hkv, scale, bs, topk,
                                                   out_dtype=torch.float32, **kw)
    finally:
        _abi.spfaSetWorkgroupGeometry(0)
        _abi.spfaSetSharedScoring(1)
    big = float(auto[0].abs().max())
    for key, r in res.items():
        assert all(torch.equal(x, y) for x, y in zip(r[1:], auto[1:])), key
        assert float((r[0] - auto[0]).abs().max()) <= 2e-6 * big, key
    want, wpos, wlen, wmax = ref.sparse_paged_fusion_attention(q, k, v, table, l1, ids, seq, Hq, Hkv, scale, bs, topk, 256, **okw)
    out, pos, plen, pmax = auto
    assert np.array_equal(pos.cpu().numpy(), wpos) and np.array_equal(plen.cpu().numpy(), wlen)
    assert np.array_equal(pmax.cpu().numpy(), wmax)
    assert float(parity.rel_err(H.f32(out), want).max()) <= TOL
    # lists given (the standalone op): both geometries as well
    act = pmax[:, 0].contiguous()
    outs = []
    try:
        for geo in (1, 2):
            _abi.check(_abi.spfaSetWorkgroupGeometry(geo), "spfaSetWorkgroupGeometry")
            outs.append(H.call_attention(qd, kd, vd, td, pos, act, Hq, Hkv, scale, bs, out_dtype=torch.float32, **kw))
    finally:
        _abi.spfaSetWorkgroupGeometry(0)
    assert float((outs[0] - outs[1]).abs().max()) <= 2e-6 * big


@pytest.mark.parametrize("dtype,layout,Hq,Hkv,D,bs,B,max_page,hi", [
    ("f16", "BSH", 8, 8, 128, 128, 3, 40, 40),       # quarter pages of 32 tokens
    ("f16", "BNSD", 4, 4, 128, 16, 3, 60, 60),       # pages of two steps: halves
    ("f16", "BSH", 6, 2, 128, 48, 3, 30, 30),        # group of 3 through the VALU streamer; 48-token pages: halves
    ("bf16", "BSH", 4, 4, 64, 256, 2, 20, 20),
    ("i8", "BSH", 8, 8, 128, 128, 3, 40, 40),
    ("i8", "BSH", 8, 2, 128, 64, 3, 40, 40),         # int8 group pass (MFMA streamer, 32-token tiles)
    ("f16", "BSH", 32, 4, 128, 128, 2, 50, 50),      # group of 8: MFMA streamer
    ("bf16", "BSH", 16, 2, 128, 48, 2, 30, 30),      # MFMA streamer, a page of one and a half tiles: whole pages
    ("f16", "BSH", 2, 2, 128, 16, 1, 1500, 1500),    # lists far longer than the chunk count: chunks of several units
    ("f16", "BSH", 4, 4, 512, 32, 2, 24, 24),
])
def test_shared_last_items_in_sub_page_chunks(H, dtype, layout, Hq, Hkv, D, bs, B, max_page, hi):
    """The last item of every workgroup is streamed in chunks (down to quarter pages) that other workgroups may take:
    forced on here for small launches, with ragged lengths (partly filled last pages), lists from one page to the
    whole table, every KV dtype and both streamers; compared with the oracle and with the launch without sharing."""
    from custom_ops import _abi
    rng = datagen.rng_for(4321 + Hq * 5 + D + bs)
    kvdt = "i8" if dtype == "i8" else "f16"
    n_blocks = min(B * max_page, 200)
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    q = datagen.ref_rand_f16(rng, (B, Hq, D))
    k, v = datagen.kv_case(9, n_blocks, bs, Hkv, D, layout, dist="randn", dtype=kvdt)
    sc = off = None
    if dtype == "i8":
        sc, off = datagen.antiquant_case(9, Hkv, D)
    if dtype == "bf16":
        q, k, v = (ref.round_to_bf16(x.astype(np.float32)) for x in (q, k, v))
    table = rng.integers(0, n_blocks, size=(B, max_page)).astype(np.int32)
    pos = _random_lists(rng, B, Hkv, max_page, hi, 1, hi)
    pos[0, 0, :] = np.arange(hi, dtype=np.int32)                                          # a full list
    act = [int(x) for x in rng.integers(1, hi * bs + 1, size=B)]
    act[0] = hi * bs - 1                                                                  # all but one token
    scale = 1.0 / np.sqrt(D)
    want = ref.sparse_paged_attention(q, k, v, table, pos, act, Hq, Hkv, scale, bs, layout, sc, off)
    qd = H.dev(q, tdt)
    kd, vd = (H.dev(x) if dtype == "i8" else H.dev(x, tdt) for x in (k, v))
    kw = dict(antiq_scale=H.dev(sc, tdt), antiq_offset=H.dev(off, tdt)) if dtype == "i8" else {}
    outs = {}
    _abi.check(_abi.spfaSetMaxSplit(1), "spfaSetMaxSplit")
    try:
        for tail in (0, 2, 2):
            _abi.check(_abi.spfaSetTailSharing(tail), "spfaSetTailSharing")
            out = H.f32(H.call_attention(qd, kd, vd, H.dev(table), H.dev(pos), torch.tensor(act, dtype=torch.int64),
                                         Hq, Hkv, scale, bs, out_dtype=torch.float32, **kw))
            assert float(parity.rel_err(out, want).max()) <= TOL, tail
            outs.setdefault(tail, out)
    finally:
        _abi.spfaSetTailSharing(1)
        _abi.spfaSetMaxSplit(16)
    assert float(np.abs(outs[2] - outs[0]).max()) <= 4e-6 * float(np.abs(outs[0]).max())
