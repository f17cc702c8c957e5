"""GPU parity of the optional operands of the attention core that the sparse paged ops inherit from the
incremental-flash-attention interface: pse_shift, attention_mask, the int8 output quantisation
(quant_scale2 / quant_offset2) and per-tensor antiquant parameters.

The reference holds no CPU path and no test for these operands on the sparse ops ("parity unpinned"): the
oracle restates the kernel code (sparse_paged_fusion_attention_split_Bbn2s2_Us2.h:3363-3404 score bias and
mask, :2192-2194,2239-2244 their column offsets, :4016-4111 output quantisation, :2036-2038 per-tensor
parameters).  Tolerance 1e-3 relative on float outputs; int8 outputs must equal the oracle's quantisation of
the kernel's own float32 result exactly.
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


def _lists(rng, B, rows, max_page, P, lo, hi):
    pos = np.full((B, rows, P), SENT, np.int32)
    for b in range(B):
        for r in range(rows):
            n = int(rng.integers(lo, hi + 1))
            pos[b, r, :n] = np.sort(rng.choice(max_page, size=n, replace=False))
    return pos


def _case(seed, dtype, Hq, Hkv, rows, D, bs, B=3, n_blocks=40, max_page=24, P=12, layout="BSH"):
    rng = datagen.rng_for(seed)
    q = datagen.ref_rand_f16(rng, (B, Hq, D))
    k, v = datagen.kv_case(seed, n_blocks, bs, Hkv, D, layout, dist="randn", dtype="i8" if dtype == "i8" else "f16")
    if dtype == "bf16":
        q, k, v = (ref.round_to_bf16(x.astype(np.float32)) for x in (q, k, v))
    table = rng.integers(0, n_blocks, size=(B, max_page)).astype(np.int32)
    pos = _lists(rng, B, Hq if rows == "q" else Hkv, max_page, P, 1, min(P - 2, max_page - 1))
    act = [int(x) for x in rng.integers(1, P * bs, size=B)]
    return rng, q, k, v, table, pos, act


def _pse_mask(rng, B, Hq, S, pse_batch, dtype, with_pse=True, with_mask=True):
    pse = mask = None
    if with_pse:
        pse = (rng.standard_normal((B if pse_batch == "B" else 1, Hq, 1, S)) * 2.0).astype(np.float16)
        if dtype == "bf16":
            pse = ref.round_to_bf16(pse.astype(np.float32))
    if with_mask:
        mask = rng.random((B, 1, 1, S)) < 0.3
    return pse, mask


@pytest.mark.parametrize("dtype,Hq,Hkv,rows,D,bs,pse_batch,which", [
    ("f16", 8, 8, "kv", 128, 128, "B", "both"),
    ("f16", 8, 2, "kv", 128, 128, "1", "both"),       # shared lists, GQA: every head becomes its own pass
    ("f16", 8, 2, "q", 128, 64, "B", "both"),         # one list per q head
    ("bf16", 16, 2, "kv", 64, 32, "B", "pse"),
    ("bf16", 4, 4, "kv", 256, 16, "1", "mask"),
    ("i8", 8, 2, "kv", 128, 128, "B", "both"),
    ("i8", 4, 4, "q", 128, 32, "1", "mask"),
])
@pytest.mark.parametrize("max_split", [16, 1])
def test_pse_and_mask_vs_oracle(H, dtype, Hq, Hkv, rows, D, bs, pse_batch, which, max_split):
    from custom_ops import _abi
    rng, q, k, v, table, pos, act = _case(400 + Hq + D, dtype, Hq, Hkv, rows, D, bs)
    B, S = q.shape[0], table.shape[1] * bs
    pse, mask = _pse_mask(rng, B, Hq, S, pse_batch, dtype, which in ("both", "pse"), which in ("both", "mask"))
    kw, okw = {}, {}
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    if dtype == "i8":
        a_s, a_o = datagen.antiquant_case(5, Hkv, D, True)
        kw = dict(antiq_scale=H.dev(a_s), antiq_offset=H.dev(a_o))
        okw = dict(antiquant_scale=a_s.astype(np.float32), antiquant_offset=a_o.astype(np.float32))
    scale = D ** -0.5
    want = ref.sparse_paged_attention(q, k, v, table, pos, act, Hq, Hkv, scale, bs, "BSH", pse_shift=pse,
                                      atten_mask=mask, **okw)
    plain = ref.sparse_paged_attention(q, k, v, table, pos, act, Hq, Hkv, scale, bs, "BSH", **okw)
    assert float(np.abs(want - plain).max()) > 1e-2 * float(np.abs(plain).max())   # the operands matter
    kd, vd = (H.dev(k), H.dev(v)) if dtype == "i8" else (H.dev(k, tdt), H.dev(v, tdt))
    _abi.check(_abi.spfaSetMaxSplit(max_split), "spfaSetMaxSplit")
    try:
        out = H.call_attention(H.dev(q, tdt), kd, vd, H.dev(table), H.dev(pos), torch.tensor(act), Hq, Hkv, scale, bs,
                               out_dtype=torch.float32, pse=None if pse is None else H.dev(pse, tdt),
                               mask=None if mask is None else H.dev(mask), **kw)
    finally:
        _abi.spfaSetMaxSplit(16)
    err = float(parity.rel_err(H.f32(out), want).max())
    assert err <= TOL, f"max rel err {err:.3e}"


def test_zero_bias_and_clear_mask_change_nothing(H):
    """pse_shift == 0 and an all-false mask: the per-head passes must agree with the grouped (matrix-core) pass."""
    rng, q, k, v, table, pos, act = _case(77, "f16", 8, 2, "kv", 128, 128)
    B, Hq, bs = q.shape[0], 8, 128
    S = table.shape[1] * bs
    args = (H.dev(q), H.dev(k), H.dev(v), H.dev(table), H.dev(pos), torch.tensor(act), Hq, 2, 128 ** -0.5, bs)
    base = H.call_attention(*args, out_dtype=torch.float32)
    out = H.call_attention(*args, out_dtype=torch.float32, pse=H.dev(np.zeros((1, Hq, 1, S), np.float16)),
                           mask=H.dev(np.zeros((B, S), np.uint8)))
    assert float(parity.rel_err(H.f32(out), H.f32(base)).max()) <= 1e-4


def test_fully_masked_rows_average_their_tokens(H):
    """Every column masked: all scores are -1e12, the softmax is uniform (…Us2.h:860,3398)."""
    rng, q, k, v, table, pos, act = _case(78, "f16", 4, 4, "kv", 128, 128)
    B, Hq, bs = q.shape[0], 4, 128
    S = table.shape[1] * bs
    mask = np.ones((B, S), np.uint8)
    out = H.f32(H.call_attention(H.dev(q), H.dev(k), H.dev(v), H.dev(table), H.dev(pos), torch.tensor(act), Hq, Hq,
                                 1.0, bs, out_dtype=torch.float32, mask=H.dev(mask)))
    want = ref.sparse_paged_attention(q, k, v, table, pos, act, Hq, Hq, 1.0, bs, "BSH", atten_mask=mask)
    zero_q = ref.sparse_paged_attention(np.zeros_like(q), k, v, table, pos, act, Hq, Hq, 1.0, bs, "BSH")
    assert float(np.abs(want - zero_q).max()) < 1e-6          # uniform weights = the plain mean of the tokens
    assert float(parity.rel_err(out, want).max()) <= TOL


def test_bias_columns_follow_the_list_in_a_shared_last_item(H):
    """More items than resident workgroups and long list rows: the last item of every workgroup is streamed in
    chunks by several workgroups.  The column of a token must still be its slot in the head's list."""
    Hq, D, bs, B, P = 32, 128, 16, 40, 512
    rng, q, k, v, table, pos, act = _case(79, "f16", Hq, Hq, "kv", D, bs, B=B, n_blocks=64, max_page=24, P=P)
    act = [P * bs] * B
    S = table.shape[1] * bs
    pse, mask = _pse_mask(rng, B, Hq, S, "B", "f16")
    want = ref.sparse_paged_attention(q, k, v, table, pos, act, Hq, Hq, D ** -0.5, bs, "BSH", pse_shift=pse,
                                      atten_mask=mask)
    out = H.call_attention(H.dev(q), H.dev(k), H.dev(v), H.dev(table), H.dev(pos), torch.tensor(act), Hq, Hq,
                           D ** -0.5, bs, out_dtype=torch.float32, pse=H.dev(pse), mask=H.dev(mask))
    assert float(parity.rel_err(H.f32(out), want).max()) <= TOL


@pytest.mark.parametrize("Hq,Hkv,kvdt", [(8, 2, "f16"), (4, 4, "bf16"), (8, 2, "i8")])
def test_fusion_with_pse_and_mask(H, Hq, Hkv, kvdt):
    D, C, topk, bs = 128, 256, 8, 128
    seq_lens = [4000, 32 * 128, 129, 77]
    B, kv_page_len, max_page = len(seq_lens), 96, 80
    q, l1 = datagen.cent_case(61 + Hq, B, Hq, Hkv, D, C, grid=True)
    ids, table, seq = datagen.page_case(61 + Hq, Hkv, C, kv_page_len, B + 1, max_page, seq_lens)
    k, v = datagen.kv_case(61, kv_page_len, bs, Hkv, D, "BSH", dist="randn", dtype="i8" if kvdt == "i8" else "f16")
    tdt = torch.bfloat16 if kvdt == "bf16" else torch.float16
    if kvdt == "bf16":
        k, v = (ref.round_to_bf16(x.astype(np.float32)) for x in (k, v))
    rng = datagen.rng_for(5)
    pse, mask = _pse_mask(rng, B, Hq, max_page * bs, "B", kvdt)
    kw, okw = {}, {}
    if kvdt == "i8":
        a_s, a_o = datagen.antiquant_case(5, Hkv, D, True)
        kw = dict(antiq_scale=H.dev(a_s), antiq_offset=H.dev(a_o))
        okw = dict(antiquant_scale=a_s.astype(np.float32), antiquant_offset=a_o.astype(np.float32))
    scale = D ** -0.5
    want, wpos, wlen, wmax = ref.sparse_paged_fusion_attention(q, k, v, table, l1, ids, seq, Hq, Hkv, scale, bs, topk,
                                                               256, pse_shift=pse, atten_mask=mask, **okw)
    kd, vd = (H.dev(k), H.dev(v)) if kvdt == "i8" else (H.dev(k, tdt), H.dev(v, tdt))
    out, pos, plen, pmax = H.call_fusion(H.dev(q, tdt), kd, vd, H.dev(table), H.dev(l1, tdt), H.dev(ids), H.dev(seq),
                                         Hq, Hkv, scale, bs, topk, out_dtype=torch.float32, pse=H.dev(pse, tdt),
                                         mask=H.dev(mask), **kw)
    assert np.array_equal(pos.cpu().numpy(), wpos) and np.array_equal(plen.cpu().numpy(), wlen)
    assert np.array_equal(pmax.cpu().numpy(), wmax)
    err = float(parity.rel_err(H.f32(out), want).max())
    assert err <= TOL, f"max rel err {err:.3e}"


@pytest.mark.parametrize("dtype,Hq,Hkv,per_channel,param_dtype,with_offset", [
    ("f16", 8, 2, False, "f32", True),
    ("f16", 8, 8, True, "f32", False),
    ("bf16", 8, 2, True, "bf16", True),
    ("bf16", 4, 4, False, "bf16", False),
    ("i8", 8, 2, True, "f32", True),
])
def test_int8_output_quantisation(H, dtype, Hq, Hkv, per_channel, param_dtype, with_offset):
    D, bs = 128, 128
    rng, q, k, v, table, pos, act = _case(500 + Hq, dtype, Hq, Hkv, "kv", D, bs)
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    n = Hq * D if per_channel else 1
    sc = (rng.random(n) * 600 + 300).astype(np.float32)        # randn V averaged over many tokens: outputs ~ 0.05
    of = (rng.standard_normal(n) * 3).astype(np.float32) if with_offset else None
    if param_dtype == "bf16":
        sc = ref.round_to_bf16(sc)
        of = None if of is None else ref.round_to_bf16(of)
    pdt = torch.bfloat16 if param_dtype == "bf16" else torch.float32
    kw, okw = {}, {}
    if dtype == "i8":
        a_s, a_o = datagen.antiquant_case(5, Hkv, D, True)
        kw = dict(antiq_scale=H.dev(a_s), antiq_offset=H.dev(a_o))
        okw = dict(antiquant_scale=a_s.astype(np.float32), antiquant_offset=a_o.astype(np.float32))
    kd, vd = (H.dev(k), H.dev(v)) if dtype == "i8" else (H.dev(k, tdt), H.dev(v, tdt))
    args = (H.dev(q, tdt), kd, vd, H.dev(table), H.dev(pos), torch.tensor(act), Hq, Hkv, D ** -0.5, bs)
    shape = (1, Hq, 1, D) if per_channel else (1,)
    out8 = H.call_attention(*args, quant_scale2=H.dev(sc.reshape(shape), pdt),
                            quant_offset2=None if of is None else H.dev(of.reshape(shape), pdt), **kw)
    assert out8.dtype == torch.int8 and out8.shape == (q.shape[0], Hq, D)
    out32 = H.f32(H.call_attention(*args, out_dtype=torch.float32, **kw))
    got = out8.cpu().numpy().astype(np.float32)
    assert np.array_equal(got, ref.post_quant(out32, sc, of))               # the kernel's own fp32 result, quantised
    assert len(np.unique(got)) > 20 and np.abs(got).max() <= 128
    want = ref.sparse_paged_attention(q, k, v, table, pos, act, Hq, Hkv, D ** -0.5, bs, "BSH", quant_scale2=sc,
                                      quant_offset2=of, **okw)
    diff = np.abs(got - want)
    assert diff.max() <= 1 and (diff > 0).mean() < 0.02                     # vs the float64 oracle: rare 1-step flips


def test_int8_output_saturates_and_rounds_half_away(H):
    """Constant V makes the output exact, so the quantisation steps can be checked value by value."""
    Hq, D, bs, B = 4, 128, 16, 1
    k = np.zeros((4, bs, Hq, D), np.float16)
    v = np.zeros((4, bs, Hq, D), np.float16)
    vals = [0.5, -0.5, 2.5, 100.0]
    for h, x in enumerate(vals):
        v[:, :, h, :] = x
    table = np.arange(4, dtype=np.int32).reshape(1, 4)
    pos = np.tile(np.array([0, 1, SENT, SENT], np.int32), (B, Hq, 1))
    q = np.full((B, Hq, D), 0.1, np.float16)
    for sc, of, want in [(1.0, None, [1, -1, 3, 100]), (2.0, 0.5, [2, -1, 6, 127]), (-3.0, None, [-2, 2, -8, -128]),
                         (1.0, 0.25, [1, 0, 3, 100])]:
        out = H.call_attention(H.dev(q), H.dev(k), H.dev(v), H.dev(table), H.dev(pos), [2 * bs], Hq, Hq, 1.0, bs,
                               quant_scale2=H.dev(np.array([sc], np.float32)),
                               quant_offset2=None if of is None else H.dev(np.array([of], np.float32)))
        got = out.cpu().numpy()[0, :, 0].tolist()
        assert got == want, (sc, of, got)
        assert (out.cpu().numpy() == np.array(want, np.int8)[None, :, None]).all()


@pytest.mark.parametrize("Hq,Hkv,with_offset", [(8, 2, True), (4, 4, False), (16, 1, True)])
def test_per_tensor_antiquant(H, Hq, Hkv, with_offset):
    D, bs = 128, 128
    rng, q, k, v, table, pos, act = _case(600 + Hq, "i8", Hq, Hkv, "kv", D, bs)
    a_s = np.array([0.013, 0.021], np.float16)
    a_o = np.array([3.0, -1.5], np.float16) if with_offset else None
    want = ref.sparse_paged_attention(q, k, v, table, pos, act, Hq, Hkv, D ** -0.5, bs, "BSH",
                                      antiquant_scale=a_s.astype(np.float32),
                                      antiquant_offset=None if a_o is None else a_o.astype(np.float32))
    out = H.call_attention(H.dev(q), H.dev(k), H.dev(v), H.dev(table), H.dev(pos), torch.tensor(act), Hq, Hkv,
                           D ** -0.5, bs, out_dtype=torch.float32, antiq_scale=H.dev(a_s),
                           antiq_offset=None if a_o is None else H.dev(a_o))
    err = float(parity.rel_err(H.f32(out), want).max())
    assert err <= TOL, f"max rel err {err:.3e}"
    # the same parameters spelled per channel give the same result
    full_s = np.broadcast_to(a_s.reshape(2, 1, 1), (2, Hkv, D)).copy()
    full_o = None if a_o is None else np.broadcast_to(a_o.reshape(2, 1, 1), (2, Hkv, D)).copy()
    out2 = H.call_attention(H.dev(q), H.dev(k), H.dev(v), H.dev(table), H.dev(pos), torch.tensor(act), Hq, Hkv,
                            D ** -0.5, bs, out_dtype=torch.float32, antiq_scale=H.dev(full_s),
                            antiq_offset=None if full_o is None else H.dev(full_o))
    assert torch.equal(out, out2)


def test_operand_errors(H):
    from custom_ops._abi import AclError
    B, Hq, D, bs = 2, 4, 128, 128
    q = H.dev(np.zeros((B, Hq, D), np.float16))
    kv = H.dev(np.zeros((4, bs, Hq, D), np.float16))
    table = H.dev(np.zeros((B, 4), np.int32))
    pos = H.dev(np.zeros((B, Hq, 4), np.int32))
    args = (q, kv, kv, table, pos, [128], Hq, Hq, 1.0, bs)
    with pytest.raises(AclError, match="pseShift"):
        H.call_attention(*args, pse=H.dev(np.zeros((B, Hq, 1, 4 * bs - 1), np.float16)))      # S < sMax
    with pytest.raises(AclError, match="pseShift"):
        H.call_attention(*args, pse=H.dev(np.zeros((3, Hq, 1, 4 * bs), np.float16)))          # batch 3
    with pytest.raises(AclError, match="pseShift dtype"):
        H.call_attention(*args, pse=H.dev(np.zeros((B, Hq, 1, 4 * bs), np.float32)))
    with pytest.raises(AclError, match="attenMask"):
        H.call_attention(*args, mask=H.dev(np.zeros((B + 1, 4 * bs), np.uint8)))
    with pytest.raises(AclError, match="attenMask"):
        H.call_attention(*args, mask=H.dev(np.zeros((B, 4 * bs), np.int32)))
    with pytest.raises(AclError, match="quantScale2 is absent"):
        H.call_attention(*args, out_dtype=torch.int8)
    with pytest.raises(AclError, match="not int8, but quantScale2"):
        H.call_attention(*args, out_dtype=torch.float16, quant_scale2=H.dev(np.ones(1, np.float32)))
    with pytest.raises(AclError, match="bfloat16 quantScale2"):
        H.call_attention(*args, quant_scale2=H.dev(np.ones(1, np.float32), torch.bfloat16))
    with pytest.raises(AclError, match="quantScale2 must hold"):
        H.call_attention(*args, quant_scale2=H.dev(np.ones(7, np.float32)))
