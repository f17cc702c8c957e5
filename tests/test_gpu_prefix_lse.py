"""GPU parity of the two operands of the reference KERNEL that the aclnn signatures of the sparse ops leave out and
that arrive here through one-shot settings (`spfaSetSoftmaxLseOut`, `spfaSetSharedPrefix`; keywords of `custom_ops`):
the softmax log-sum-exp output and the shared prefix.

No CPU path, test or fixture of the reference covers either on the sparse ops ("parity unpinned"): the oracle restates
the kernel source (sparse_paged_fusion_attention_split_Bbn2s2_Us2.h:3130-3159 LSE, :1289-1300 its value for empty rows,
:2651-2659 the key-offset term of int8 KV; :1418-1439 prefix operands, :2263-2266,2283-2285 the column shift of
pse_shift / atten_mask).  Independent pins: LSE against torch.logsumexp over the gathered scores; a prefix made of the
same rows as leading pages of the list must give the output of the plain op on the longer list.
Tolerance: attention_out 1e-3 relative (against max(|want|, 1 % of the largest |want|)); LSE 1e-4 absolute + 1e-5 relative.
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
    pos = _lists(rng, B, Hq if rows == "q" else Hkv, max_page, P, 0, min(P - 2, max_page - 1))
    act = [int(x) for x in rng.integers(1, P * bs, size=B)]
    return rng, q, k, v, table, pos, act


def _prefix(rng, dtype, S, Hkv, D, layout):
    if dtype == "i8":
        kp = rng.integers(-127, 128, size=(S, Hkv, D)).astype(np.int8)
        vp = rng.integers(-127, 128, size=(S, Hkv, D)).astype(np.int8)
    else:
        kp = rng.standard_normal((S, Hkv, D)).astype(np.float16)
        vp = rng.standard_normal((S, Hkv, D)).astype(np.float16)
        if dtype == "bf16":
            kp, vp = (ref.round_to_bf16(x.astype(np.float32)) for x in (kp, vp))
    if layout == "BSH":
        return kp.reshape(1, S, Hkv * D), vp.reshape(1, S, Hkv * D)
    return (np.ascontiguousarray(kp.transpose(1, 0, 2)).reshape(1, Hkv, S, D),
            np.ascontiguousarray(vp.transpose(1, 0, 2)).reshape(1, Hkv, S, D))


def _quant(H, dtype, Hkv, D, with_offset=True):
    if dtype != "i8":
        return {}, {}
    a_s, a_o = datagen.antiquant_case(5, Hkv, D, True)
    kw = dict(antiq_scale=H.dev(a_s))
    okw = dict(antiquant_scale=a_s.astype(np.float32))
    if with_offset:
        kw["antiq_offset"] = H.dev(a_o)
        okw["antiquant_offset"] = a_o.astype(np.float32)
    return kw, okw


def _lse_close(got, want):
    live = want > -1e38
    assert np.array_equal(got[~live], want[~live]), "rows without tokens must hold -3.4e38"
    err = np.abs(got[live] - want[live])
    assert (err <= 1e-4 + 1e-5 * np.abs(want[live])).all(), f"LSE max abs err {err.max():.3e}"


@pytest.mark.parametrize("dtype,Hq,Hkv,rows,D,bs,layout", [
    ("f16", 8, 8, "kv", 128, 128, "BSH"),
    ("f16", 8, 2, "kv", 128, 128, "BSH"),        # group pass on the matrix-core streamer
    ("f16", 8, 2, "q", 128, 64, "BNSD"),
    ("bf16", 16, 2, "kv", 64, 32, "BSH"),
    ("i8", 8, 2, "kv", 128, 128, "BSH"),         # key offset: the row constant is added back
    ("i8", 4, 4, "q", 128, 32, "BNSD"),
])
@pytest.mark.parametrize("max_split", [16, 1])
def test_lse_vs_oracle(H, dtype, Hq, Hkv, rows, D, bs, layout, max_split):
    from custom_ops import _abi
    rng, q, k, v, table, pos, act = _case(700 + Hq + D, dtype, Hq, Hkv, rows, D, bs, layout=layout)
    pos[0, 0, :] = SENT                                     # a row that attends nothing
    kw, okw = _quant(H, dtype, Hkv, D)
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    scale = D ** -0.5
    want, wlse = ref.sparse_paged_attention(q, k, v, table, pos, act, Hq, Hkv, scale, bs, layout, return_lse=True, **okw)
    kd, vd = (H.dev(k), H.dev(v)) if dtype == "i8" else (H.dev(k, tdt), H.dev(v, tdt))
    lse = torch.full((q.shape[0], Hq), 7.0, dtype=torch.float32, device=H.DEV)
    _abi.check(_abi.spfaSetMaxSplit(max_split), "spfaSetMaxSplit")
    try:
        out = H.call_attention(H.dev(q, tdt), kd, vd, H.dev(table), H.dev(pos), torch.tensor(act), Hq, Hkv, scale, bs,
                               out_dtype=torch.float32, softmax_lse=lse, **kw)
        again = H.call_attention(H.dev(q, tdt), kd, vd, H.dev(table), H.dev(pos), torch.tensor(act), Hq, Hkv, scale, bs,
                                 out_dtype=torch.float32, **kw)     # the setting was one-shot: no LSE asked for here
    finally:
        _abi.spfaSetMaxSplit(16)
    assert float(parity.rel_err(H.f32(out), want).max()) <= TOL
    assert float(parity.rel_err(H.f32(again), want).max()) <= TOL
    _lse_close(lse.cpu().numpy(), wlse)


def test_lse_equals_torch_logsumexp(H):
    """Independent of the oracle: gather the listed rows with torch and take logsumexp of the scaled scores."""
    Hq, D, bs = 4, 128, 128
    rng, q, k, v, table, pos, act = _case(31, "f16", Hq, Hq, "kv", D, bs)
    act = [pos.shape[2] * bs] * q.shape[0]
    lse = torch.empty((q.shape[0], Hq), dtype=torch.float32, device=H.DEV)
    H.call_attention(H.dev(q), H.dev(k), H.dev(v), H.dev(table), H.dev(pos), act, Hq, Hq, 0.25, bs, softmax_lse=lse)
    got = lse.cpu().numpy()
    kt = torch.from_numpy(k.astype(np.float32)).reshape(k.shape[0], bs, Hq, D)
    for b in range(q.shape[0]):
        for h in range(Hq):
            pages = [int(x) for x in pos[b, h] if x != SENT]
            if not pages:
                assert got[b, h] == np.float32(-3.40e38)
                continue
            rows = torch.cat([kt[int(table[b, pg]), :, h] for pg in pages])
            s = (rows @ torch.from_numpy(q[b, h].astype(np.float32))) * 0.25
            assert abs(float(torch.logsumexp(s.double(), 0)) - float(got[b, h])) <= 1e-4


@pytest.mark.parametrize("dtype,Hq,Hkv,rows,D,bs,layout,n_pre,S", [
    ("f16", 8, 8, "kv", 128, 128, "BSH", 300, 300),      # partly filled last run
    ("f16", 8, 2, "kv", 128, 128, "BSH", 256, 384),      # actual length < tensor length; matrix-core streamer
    ("f16", 8, 2, "q", 128, 64, "BNSD", 70, 100),
    ("bf16", 16, 2, "kv", 64, 32, "BNSD", 33, 64),
    ("i8", 8, 2, "kv", 128, 128, "BSH", 129, 129),
    ("i8", 4, 4, "kv", 128, 32, "BNSD", 31, 40),
])
@pytest.mark.parametrize("max_split", [16, 1])
def test_shared_prefix_vs_oracle(H, dtype, Hq, Hkv, rows, D, bs, layout, n_pre, S, max_split):
    from custom_ops import _abi
    rng, q, k, v, table, pos, act = _case(800 + Hq + D, dtype, Hq, Hkv, rows, D, bs, layout=layout)
    pos[0, 0, :] = SENT                                     # this row attends the prefix only
    kp, vp = _prefix(rng, dtype, S, Hkv, D, layout)
    kw, okw = _quant(H, dtype, Hkv, D)
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    scale = D ** -0.5
    want, wlse = ref.sparse_paged_attention(q, k, v, table, pos, act, Hq, Hkv, scale, bs, layout, key_shared_prefix=kp,
                                            value_shared_prefix=vp, shared_prefix_len=n_pre, return_lse=True, **okw)
    plain = ref.sparse_paged_attention(q, k, v, table, pos, act, Hq, Hkv, scale, bs, layout, **okw)
    assert float(np.abs(want - plain).max()) > 1e-2 * float(np.abs(plain).max())   # the prefix matters
    cv = (lambda a: H.dev(a)) if dtype == "i8" else (lambda a: H.dev(a, tdt))
    lse = torch.empty((q.shape[0], Hq), dtype=torch.float32, device=H.DEV)
    _abi.check(_abi.spfaSetMaxSplit(max_split), "spfaSetMaxSplit")
    try:
        out = H.call_attention(H.dev(q, tdt), cv(k), cv(v), H.dev(table), H.dev(pos), torch.tensor(act), Hq, Hkv, scale,
                               bs, out_dtype=torch.float32, softmax_lse=lse, key_shared_prefix=cv(kp),
                               value_shared_prefix=cv(vp), actual_shared_prefix_len=n_pre, **kw)
    finally:
        _abi.spfaSetMaxSplit(16)
    err = float(parity.rel_err(H.f32(out), want).max())
    assert err <= TOL, f"max rel err {err:.3e}"
    _lse_close(lse.cpu().numpy(), wlse)


def test_prefix_equals_leading_pages(H):
    """Independent of the oracle: a prefix holding the rows of two cache pages gives what the plain op gives on lists that
    start with those two pages."""
    Hq, Hkv, D, bs, B = 8, 2, 128, 128, 3
    rng, q, k, v, table, pos, act = _case(41, "f16", Hq, Hkv, "kv", D, bs, B=B)
    table[:, 0], table[:, 1] = 5, 9                          # logical pages 0 and 1 of every sequence
    P = pos.shape[2]
    longer = np.full((B, Hkv, P + 2), SENT, np.int32)
    for b in range(B):
        for r in range(Hkv):
            rest = [x for x in pos[b, r] if x != SENT and x > 1]
            pos[b, r, :] = SENT
            pos[b, r, :len(rest)] = rest
            longer[b, r, :2 + len(rest)] = [0, 1] + rest
    kc = k.reshape(k.shape[0], bs, Hkv * D)
    vc = v.reshape(v.shape[0], bs, Hkv * D)
    kp = np.concatenate([kc[5], kc[9]])[None]
    vp = np.concatenate([vc[5], vc[9]])[None]
    n_rest = [(P + 2) * bs] * B
    want = H.call_attention(H.dev(q), H.dev(k), H.dev(v), H.dev(table), H.dev(longer), n_rest, Hq, Hkv, 0.1, bs,
                            out_dtype=torch.float32)
    got = H.call_attention(H.dev(q), H.dev(k), H.dev(v), H.dev(table), H.dev(pos), [P * bs] * B, Hq, Hkv, 0.1, bs,
                           out_dtype=torch.float32, key_shared_prefix=H.dev(kp), value_shared_prefix=H.dev(vp))
    assert float(parity.rel_err(H.f32(got), H.f32(want)).max()) <= 1e-5


@pytest.mark.parametrize("n_pre", [128, 200])
def test_prefix_shifts_bias_and_mask_columns(H, n_pre):
    Hq, Hkv, D, bs = 8, 2, 128, 128
    rng, q, k, v, table, pos, act = _case(43, "f16", Hq, Hkv, "kv", D, bs)
    B, S = q.shape[0], table.shape[1] * bs + n_pre
    kp, vp = _prefix(rng, "f16", n_pre, Hkv, D, "BSH")
    pse = (rng.standard_normal((B, Hq, 1, S)) * 2.0).astype(np.float16)
    mask = rng.random((B, S)) < 0.3
    want = ref.sparse_paged_attention(q, k, v, table, pos, act, Hq, Hkv, D ** -0.5, bs, "BSH", pse_shift=pse,
                                      atten_mask=mask, key_shared_prefix=kp, value_shared_prefix=vp)
    out = H.call_attention(H.dev(q), H.dev(k), H.dev(v), H.dev(table), H.dev(pos), torch.tensor(act), Hq, Hkv, D ** -0.5,
                           bs, out_dtype=torch.float32, pse=H.dev(pse), mask=H.dev(mask), key_shared_prefix=H.dev(kp),
                           value_shared_prefix=H.dev(vp))
    assert float(parity.rel_err(H.f32(out), want).max()) <= TOL


@pytest.mark.parametrize("Hq,Hkv,kvdt", [(8, 2, "f16"), (4, 4, "bf16"), (8, 8, "i8")])
def test_fusion_with_prefix_and_lse(H, Hq, Hkv, kvdt):
    D, C, topk, bs = 128, 256, 8, 128
    seq_lens = [4000, 32 * 128, 129, 77]
    B, kv_page_len, max_page = len(seq_lens), 96, 80
    q, l1 = datagen.cent_case(71 + Hq, B, Hq, Hkv, D, C, grid=True)
    ids, table, seq = datagen.page_case(71 + Hq, Hkv, C, kv_page_len, B + 1, max_page, seq_lens)
    k, v = datagen.kv_case(71, kv_page_len, bs, Hkv, D, "BSH", dist="randn", dtype="i8" if kvdt == "i8" else "f16")
    tdt = torch.bfloat16 if kvdt == "bf16" else torch.float16
    if kvdt == "bf16":
        k, v = (ref.round_to_bf16(x.astype(np.float32)) for x in (k, v))
    rng = datagen.rng_for(6)
    kp, vp = _prefix(rng, kvdt, 200, Hkv, D, "BSH")
    kw, okw = _quant(H, kvdt, Hkv, D)
    scale = D ** -0.5
    (want, wlse), wpos, wlen, wmax = ref.sparse_paged_fusion_attention(
        q, k, v, table, l1, ids, seq, Hq, Hkv, scale, bs, topk, 256, key_shared_prefix=kp, value_shared_prefix=vp,
        return_lse=True, **okw)
    cv = (lambda a: H.dev(a)) if kvdt == "i8" else (lambda a: H.dev(a, tdt))
    lse = torch.empty((B, Hq), dtype=torch.float32, device=H.DEV)
    out, pos, plen, pmax = H.call_fusion(H.dev(q, tdt), cv(k), cv(v), H.dev(table), H.dev(l1, tdt), H.dev(ids),
                                         H.dev(seq), Hq, Hkv, scale, bs, topk, out_dtype=torch.float32, softmax_lse=lse,
                                         key_shared_prefix=cv(kp), value_shared_prefix=cv(vp), **kw)
    # the selection outputs do not know about the prefix
    assert np.array_equal(pos.cpu().numpy(), wpos) and np.array_equal(plen.cpu().numpy(), wlen)
    assert np.array_equal(pmax.cpu().numpy(), wmax)
    err = float(parity.rel_err(H.f32(out), want).max())
    assert err <= TOL, f"max rel err {err:.3e}"
    _lse_close(lse.cpu().numpy(), wlse)


def test_prefix_in_a_large_launch_with_shared_last_items(H):
    """More items than resident workgroups: the prefix runs travel with the page list through the sharing of last items."""
    Hq, D, bs, B, P = 32, 128, 16, 40, 256
    rng, q, k, v, table, pos, act = _case(83, "f16", Hq, Hq, "kv", D, bs, B=B, n_blocks=64, max_page=24, P=P)
    act = [P * bs] * B
    kp, vp = _prefix(rng, "f16", 100, Hq, D, "BSH")
    want, wlse = ref.sparse_paged_attention(q, k, v, table, pos, act, Hq, Hq, D ** -0.5, bs, "BSH", key_shared_prefix=kp,
                                            value_shared_prefix=vp, return_lse=True)
    lse = torch.empty((B, Hq), dtype=torch.float32, device=H.DEV)
    out = H.call_attention(H.dev(q), H.dev(k), H.dev(v), H.dev(table), H.dev(pos), torch.tensor(act), Hq, Hq, D ** -0.5,
                           bs, out_dtype=torch.float32, softmax_lse=lse, key_shared_prefix=H.dev(kp),
                           value_shared_prefix=H.dev(vp))
    assert float(parity.rel_err(H.f32(out), want).max()) <= TOL
    _lse_close(lse.cpu().numpy(), wlse)


def test_extras_errors(H):
    from custom_ops._abi import AclError
    B, Hq, D, bs = 2, 4, 128, 128
    q = H.dev(np.zeros((B, Hq, D), np.float16))
    kv = H.dev(np.zeros((4, bs, Hq * D), np.float16))
    table = H.dev(np.zeros((B, 4), np.int32))
    pos = H.dev(np.zeros((B, Hq, 4), np.int32))
    args = (q, kv, kv, table, pos, [128], Hq, Hq, 1.0, bs)
    with pytest.raises(AclError, match="softmax LSE output has"):
        H.call_attention(*args, softmax_lse=torch.empty(B * Hq - 1, dtype=torch.float32, device=H.DEV))
    with pytest.raises(AclError, match="float32"):
        H.call_attention(*args, softmax_lse=torch.empty(B * Hq, dtype=torch.float16, device=H.DEV))
    pre = H.dev(np.zeros((1, 64, Hq * D), np.float16))
    with pytest.raises(AclError, match="come together"):
        H.call_attention(*args, key_shared_prefix=pre)
    with pytest.raises(AclError, match="actual length"):
        H.call_attention(*args, key_shared_prefix=pre, value_shared_prefix=pre, actual_shared_prefix_len=65)
    with pytest.raises(AclError, match="token-level"):
        H.call_attention(*args, key_shared_prefix=pre, value_shared_prefix=pre, token_level=True)
    # a failed call leaves nothing armed: the next plain call neither reads a prefix nor writes an LSE
    out = H.call_attention(*args)
    assert torch.isfinite(out.float()).all()


def _per_token_params(rng, B, S, with_offset):
    sc = (rng.random((2, B, S)) * 0.02 + 0.002).astype(np.float32)
    of = (rng.standard_normal((2, B, S)) * 4.0).astype(np.float32) if with_offset else None
    return sc, of


@pytest.mark.parametrize("Hq,Hkv,rows,D,bs,layout,with_offset,n_pre", [
    (8, 8, "kv", 128, 128, "BSH", True, 0),
    (8, 2, "kv", 128, 128, "BSH", True, 0),          # a group: every head becomes a pass of its own
    (8, 2, "q", 128, 32, "BNSD", False, 0),
    (4, 4, "kv", 64, 64, "BSH", True, 0),
    (8, 2, "kv", 128, 128, "BSH", True, 200),        # the prefix tokens come first in the parameter columns
])
@pytest.mark.parametrize("max_split", [16, 1])
def test_per_token_antiquant_vs_oracle(H, Hq, Hkv, rows, D, bs, layout, with_offset, n_pre, max_split):
    """Per-token parameters (float32 [2, B, S] over the gathered columns).  Parity unpinned; the per-channel form of the
    same numbers is an independent check below."""
    from custom_ops import _abi
    rng, q, k, v, table, pos, act = _case(900 + Hq + D, "i8", Hq, Hkv, rows, D, bs, layout=layout)
    B = q.shape[0]
    S = table.shape[1] * bs + n_pre + 5
    sc, of = _per_token_params(rng, B, S, with_offset)
    pre_kw, pre_okw = {}, {}
    if n_pre:
        kp, vp = _prefix(rng, "i8", n_pre, Hkv, D, layout)
        pre_kw = dict(key_shared_prefix=H.dev(kp), value_shared_prefix=H.dev(vp))
        pre_okw = dict(key_shared_prefix=kp, value_shared_prefix=vp)
    scale = D ** -0.5
    want, wlse = ref.sparse_paged_attention(q, k, v, table, pos, act, Hq, Hkv, scale, bs, layout, antiquant_scale=sc,
                                            antiquant_offset=of, antiquant_mode=1, return_lse=True, **pre_okw)
    lse = torch.empty((B, Hq), dtype=torch.float32, device=H.DEV)
    _abi.check(_abi.spfaSetMaxSplit(max_split), "spfaSetMaxSplit")
    try:
        out = H.call_attention(H.dev(q), H.dev(k), H.dev(v), H.dev(table), H.dev(pos), torch.tensor(act), Hq, Hkv, scale,
                               bs, out_dtype=torch.float32, antiq_scale=H.dev(sc),
                               antiq_offset=None if of is None else H.dev(of), antiquant_mode=1, softmax_lse=lse, **pre_kw)
    finally:
        _abi.spfaSetMaxSplit(16)
    err = float(parity.rel_err(H.f32(out), want).max())
    assert err <= TOL, f"max rel err {err:.3e}"
    _lse_close(lse.cpu().numpy(), wlse)


def test_per_token_equals_per_tensor_when_constant(H):
    """Independent of the oracle: per-token parameters that are the same for every token are per-tensor parameters."""
    Hq, Hkv, D, bs = 8, 2, 128, 128
    rng, q, k, v, table, pos, act = _case(47, "i8", Hq, Hkv, "kv", D, bs)
    B, S = q.shape[0], table.shape[1] * bs
    a_s = np.array([0.0125, 0.03125], np.float16)          # exact in f16 and f32
    a_o = np.array([3.0, -1.5], np.float16)
    sc = np.broadcast_to(a_s.astype(np.float32).reshape(2, 1, 1), (2, B, S)).copy()
    of = np.broadcast_to(a_o.astype(np.float32).reshape(2, 1, 1), (2, B, S)).copy()
    args = (H.dev(q), H.dev(k), H.dev(v), H.dev(table), H.dev(pos), torch.tensor(act), Hq, Hkv, D ** -0.5, bs)
    tensor = H.call_attention(*args, out_dtype=torch.float32, antiq_scale=H.dev(a_s), antiq_offset=H.dev(a_o))
    token = H.call_attention(*args, out_dtype=torch.float32, antiq_scale=H.dev(sc), antiq_offset=H.dev(of),
                             antiquant_mode=1)
    # two different kernels (the group pass on the matrix cores against one pass per head): fp32 rounding apart
    assert float(parity.rel_err(H.f32(token), H.f32(tensor)).max()) <= 2e-4


def test_fusion_with_per_token_antiquant(H):
    Hq, Hkv, D, C, topk, bs = 8, 2, 128, 256, 8, 128
    seq_lens = [4000, 32 * 128, 129, 77]
    B, kv_page_len, max_page = len(seq_lens), 96, 80
    q, l1 = datagen.cent_case(91, B, Hq, Hkv, D, C, grid=True)
    ids, table, seq = datagen.page_case(91, Hkv, C, kv_page_len, B + 1, max_page, seq_lens)
    k, v = datagen.kv_case(91, kv_page_len, bs, Hkv, D, "BSH", dist="randn", dtype="i8")
    rng = datagen.rng_for(9)
    sc, of = _per_token_params(rng, B, max_page * bs, True)
    scale = D ** -0.5
    want, wpos, wlen, wmax = ref.sparse_paged_fusion_attention(q, k, v, table, l1, ids, seq, Hq, Hkv, scale, bs, topk, 256,
                                                               antiquant_scale=sc, antiquant_offset=of, antiquant_mode=1)
    out, pos, plen, pmax = H.call_fusion(H.dev(q), H.dev(k), H.dev(v), H.dev(table), H.dev(l1), H.dev(ids), H.dev(seq),
                                         Hq, Hkv, scale, bs, topk, out_dtype=torch.float32, antiq_scale=H.dev(sc),
                                         antiq_offset=H.dev(of), antiquant_mode=1)
    assert np.array_equal(pos.cpu().numpy(), wpos) and np.array_equal(plen.cpu().numpy(), wlen)
    assert np.array_equal(pmax.cpu().numpy(), wmax)
    err = float(parity.rel_err(H.f32(out), want).max())
    assert err <= TOL, f"max rel err {err:.3e}"


def test_per_token_antiquant_errors(H):
    from custom_ops._abi import AclError
    B, Hq, D, bs = 2, 4, 128, 128
    q = H.dev(np.zeros((B, Hq, D), np.float16))
    kv8 = H.dev(np.zeros((4, bs, Hq * D), np.int8))
    kv16 = H.dev(np.zeros((4, bs, Hq * D), np.float16))
    table = H.dev(np.zeros((B, 4), np.int32))
    pos = H.dev(np.zeros((B, Hq, 4), np.int32))
    good = H.dev(np.ones((2, B, 4 * bs), np.float32))
    with pytest.raises(AclError, match="need an int8 KV cache"):
        H.call_attention(q, kv16, kv16, table, pos, [128], Hq, Hq, 1.0, bs, antiquant_mode=1)
    with pytest.raises(AclError, match="float32 \\[2, B, S\\]"):
        H.call_attention(q, kv8, kv8, table, pos, [128], Hq, Hq, 1.0, bs, antiquant_mode=1,
                         antiq_scale=H.dev(np.ones((2, Hq, D), np.float16)))
    with pytest.raises(AclError, match="less than sMax"):
        H.call_attention(q, kv8, kv8, table, pos, [128], Hq, Hq, 1.0, bs, antiquant_mode=1,
                         antiq_scale=H.dev(np.ones((2, B, 4 * bs - 1), np.float32)))
    with pytest.raises(AclError, match="antiquant mode"):
        H.call_attention(q, kv8, kv8, table, pos, [128], Hq, Hq, 1.0, bs, antiquant_mode=4, antiq_scale=good)
    with pytest.raises(AclError, match="\\[2, Hkv = 4\\]"):
        H.call_attention(q, kv8, kv8, table, pos, [128], Hq, Hq, 1.0, bs, antiquant_mode=2, antiq_scale=good)
    with pytest.raises(AclError, match="float32 \\[2, B, Hkv, S\\]"):
        H.call_attention(q, kv8, kv8, table, pos, [128], Hq, Hq, 1.0, bs, antiquant_mode=3, antiq_scale=good)
    with pytest.raises(AclError, match="holds 3 heads"):
        H.call_attention(q, kv8, kv8, table, pos, [128], Hq, Hq, 1.0, bs, antiquant_mode=3,
                         antiq_scale=H.dev(np.ones((2, B, 3, 4 * bs), np.float32)))
    # the mode was one-shot: a per-channel call right after works with per-channel shapes
    out = H.call_attention(q, kv8, kv8, table, pos, [128], Hq, Hq, 1.0, bs,
                           antiq_scale=H.dev(np.ones((2, Hq, D), np.float16)))
    assert torch.isfinite(out.float()).all()


def test_per_token_antiquant_in_a_large_launch(H):
    """More items than resident workgroups: per-token parameters follow the list slot through shared last items and the
    rolling item pipeline (column = prefix tokens + slot * block_size + token)."""
    Hq, D, bs, B, P = 32, 128, 32, 40, 128
    rng, q, k, v, table, pos, act = _case(87, "i8", Hq, Hq, "kv", D, bs, B=B, n_blocks=64, max_page=24, P=P)
    act = [P * bs] * B
    kp, vp = _prefix(rng, "i8", 50, Hq, D, "BSH")
    S = table.shape[1] * bs + 50
    sc, of = _per_token_params(rng, B, S, True)
    want, wlse = ref.sparse_paged_attention(q, k, v, table, pos, act, Hq, Hq, D ** -0.5, bs, "BSH", antiquant_scale=sc,
                                            antiquant_offset=of, antiquant_mode=1, key_shared_prefix=kp,
                                            value_shared_prefix=vp, return_lse=True)
    lse = torch.empty((B, Hq), dtype=torch.float32, device=H.DEV)
    out = H.call_attention(H.dev(q), H.dev(k), H.dev(v), H.dev(table), H.dev(pos), torch.tensor(act), Hq, Hq, D ** -0.5,
                           bs, out_dtype=torch.float32, antiq_scale=H.dev(sc), antiq_offset=H.dev(of), antiquant_mode=1,
                           softmax_lse=lse, key_shared_prefix=H.dev(kp), value_shared_prefix=H.dev(vp))
    assert float(parity.rel_err(H.f32(out), want).max()) <= TOL
    _lse_close(lse.cpu().numpy(), wlse)


@pytest.mark.parametrize("Hq,Hkv,rows,D,bs,layout,with_offset,fused", [
    (8, 8, "kv", 128, 128, "BSH", True, False),
    (8, 2, "kv", 128, 128, "BSH", True, False),          # a group pass on the matrix cores
    (8, 2, "q", 128, 32, "BNSD", False, False),
    (32, 4, "kv", 64, 64, "BSH", True, False),
    (8, 2, "kv", 128, 128, "BSH", True, True),
])
def test_per_head_antiquant_vs_oracle(H, Hq, Hkv, rows, D, bs, layout, with_offset, fused):
    """Antiquant mode 2: one key and one value parameter per kv head, [2, Hkv] in the query dtype
    (sparse_paged_fusion_attention_tiling_check.cc:488-497).  Parity unpinned (the oracle restates the kernel source);
    the per-channel form of the same numbers is an independent check."""
    rng, q, k, v, table, pos, act = _case(1200 + Hq + D, "i8", Hq, Hkv, rows, D, bs, layout=layout)
    B = q.shape[0]
    sc = (rng.random((2, Hkv)) * 0.02 + 0.002).astype(np.float16)
    of = (rng.standard_normal((2, Hkv)) * 4.0).astype(np.float16) if with_offset else None
    scale = D ** -0.5
    chan_s = np.broadcast_to(sc[:, :, None], (2, Hkv, D)).copy()
    chan_o = None if of is None else np.broadcast_to(of[:, :, None], (2, Hkv, D)).copy()
    if fused:
        C, topk = 256, 8
        _, l1 = datagen.cent_case(3, B, Hq, Hkv, D, C, grid=True)
        ids = rng.integers(0, C, size=(Hkv, k.shape[0])).astype(np.int32)
        seq = np.array(act, np.int32)
        want, wpos, wlen, wmax = ref.sparse_paged_fusion_attention(q, k, v, table, l1, ids, seq, Hq, Hkv, scale, bs, topk, 256,
                                                                   kv_layout=layout, antiquant_scale=sc, antiquant_offset=of,
                                                                   antiquant_mode=2)
        out, gpos, _, _ = H.call_fusion(H.dev(q), H.dev(k), H.dev(v), H.dev(table), H.dev(l1), H.dev(ids), H.dev(seq), Hq, Hkv,
                                        scale, bs, topk, out_dtype=torch.float32, antiq_scale=H.dev(sc),
                                        antiq_offset=None if of is None else H.dev(of), antiquant_mode=2)
        assert np.array_equal(gpos.cpu().numpy(), wpos)
        assert float(parity.rel_err(H.f32(out), want).max()) <= TOL
        return
    want, wlse = ref.sparse_paged_attention(q, k, v, table, pos, act, Hq, Hkv, scale, bs, layout, antiquant_scale=sc,
                                            antiquant_offset=of, antiquant_mode=2, return_lse=True)
    lse = torch.empty((B, Hq), dtype=torch.float32, device=H.DEV)
    args = (H.dev(q), H.dev(k), H.dev(v), H.dev(table), H.dev(pos), torch.tensor(act), Hq, Hkv, scale, bs)
    out = H.call_attention(*args, out_dtype=torch.float32, antiq_scale=H.dev(sc),
                           antiq_offset=None if of is None else H.dev(of), antiquant_mode=2, softmax_lse=lse)
    assert float(parity.rel_err(H.f32(out), want).max()) <= TOL
    _lse_close(lse.cpu().numpy(), wlse)
    # the same numbers spelled out per channel: the same kernel, the same arithmetic
    chan = H.call_attention(*args, out_dtype=torch.float32, antiq_scale=H.dev(chan_s),
                            antiq_offset=None if chan_o is None else H.dev(chan_o))
    assert float((chan - out).abs().max()) <= 2e-6 * float(chan.abs().max())


@pytest.mark.parametrize("Hq,Hkv,rows,D,bs,layout,with_offset,n_pre", [
    (8, 8, "kv", 128, 128, "BSH", True, 0),
    (8, 2, "kv", 128, 128, "BSH", True, 0),
    (8, 2, "q", 128, 32, "BNSD", False, 0),
    (8, 2, "kv", 128, 128, "BSH", True, 200),
])
def test_per_token_head_antiquant_vs_oracle(H, Hq, Hkv, rows, D, bs, layout, with_offset, n_pre):
    """Antiquant mode 3: per token and kv head, float32 [2, B, Hkv, S] (…_tiling_check.cc:466-487, kernel offsets
    sparse_paged_fusion_attention_split_Bbn2s2_Us2.h:2203-2206).  Parity unpinned; parameters that are the same for every
    head must give what mode 1 gives."""
    rng, q, k, v, table, pos, act = _case(1300 + Hq + D, "i8", Hq, Hkv, rows, D, bs, layout=layout)
    B = q.shape[0]
    S = table.shape[1] * bs + n_pre + 5
    sc = (rng.random((2, B, Hkv, S)) * 0.02 + 0.002).astype(np.float32)
    of = (rng.standard_normal((2, B, Hkv, S)) * 4.0).astype(np.float32) if with_offset else None
    pre_kw, pre_okw = {}, {}
    if n_pre:
        kp, vp = _prefix(rng, "i8", n_pre, Hkv, D, layout)
        pre_kw = dict(key_shared_prefix=H.dev(kp), value_shared_prefix=H.dev(vp))
        pre_okw = dict(key_shared_prefix=kp, value_shared_prefix=vp)
    scale = D ** -0.5
    want, wlse = ref.sparse_paged_attention(q, k, v, table, pos, act, Hq, Hkv, scale, bs, layout, antiquant_scale=sc,
                                            antiquant_offset=of, antiquant_mode=3, return_lse=True, **pre_okw)
    lse = torch.empty((B, Hq), dtype=torch.float32, device=H.DEV)
    args = (H.dev(q), H.dev(k), H.dev(v), H.dev(table), H.dev(pos), torch.tensor(act), Hq, Hkv, scale, bs)
    out = H.call_attention(*args, out_dtype=torch.float32, antiq_scale=H.dev(sc),
                           antiq_offset=None if of is None else H.dev(of), antiquant_mode=3, softmax_lse=lse, **pre_kw)
    assert float(parity.rel_err(H.f32(out), want).max()) <= TOL
    _lse_close(lse.cpu().numpy(), wlse)
    # head-independent parameters: mode 3 == mode 1, bit for bit (the same kernel reads the same numbers)
    sc1 = sc[:, :, 0].copy()
    of1 = None if of is None else of[:, :, 0].copy()
    sc3 = np.broadcast_to(sc1[:, :, None, :], sc.shape).copy()
    of3 = None if of1 is None else np.broadcast_to(of1[:, :, None, :], sc.shape).copy()
    a = H.call_attention(*args, out_dtype=torch.float32, antiq_scale=H.dev(sc1),
                         antiq_offset=None if of1 is None else H.dev(of1), antiquant_mode=1, **pre_kw)
    b = H.call_attention(*args, out_dtype=torch.float32, antiq_scale=H.dev(sc3),
                         antiq_offset=None if of3 is None else H.dev(of3), antiquant_mode=3, **pre_kw)
    assert torch.equal(a, b)
