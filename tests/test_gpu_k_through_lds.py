"""Group passes on the matrix-core streamer fetch K either in whole rows through a wave-private LDS image (the default
for head dims up to 128; the image lies in the bytes of the tile's V image) or with strided loads straight into operand
order.  The two forms feed the same operands to the
same matrix instructions, so with a reproducible summation order (sharing of last items off) the outputs must be equal
bit for bit; each is also held to the oracle.  Covers the fused op and token-level lists shared by a group, f16 / bf16 /
int8 KV, both layouts, head dims 64 / 128 / 256, ragged lengths, 16-head passes."""
import numpy as np
import pytest
import torch

import datagen
import parity
from oracle import cport, ref

pytestmark = pytest.mark.gpu
SENT = 0x7FFFFFFF


def _fusion_case(seed, B, Hq, Hkv, D, C, bs, max_page, kvdt, layout):
    import gpu_helpers as H
    rng = datagen.rng_for(seed)
    seq_lens = [int(x) for x in rng.integers(1, max_page * bs + 1, size=B)]
    seq_lens[0] = max_page * bs
    q, l1 = datagen.cent_case(seed, B, Hq, Hkv, D, C, grid=True)
    kv_page_len = 200
    ids, table, seq = datagen.page_case(seed, Hkv, C, kv_page_len, B, max_page, seq_lens)
    k, v = datagen.kv_case(seed, kv_page_len, bs, Hkv, D, layout, dist="randn", dtype="i8" if kvdt == "i8" else "f16")
    host = dict(q=q, l1=l1, ids=ids, table=table, seq=seq, k=k, v=v)
    if kvdt == "i8":
        host["a_scale"], host["a_off"] = datagen.antiquant_case(seed, Hkv, D)
    return host, {n: H.dev(a) for n, a in host.items()}


@pytest.mark.parametrize("B,Hq,Hkv,D,C,bs,max_page,kvdt,layout", [
    (40, 16, 4, 128, 256, 128, 40, "f16", "BSH"),      # 160 items of 4 heads
    (70, 32, 4, 128, 256, 128, 30, "f16", "BNSD"),     # 280 items of 8 heads: more than the resident workgroups
    (30, 16, 2, 64, 256, 64, 50, "f16", "BSH"),
    (36, 16, 4, 128, 384, 128, 40, "i8", "BSH"),
    (24, 8, 2, 256, 256, 32, 60, "i8", "BNSD"),        # head dim 256: strided K loads only
    (3, 32, 2, 128, 256, 128, 48, "f16", "BSH"),       # 6 items of 16 heads, each dealt to several workgroups
    (5, 16, 2, 64, 256, 16, 200, "f16", "BSH"),        # 8 heads, 16-token pages
])
def test_fused_group_passes_equal_with_k_through_lds_and_without(B, Hq, Hkv, D, C, bs, max_page, kvdt, layout):
    import custom_ops
    import gpu_helpers as H
    from custom_ops import _abi
    topk = 8
    host, dev = _fusion_case(8100 + B, B, Hq, Hkv, D, C, bs, max_page, kvdt, layout)
    kw = dict(kv_dt="i8", antiq_scale=host["a_scale"], antiq_offset=host["a_off"]) if kvdt == "i8" else {}
    want, wpos, wlen, wmax = cport.fusion(host["q"], host["k"], host["v"], host["table"], host["l1"], host["ids"], host["seq"],
                                          Hq, Hkv, D ** -0.5, bs, topk, 256, kv_layout=layout, **kw)
    outs = {}
    _abi.check(_abi.spfaSetTailSharing(0), "spfaSetTailSharing")
    try:
        lds_without = None
        for setting in (0, 1):
            _abi.check(_abi.spfaSetKThroughLds(setting), "spfaSetKThroughLds")
            pos = torch.full((B, Hq, 256), -7, dtype=torch.int32, device=H.DEV)
            plen = torch.full((B, Hq, 8), -7, dtype=torch.int32, device=H.DEV)
            pmax = torch.full((B, 8), -7, dtype=torch.int64, device=H.DEV)
            plan = custom_ops.PreparedFusion(dev["q"], dev["k"], dev["v"], dev["table"], dev["l1"], dev["ids"], dev["seq"],
                                             pos, plen, pmax, Hq, D ** -0.5, Hkv, bs, k=topk, out_dtype=torch.float32,
                                             antiquant_scale=dev.get("a_scale"), antiquant_offset=dev.get("a_off"))
            d = plan.describe()
            assert d["form"] == "mfma", d
            if setting == 0:
                assert d["k_through_lds"] == 0, d
                lds_without = d["lds_bytes"]
            else:      # head dims up to 128; the K image shares the V image's bytes, so the plan needs no more LDS
                assert d["k_through_lds"] == int(D <= 128) and d["lds_bytes"] == lds_without, (d, lds_without)
            out = plan.run()
            torch.cuda.synchronize()
            assert np.array_equal(pos.cpu().numpy(), wpos) and np.array_equal(plen.cpu().numpy(), wlen)
            assert float(parity.rel_err(H.f32(out), want).max()) <= 1e-3
            outs[setting] = out.clone()
    finally:
        _abi.spfaSetKThroughLds(1)
        _abi.spfaSetTailSharing(1)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("Hq,Hkv,D,bs,kvdt,layout,max_tokens", [
    (16, 4, 128, 128, "f16", "BSH", 900),
    (32, 4, 64, 16, "f16", "BNSD", 500),
    (16, 4, 128, 128, "i8", "BSH", 5000),     # two windows
])
def test_shared_token_lists_equal_with_k_through_lds_and_without(Hq, Hkv, D, bs, kvdt, layout, max_tokens):
    import custom_ops
    import gpu_helpers as H
    from custom_ops import _abi
    B, max_page = 3, 48
    rng = datagen.rng_for(max_tokens + Hq)
    n_blocks = 120
    if kvdt == "i8":
        k, v = datagen.kv_case(5, n_blocks, bs, Hkv, D, layout, dtype="i8")
        sc, off = datagen.antiquant_case(5, Hkv, D)
    else:
        k, v = datagen.kv_case(5, n_blocks, bs, Hkv, D, layout, dist="randn")
        sc = off = None
    table = rng.integers(0, n_blocks, size=(B, max_page)).astype(np.int32)
    tp = np.full((B, Hkv, max_tokens), SENT, dtype=np.int32)
    limit = min(max_tokens, max_page * bs)
    for b in range(B):
        for r in range(Hkv):
            n = limit if (b, r) == (0, 0) else int(rng.integers(1, limit + 1))
            tp[b, r, :n] = np.sort(rng.choice(max_page * bs, size=n, replace=False))
    q = datagen.ref_rand_f16(rng, (B, Hq, D)).astype(np.float32)
    want = ref.sparse_paged_attention(q, k, v, table, tp, None, Hq, Hkv, D ** -0.5, bs, layout, sc, off, token_level=True)
    e = H.empty()
    dsc, doff = (H.dev(sc.astype(np.float16)), H.dev(off.astype(np.float16))) if sc is not None else (e, e)
    args = (H.dev(q.astype(np.float16)), H.dev(k), H.dev(v))
    outs = {}
    try:
        for setting in (1, 0):
            _abi.check(_abi.spfaSetKThroughLds(setting), "spfaSetKThroughLds")
            out = custom_ops.sparse_paged_attention(*args, e, e, None, e, e, e, e, e, dsc, doff, H.dev(table),
                                                    H.empty(torch.int64), H.dev(tp), Hq, D ** -0.5, "BNSD", Hkv, bs, 1,
                                                    out_dtype=torch.float32, token_level=True)
            torch.cuda.synchronize()
            assert float(parity.rel_err(H.f32(out), want).max()) <= 1e-3
            outs[setting] = out.clone()
    finally:
        _abi.spfaSetKThroughLds(1)
    # token lists are streamed without the sharing of last items, but pages are claimed one by one: the partial sums of
    # a wave depend on which pages it got
    assert float((outs[0] - outs[1]).abs().max()) <= 2e-6 * float(outs[1].abs().max())
