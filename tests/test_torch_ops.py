"""`torch.ops.myops.*`: the dispatcher-level surface (extension/csrc/registration.cpp:43-57) exists, refuses CPU
tensors (there is no CPU implementation) and, on the GPU, returns what `custom_ops.*` returns."""
import numpy as np
import pytest
import torch

import custom_ops  # noqa: F401  (registers the schemas)
import datagen

OPS = ("compute_cent", "select_position", "cent_select", "sparse_paged_attention", "sparse_paged_fusion_attention")


def test_schemas_follow_the_reference():
    for name in OPS:
        assert hasattr(torch.ops.myops, name)
    s = torch.ops.myops.sparse_paged_fusion_attention.default._schema
    names = [a.name for a in s.arguments]
    assert names[:3] == ["query", "key_list", "value_list"]
    assert names[15:21] == ["l1_cent", "block_ids", "total_seq_len", "block_position", "page_position_length",
                            "max_page_position_length"]
    assert names[21:] == ["num_heads", "scale_value", "input_layout", "num_key_value_heads", "block_size",
                          "inner_precise"]
    assert len(s.returns) == 3
    s = torch.ops.myops.sparse_paged_attention.default._schema
    assert [a.name for a in s.arguments][13:16] == ["blocktable", "kv_padding_size", "blockposition"]


def test_cpu_tensors_are_refused():
    q = torch.zeros((2, 8, 64), dtype=torch.float16)
    c = torch.zeros((2, 256, 64), dtype=torch.float16)
    with pytest.raises(NotImplementedError):
        torch.ops.myops.compute_cent(q, c)
    with pytest.raises(RuntimeError):
        custom_ops.compute_cent(q, c)


@pytest.mark.gpu
def test_dispatcher_ops_equal_python_ops():
    dev = "cuda:0"
    B, Hq, Hkv, D, C, bs, pages = 3, 8, 2, 128, 256, 128, 24
    q, l1 = datagen.cent_case(11, B, Hq, Hkv, D, C)
    n_blocks = B * pages
    block_ids, table, seq = datagen.page_case(11, Hkv, C, n_blocks, B, pages, [pages * bs, 5 * bs + 3, 1],
                                              aliasing=False)
    k_np, v_np = datagen.kv_case(11, n_blocks, bs, Hkv, D, "BSH")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    q, l1, block_ids, table, seq, key, val = map(t, (q, l1, block_ids, table, seq, k_np, v_np))
    e = torch.empty(0, dtype=torch.float16, device=dev)
    ei = torch.empty(0, dtype=torch.int64, device=dev)

    idx = torch.ops.myops.compute_cent(q, l1)
    assert torch.equal(idx, custom_ops.compute_cent(q, l1))
    a = torch.ops.myops.select_position(block_ids, table, seq, idx)
    b = custom_ops.select_position(block_ids, table, seq, idx)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    a = torch.ops.myops.cent_select(q, l1, block_ids, table, seq)
    b = custom_ops.cent_select(q, l1, block_ids, table, seq)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    pos, plen, pmax = b

    act = pmax[:, 0].contiguous()
    o1 = torch.ops.myops.sparse_paged_attention(q, [key], [val], e, e, act, e, e, e, e, e, e, e, table, ei, pos,
                                                Hq, D ** -0.5, "BNSD", Hkv, bs, 1)
    o2 = custom_ops.sparse_paged_attention(q, key, val, e, e, act, e, e, e, e, e, e, e, table, ei, pos,
                                           Hq, D ** -0.5, "BNSD", Hkv, bs, 1)
    assert torch.equal(o1, o2)

    def fused(fn, kv):
        p = torch.empty((B, Hq, 256), dtype=torch.int32, device=dev)
        n = torch.empty((B, Hq, 8), dtype=torch.int32, device=dev)
        m = torch.empty((B, 8), dtype=torch.int64, device=dev)
        out, p2, m2 = fn(q, kv(key), kv(val), e, e, torch.tensor([pages * bs // 8]), e, e, e, e, e, e, e, table, ei,
                         l1, block_ids, seq, p, n, m, Hq, D ** -0.5, "BNSD", Hkv, bs, 1)
        assert p2.data_ptr() == p.data_ptr() and m2.data_ptr() == m.data_ptr()
        return out, p, n, m
    a = fused(torch.ops.myops.sparse_paged_fusion_attention, lambda x: [x])
    b = fused(custom_ops.sparse_paged_fusion_attention, lambda x: x)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert torch.equal(a[1], pos) and torch.equal(a[2], plen) and torch.equal(a[3], pmax)


@pytest.mark.gpu
def test_dispatcher_passes_the_optional_operands():
    """pse_shift / attention_mask / quant_scale2 travel through the registered schema like any other tensor."""
    dev = "cuda:0"
    B, Hq, Hkv, D, bs, pages = 2, 4, 4, 128, 128, 6
    rng = datagen.rng_for(3)
    q = torch.from_numpy(datagen.ref_rand_f16(rng, (B, Hq, D))).to(dev)
    k_np, v_np = datagen.kv_case(3, B * pages, bs, Hkv, D, "BSH", dist="randn")
    key, val = torch.from_numpy(k_np).to(dev), torch.from_numpy(v_np).to(dev)
    table = torch.arange(B * pages, dtype=torch.int32, device=dev).reshape(B, pages)
    pos = torch.arange(pages, dtype=torch.int32, device=dev).repeat(B, Hkv, 1).contiguous()
    e = torch.empty(0, dtype=torch.float16, device=dev)
    ei = torch.empty(0, dtype=torch.int64, device=dev)
    S = pages * bs
    pse = torch.from_numpy((rng.standard_normal((B, Hq, 1, S)) * 2).astype(np.float16)).to(dev)
    mask = torch.from_numpy(rng.random((B, S)) < 0.3).to(dev)
    sc = torch.full((1,), 300.0, dtype=torch.float32, device=dev)
    act = torch.tensor([S] * B)
    args = lambda kv: (q, kv(key), kv(val), pse, mask, act, e, e, e, sc, e, e, e, table, ei, pos, Hq, D ** -0.5, "BNSD",
                       Hkv, bs, 1)
    o1 = torch.ops.myops.sparse_paged_attention(*args(lambda x: [x]))
    o2 = custom_ops.sparse_paged_attention(*args(lambda x: x))
    assert o1.dtype == torch.int8 and torch.equal(o1, o2)
    plain = custom_ops.sparse_paged_attention(q, key, val, e, e, act, e, e, e, sc, e, e, e, table, ei, pos, Hq, D ** -0.5,
                                              "BNSD", Hkv, bs, 1)
    assert not torch.equal(o1, plain)          # bias and mask took effect


def test_meta_implementations_give_the_output_shapes():
    """Shape-only implementations (the torch_npu fork registers the same, op_plugin/python/meta/
    _meta_registrations.py:38-93): the ops can be traced without running a kernel."""
    m = lambda shape, dt=torch.float16: torch.empty(shape, dtype=dt, device="meta")
    B, Hq, Hkv, D, C = 4, 8, 2, 128, 256
    q, l1 = m((B, Hq, D)), m((Hkv, C, D))
    ids, table, seq = m((Hkv, 96), torch.int32), m((B + 2, 80), torch.int32), m((B - 1,), torch.int32)
    assert torch.ops.myops.compute_cent(q, l1).shape == (B, Hq, 64)
    pos, plen, pmax = torch.ops.myops.cent_select(q, l1, ids, table, seq)
    assert pos.shape == (B - 1, Hq, 256) and plen.shape == (B - 1, Hq, 8) and pmax.shape == (B - 1, 8)
    assert pmax.dtype == torch.int64 and pos.dtype == torch.int32
    idx = m((B - 1, Hq, 64), torch.int32)
    a, b, c = torch.ops.myops.select_position(ids, table, seq, idx)
    assert a.shape == (B - 1, Hq, 256) and b.shape == (B - 1, Hq, 8) and c.shape == (B - 1, Hq, 80)
    e, ei = m((0,)), m((0,), torch.int64)
    kv = m((96, 128, Hkv, D))
    out = torch.ops.myops.sparse_paged_attention(q, [kv], [kv], e, e, ei, e, e, e, e, e, e, e, table, ei, pos, Hq, 0.1,
                                                 "BNSD", Hkv, 128, 1)
    assert out.shape == q.shape and out.dtype == q.dtype
    out = torch.ops.myops.sparse_paged_attention(q, [kv], [kv], e, e, ei, e, e, e, m((1,), torch.float32), e, e, e,
                                                 table, ei, pos, Hq, 0.1, "BNSD", Hkv, 128, 1)
    assert out.dtype == torch.int8
    o, p2, m2 = torch.ops.myops.sparse_paged_fusion_attention(q, [kv], [kv], e, e, ei, e, e, e, e, e, e, e, table, ei, l1,
                                                              ids, seq, pos, plen, pmax, Hq, 0.1, "BNSD", Hkv, 128, 1)
    assert o.shape == q.shape and p2.shape == pos.shape and m2.shape == pmax.shape


def _small_fused_case(dev, seed=5):
    B, Hq, Hkv, D, C, bs, pages = 3, 8, 2, 128, 256, 128, 24
    q, l1 = datagen.cent_case(seed, B, Hq, Hkv, D, C)
    block_ids, table, seq = datagen.page_case(seed, Hkv, C, B * pages, B, pages, [pages * bs, 9 * bs + 3, 17 * bs],
                                              aliasing=False)
    k_np, v_np = datagen.kv_case(seed, B * pages, bs, Hkv, D, "BSH")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return dict(B=B, Hq=Hq, Hkv=Hkv, D=D, C=C, bs=bs, np=(q, l1, block_ids, table, seq, k_np, v_np),
                dev=tuple(map(t, (q, l1, block_ids, table, seq, k_np, v_np))))


@pytest.mark.gpu
def test_fused_op_traces_under_torch_compile():
    """torch.compile(fullgraph=True) must get through the fused op without a graph break: it propagates shapes with the
    meta implementation and calls the real one.  (backend "eager": the captured graph is run as it is.)"""
    dev = "cuda:0"
    c = _small_fused_case(dev)
    q, l1, block_ids, table, seq, key, val = c["dev"]
    B, Hq, Hkv, D, bs = c["B"], c["Hq"], c["Hkv"], c["D"], c["bs"]
    e = torch.empty(0, dtype=torch.float16, device=dev)
    ei = torch.empty(0, dtype=torch.int64, device=dev)

    def layer(q, pos, plen, pmax):
        out, pos2, pmax2 = torch.ops.myops.sparse_paged_fusion_attention(
            q, [key], [val], e, e, ei, e, e, e, e, e, e, e, table, ei, l1, block_ids, seq, pos, plen, pmax, Hq,
            D ** -0.5, "BNSD", Hkv, bs, 1)
        return out.float() * 2.0, pmax2

    bufs = lambda: (torch.empty((B, Hq, 256), dtype=torch.int32, device=dev),
                    torch.empty((B, Hq, 8), dtype=torch.int32, device=dev),
                    torch.empty((B, 8), dtype=torch.int64, device=dev))
    want_bufs, got_bufs = bufs(), bufs()
    want, want_max = layer(q, *want_bufs)
    got, got_max = torch.compile(layer, fullgraph=True, backend="eager")(q, *got_bufs)
    torch.cuda.synchronize()
    assert torch.equal(got, want) and torch.equal(got_max, want_max)
    assert all(torch.equal(a, b) for a, b in zip(got_bufs, want_bufs))


@pytest.mark.gpu
def test_dispatch_defaults_reach_the_kernel_variant():
    """torch.ops.myops.* take no keyword extras; set_dispatch_defaults selects k and the device kernel's sink/recent
    pages and pageLen/8 cap (cent_select.cpp:350-362,410-414) for them.  Checked against the oracle."""
    from oracle import ref
    dev = "cuda:0"
    c = _small_fused_case(dev, seed=8)
    qn, l1n, idsn, tablen, seqn, _, _ = c["np"]
    q, l1, block_ids, table, seq, _, _ = c["dev"]
    before = custom_ops.set_dispatch_defaults(k=16, sink_recent=True, max_pages_cap=True)
    try:
        pos, plen, pmax = torch.ops.myops.cent_select(q, l1, block_ids, table, seq)
    finally:
        custom_ops.set_dispatch_defaults(**before)
    wpos, wlen, wmax, _ = ref.cent_select(qn, l1n, idsn, tablen, seqn, k=16, sink_recent=True, max_pages_cap=True)
    assert np.array_equal(pos.cpu().numpy(), wpos) and np.array_equal(plen.cpu().numpy(), wlen)
    assert np.array_equal(pmax.cpu().numpy(), wmax)
    plain = torch.ops.myops.cent_select(q, l1, block_ids, table, seq)        # defaults again: k = 64, oracle behaviour
    w2 = ref.cent_select(qn, l1n, idsn, tablen, seqn, k=64)
    assert np.array_equal(plain[0].cpu().numpy(), w2[0]) and np.array_equal(plain[1].cpu().numpy(), w2[1])
