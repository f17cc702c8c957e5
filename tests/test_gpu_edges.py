"""Edge cases of the chain on the GPU, against the oracle: large cluster counts (the top-k path whose row does not
fit the registers), k = C, exact ties, empty batches, rows that overflow their slots, the sink/recent and cap
variants inside the fused op.  Integer outputs bit-exact, attention within 1e-3."""
import numpy as np
import pytest
import torch

import datagen
import parity
from oracle import ref

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(scope="module")
def H():
    import gpu_helpers
    return gpu_helpers


@pytest.fixture(scope="module")
def ops():
    import custom_ops
    return custom_ops


@pytest.mark.parametrize("C,k,Hq,Hkv,D", [(4096, 64, 4, 2, 128), (16384, 33, 2, 2, 64), (2048, 100, 4, 4, 128),
                                           (2560, 8, 6, 2, 32),
                                           (512, 16, 4, 2, 512), (1024, 32, 4, 4, 512),    # the largest head dim
                                           (640, 640, 8, 8, 512),    # ... with k = C: 1 KiB rows beside the largest sort buffers
                                           (1024, 1000, 4, 2, 256),
                                           (16384, 1024, 2, 1, 128),   # the documented upper limits together: one 64 KiB key row
                                                                       # per pass beside an 8 KiB sort buffer
                                           (100, 7, 6, 2, 32)])   # D = 32 scores 256 centroids per step, C pads to 128
def test_large_cluster_counts(H, ops, C, k, Hq, Hkv, D):
    B, bs, pages = 5, 128, 40
    q, l1 = datagen.cent_case(300 + C, B, Hq, Hkv, D, C, grid=True)
    ids, table, seq = datagen.page_case(300 + C, Hkv, C, B * pages, B, pages,
                                        [pages * bs, 17 * bs + 1, 3, 30 * bs, 9 * bs], aliasing=False)
    ids[:] = ids % 64      # few distinct ids on the pages: a top-k hit selects a visible share of them
    want_idx = ref.compute_cent(q, l1, k)
    sc = parity.exact_scores(q, l1)
    got = ops.compute_cent(H.dev(q), H.dev(l1), k=k).cpu().numpy()
    parity.assert_indices_match(got, want_idx, k, datagen.has_tie_in_top(sc, k), sc)
    assert np.array_equal(got, want_idx)
    wpos, wlen, wmax, _ = ref.cent_select(q, l1, ids, table, seq, k, bs, 256)
    pos, length, mx = ops.cent_select(H.dev(q), H.dev(l1), H.dev(ids), H.dev(table), H.dev(seq), k=k)
    assert np.array_equal(pos.cpu().numpy(), wpos)
    assert np.array_equal(length.cpu().numpy(), wlen)
    assert np.array_equal(mx.cpu().numpy(), wmax)
    kc, vc = datagen.kv_case(7, B * pages, bs, Hkv, D, "BSH", dist="randn")
    want, wpos2, _, _ = ref.sparse_paged_fusion_attention(q, kc, vc, table, l1, ids, seq, Hq, Hkv, D ** -0.5, bs, k, 256)
    out, fpos, flen, fmax = H.call_fusion(H.dev(q), H.dev(kc), H.dev(vc), H.dev(table), H.dev(l1), H.dev(ids),
                                          H.dev(seq), Hq, Hkv, D ** -0.5, bs, k, out_dtype=torch.float32)
    assert np.array_equal(fpos.cpu().numpy(), wpos2)
    assert float(parity.rel_err(H.f32(out), want).max()) <= TOL


def test_k_equals_cluster_count_and_exact_ties(H, ops):
    """All scores of a row equal (q = 0): the k best are the k lowest ids; k = C selects every cluster."""
    B, Hq, Hkv, D, C = 2, 4, 2, 64, 384
    q, l1 = datagen.cent_case(5, B, Hq, Hkv, D, C, grid=True)
    q[0] = 0
    l1[1, 100:200] = l1[1, 100]          # a run of identical centroids: identical scores for every q
    for k in (1, 64, 200, C):
        want = ref.compute_cent(q, l1, k)
        got = ops.compute_cent(H.dev(q), H.dev(l1), k=k).cpu().numpy()
        assert np.array_equal(got, want), k
    assert np.array_equal(ops.compute_cent(H.dev(q), H.dev(l1), k=7).cpu().numpy()[0], np.tile(np.arange(7), (Hq, 1)))
    ids, table, seq = datagen.page_case(5, Hkv, C, 64, B, 30, [30 * 128, 11 * 128 + 5])
    for k in (1, C):
        wpos, wlen, wmax, _ = ref.cent_select(q, l1, ids, table, seq, k, 128, 256)
        pos, length, mx = ops.cent_select(H.dev(q), H.dev(l1), H.dev(ids), H.dev(table), H.dev(seq), k=k)
        assert np.array_equal(pos.cpu().numpy(), wpos) and np.array_equal(length.cpu().numpy(), wlen)
        assert np.array_equal(mx.cpu().numpy(), wmax)
    assert int(wlen[0, 0, 0]) == 30      # k = C: every page of the sequence


def test_empty_sequences_and_empty_batch(H, ops):
    Hq, Hkv, D, C, bs = 4, 2, 64, 128, 128
    q, l1 = datagen.cent_case(9, 3, Hq, Hkv, D, C, grid=True)
    ids, table, seq = datagen.page_case(9, Hkv, C, 32, 3, 8, [0, 5 * bs, 0])
    kc, vc = datagen.kv_case(9, 32, bs, Hkv, D, "BSH", dist="randn")
    want, wpos, wlen, wmax = ref.sparse_paged_fusion_attention(q, kc, vc, table, l1, ids, seq, Hq, Hkv, 0.125, bs, 8, 256)
    out, pos, plen, pmax = H.call_fusion(H.dev(q), H.dev(kc), H.dev(vc), H.dev(table), H.dev(l1), H.dev(ids),
                                         H.dev(seq), Hq, Hkv, 0.125, bs, 8, out_dtype=torch.float32)
    assert np.array_equal(pos.cpu().numpy(), wpos) and np.array_equal(plen.cpu().numpy(), wlen)
    assert np.array_equal(pmax.cpu().numpy(), wmax)
    assert (H.f32(out)[[0, 2]] == 0).all()                         # nothing selected -> zeros
    assert float(parity.rel_err(H.f32(out), want).max()) <= TOL
    # a batch of zero sequences: an empty tensor means "absent operand" on this surface (add_custom.cpp:32-35), so
    # the call is refused exactly as a missing query would be
    from custom_ops._abi import AclError
    e_q = torch.empty((0, Hq, D), dtype=torch.float16, device=H.DEV)
    with pytest.raises(AclError, match="query"):
        ops.compute_cent(e_q, H.dev(l1), k=8)


@pytest.mark.parametrize("Hq,Hkv", [(4, 4), (8, 2)])
def test_fused_variants_and_slot_overflow(H, Hq, Hkv):
    """sink/recent pages, the pageLen/8 cap and rows with fewer slots than selected pages, inside the fused op."""
    D, C, bs, pages, B = 128, 64, 128, 48, 3
    q, l1 = datagen.cent_case(21, B, Hq, Hkv, D, C, grid=True)
    ids, table, seq = datagen.page_case(21, Hkv, C, B * pages, B, pages, [pages * bs, 20 * bs + 9, 2 * bs],
                                        aliasing=False)
    kc, vc = datagen.kv_case(21, B * pages, bs, Hkv, D, "BSH", dist="randn")
    dq, dk, dv, dt, dl, di, ds = (H.dev(x) for x in (q, kc, vc, table, l1, ids, seq))
    for kw in (dict(sink_recent=True), dict(max_pages_cap=True), dict(sink_recent=True, max_pages_cap=True), dict()):
        for k, P in ((16, 256), (32, 12)):
            want, wpos, wlen, wmax = ref.sparse_paged_fusion_attention(q, kc, vc, table, l1, ids, seq, Hq, Hkv,
                                                                       D ** -0.5, bs, k, P, **kw)
            out, pos, plen, pmax = H.call_fusion(dq, dk, dv, dt, dl, di, ds, Hq, Hkv, D ** -0.5, bs, k, P=P,
                                                 out_dtype=torch.float32, **kw)
            assert np.array_equal(pos.cpu().numpy(), wpos), (kw, k, P)
            assert np.array_equal(plen.cpu().numpy(), wlen), (kw, k, P)
            assert np.array_equal(pmax.cpu().numpy(), wmax), (kw, k, P)
            assert float(parity.rel_err(H.f32(out), want).max()) <= TOL, (kw, k, P)
