"""assign_block_ids: cluster ids for newly filled KV pages (the producer of block_ids; the reference has none, so the
oracle restates this library's own definition: parity unpinned).

Bit-exact cases use pages whose K rows are all equal and lie on the 8-bit grid n/256: the page mean is then that row
exactly, every product with a grid centroid has 16 significant bits and any fp32 summation order gives the same score,
so the argmax (ties: smallest id) is reproducible bit for bit.  Reference-distribution data is checked through the
property that matters: the chosen cluster's score is the row's maximum up to fp32 summation noise."""
import numpy as np
import pytest
import torch

import datagen
from oracle import ref

pytestmark = pytest.mark.gpu


def _bf16_tensor(a_f32, dev):
    return torch.from_numpy(np.ascontiguousarray(a_f32, dtype=np.float32)).to(dev).to(torch.bfloat16)


def _case(seed, B, Hkv, D, C, bs, max_page, layout, dt, const_pages=True):
    rng = datagen.rng_for(seed)
    n_blocks = B * max_page + 3
    table = rng.permutation(n_blocks)[:B * max_page].reshape(B, max_page).astype(np.int32)
    seq = rng.integers(1, max_page * bs + 1, size=B).astype(np.int32)
    seq[0] = max_page * bs                                   # a full table
    if B > 2:
        seq[2] = 3 * bs                                      # ends on a page boundary
    page_len = (seq + bs - 1) // bs
    first = np.array([rng.integers(0, pl + 1) for pl in page_len], dtype=np.int32)
    first[0] = 0
    if B > 1:
        first[1] = page_len[1]                               # nothing new
    if const_pages:
        vec = datagen.grid_f16(rng, (n_blocks, 1, Hkv, D)).astype(np.float32)
        k = np.broadcast_to(vec, (n_blocks, bs, Hkv, D)).copy()
        l1 = datagen.grid_f16(rng, (Hkv, C, D)).astype(np.float32)
    else:
        k = datagen.ref_rand_f16(rng, (n_blocks, bs, Hkv, D)).astype(np.float32)
        l1 = datagen.randn_f16(rng, (Hkv, C, D)).astype(np.float32)
    if dt == "bf16":
        k, l1 = ref.round_to_bf16(k), ref.round_to_bf16(l1)
    if layout == "BNSD":
        k = np.ascontiguousarray(k.transpose(0, 2, 1, 3))
    ids0 = rng.integers(0, C, size=(Hkv, n_blocks)).astype(np.int32)
    return dict(k=k, l1=l1, table=table, seq=seq, first=first, ids0=ids0, bs=bs, layout=layout, dt=dt)


def _run(c, max_new=0, antiq=None):
    import custom_ops
    import gpu_helpers as H
    conv = (lambda a: _bf16_tensor(a, H.DEV)) if c["dt"] == "bf16" else (lambda a: H.dev(a.astype(np.float16)))
    key = H.dev(c["k"]) if c["k"].dtype == np.int8 else conv(c["k"])
    ids = H.dev(c["ids0"].copy())
    kw = {}
    if antiq is not None:
        kw = dict(antiquant_scale=conv(antiq[0]), antiquant_offset=None if antiq[1] is None else conv(antiq[1]))
    out = custom_ops.assign_block_ids(key, H.dev(c["table"]), H.dev(c["seq"]), conv(c["l1"]), ids, H.dev(c["first"]),
                                      block_size=c["bs"], max_new_pages=max_new, **kw)
    torch.cuda.synchronize()
    assert out is ids
    return ids.cpu().numpy()


@pytest.mark.parametrize("B,Hkv,D,C,bs,max_page,layout,dt", [
    (5, 2, 128, 256, 128, 12, "BSH", "f16"),
    (5, 2, 128, 300, 128, 12, "BNSD", "f16"),        # cluster count off the tile size
    (3, 4, 64, 128, 16, 9, "BSH", "f16"),
    (4, 2, 128, 256, 128, 10, "BSH", "bf16"),
    (20, 8, 128, 512, 128, 40, "BSH", "f16"),        # ~400 new pages per head: many groups of 16
    (2, 1, 256, 2048, 64, 6, "BSH", "f16"),
    (3, 2, 32, 64, 32, 5, "BNSD", "f16"),
])
def test_assign_is_bit_exact_on_grid_pages(B, Hkv, D, C, bs, max_page, layout, dt):
    c = _case(17 + B + C, B, Hkv, D, C, bs, max_page, layout, dt)
    got = _run(c)
    want, _ = ref.assign_block_ids(c["k"], c["table"], c["seq"], c["first"], c["l1"], c["ids0"], bs, layout, 0, dt=dt)
    assert np.array_equal(got, want)
    assert (got != c["ids0"]).any()                  # something was assigned; everything else kept its id (equality above)


def test_table_entries_outside_the_pool_are_skipped():
    """An unassigned slot (-1) or an entry beyond the pool inside the new range names no page: no cluster id is
    written for it — in particular not over physical page 0 or the last page, where a clamp would send it."""
    c = _case(77, 4, 2, 128, 256, 128, 10, "BSH", "f16")
    page_len = (c["seq"] + 127) // 128
    c["first"][:] = 0
    n_blocks = c["k"].shape[0]
    c["table"][0, 1] = -1
    c["table"][0, int(page_len[0]) - 1] = n_blocks + 5
    c["table"][2, 0] = -7
    used = set(int(x) for b in range(4) for x in c["table"][b, :page_len[b]] if 0 <= x < n_blocks)
    got = _run(c)
    want, _ = ref.assign_block_ids(c["k"], c["table"], c["seq"], c["first"], c["l1"], c["ids0"], 128, "BSH", 0)
    assert np.array_equal(got, want)
    untouched = [p for p in range(n_blocks) if p not in used]
    assert untouched and np.array_equal(got[:, untouched], c["ids0"][:, untouched])


def test_max_new_pages_bounds_what_one_call_assigns():
    c = _case(91, 6, 2, 128, 256, 128, 16, "BSH", "f16")
    for max_new in (1, 3):
        got = _run(c, max_new=max_new)
        want, _ = ref.assign_block_ids(c["k"], c["table"], c["seq"], c["first"], c["l1"], c["ids0"], 128, "BSH", max_new)
        assert np.array_equal(got, want)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_assign_picks_the_best_cluster_on_reference_distribution_data(dt):
    """Random K rows: the mean carries fp32 summation noise before it is rounded, so ids are compared through scores:
    the chosen cluster's (float64) score is the row maximum up to that noise, and nearly all ids are identical."""
    c = _case(5, 6, 4, 128, 512, 128, 10, "BSH", dt, const_pages=False)
    got = _run(c)
    want, scores = ref.assign_block_ids(c["k"], c["table"], c["seq"], c["first"], c["l1"], c["ids0"], 128, "BSH", 0, dt=dt)
    n = agree = 0
    for (b, page), sc in scores.items():
        phys = c["table"][b, page]
        for n2 in range(sc.shape[0]):
            n += 1
            agree += int(got[n2, phys] == want[n2, phys])
            best = sc[n2].max()
            assert sc[n2, got[n2, phys]] >= best - 2e-2 * abs(best) - 1e-3      # one dtype ulp of the mean, D terms
    assert n > 50 and agree >= 0.97 * n
    untouched = np.ones_like(got, dtype=bool)
    for (b, page) in scores:
        untouched[:, c["table"][b, page]] = False
    assert np.array_equal(got[untouched], c["ids0"][untouched])


def test_assign_with_int8_keys():
    rng = datagen.rng_for(77)
    c = _case(77, 4, 2, 128, 256, 128, 8, "BSH", "f16", const_pages=False)
    c["k"] = rng.integers(-127, 128, size=c["k"].shape).astype(np.int8)
    scale, off = datagen.antiquant_case(77, 2, 128)
    got = _run(c, antiq=(scale.astype(np.float32), off.astype(np.float32)))
    want, scores = ref.assign_block_ids(c["k"], c["table"], c["seq"], c["first"], c["l1"], c["ids0"], 128, "BSH", 0,
                                        antiquant_scale=scale, antiquant_offset=off)
    n = agree = 0
    for (b, page), sc in scores.items():
        phys = c["table"][b, page]
        for n2 in range(sc.shape[0]):
            n += 1
            agree += int(got[n2, phys] == want[n2, phys])
            best = sc[n2].max()
            assert sc[n2, got[n2, phys]] >= best - 2e-2 * abs(best) - 1e-3
    assert agree >= 0.95 * n


def test_assigned_ids_feed_the_fused_op():
    """End to end: ids produced here make cent_select pick exactly the pages whose own best cluster is in the query's
    top-k — the consumer (fused op) and the producer agree on what a cluster id means."""
    import custom_ops
    import gpu_helpers as H
    c = _case(33, 3, 2, 128, 256, 128, 12, "BSH", "f16")
    c["first"][:] = 0
    ids = _run(c)
    Hq, k = 8, 8
    rng = datagen.rng_for(34)
    q = datagen.grid_f16(rng, (3, Hq, 128))
    pos, plen, pmax = custom_ops.cent_select(H.dev(q), H.dev(c["l1"].astype(np.float16)), H.dev(ids), H.dev(c["table"]),
                                             H.dev(c["seq"]), k=k)
    wpos, wlen, wmax, _ = ref.cent_select(q, c["l1"].astype(np.float16), ids, c["table"], c["seq"], k=k)
    assert np.array_equal(pos.cpu().numpy(), wpos) and np.array_equal(plen.cpu().numpy(), wlen)
