"""Batched selection of queue-fed launches: when a fused launch with one q head per pass has more items than resident
workgroups, the work queue hands out super-items (up to 8 consecutive sequences of one kv head) whose rows are scored in
one pass over the head's centroids.  Nothing about the results may change: against the oracle, against the same launch
with the batching off, in both workgroup geometries, with a batch that does not divide by the group size, with and
without the sharing of last items."""
import numpy as np
import pytest
import torch

import datagen
import parity
from oracle import cport

pytestmark = pytest.mark.gpu


def _case(seed, B, H, D, C, bs, max_page, kv_page_len, kv="f16"):
    import gpu_helpers as Hh
    rng = datagen.rng_for(seed)
    seq_lens = [int(x) for x in rng.integers(1, max_page * bs + 1, size=B)]
    seq_lens[0] = max_page * bs
    seq_lens[-1] = 1
    q, l1 = datagen.cent_case(seed, B, H, H, D, C, grid=True)
    ids, table, seq = datagen.page_case(seed, H, C, kv_page_len, B, max_page, seq_lens)
    ids[:] = ids % 48                      # few distinct ids: lists of a visible length
    k, v = datagen.kv_case(seed, kv_page_len, bs, H, D, "BSH", dist="randn", dtype="i8" if kv == "int8" else "f16")
    host = dict(q=q, l1=l1, ids=ids, table=table, seq=seq, k=k, v=v)
    if kv == "int8":
        host["a_scale"], host["a_off"] = datagen.antiquant_case(seed, H, D)
    dev = {n: Hh.dev(a) for n, a in host.items()}
    return host, dev


def _plan(dev, B, H, D, bs, topk, **kw):
    import custom_ops
    import gpu_helpers as Hh
    pos = torch.full((B, H, 256), -7, dtype=torch.int32, device=Hh.DEV)
    plen = torch.full((B, H, 8), -7, dtype=torch.int32, device=Hh.DEV)
    pmax = torch.full((B, 8), -7, dtype=torch.int64, device=Hh.DEV)
    plan = custom_ops.PreparedFusion(dev["q"], dev["k"], dev["v"], dev["table"], dev["l1"], dev["ids"], dev["seq"], pos,
                                     plen, pmax, H, D ** -0.5, H, bs, k=topk, out_dtype=torch.float32,
                                     antiquant_scale=dev.get("a_scale"), antiquant_offset=dev.get("a_off"), **kw)
    return plan, pos, plen, pmax


@pytest.mark.parametrize("B,H,D,C,bs,max_page,kv,want_j", [
    (203, 32, 128, 256, 128, 24, "f16", 8),      # three rounds of 8, then eleven single sequences per head
    (100, 32, 64, 256, 64, 40, "f16", 8),        # a round of 8, a round of 4, four singles
    (41, 32, 128, 384, 128, 40, "f16", 4),       # a round of 4, nine singles
    (124, 32, 128, 256, 128, 64, "int8", 8),     # rounds of 8, 4 and 2, then twelve singles
    (260, 32, 128, 256, 16, 70, "f16", 8),       # 16-token pages, 70-column tables
    (19, 16, 128, 256, 128, 30, "f16", 0),       # too few sequences for a round of pairs: the plain queue
])
def test_super_items_against_the_oracle_and_the_plain_queue(B, H, D, C, bs, max_page, kv, want_j):
    from custom_ops import _abi
    topk = 8
    host, dev = _case(7000 + B, B, H, D, C, bs, max_page, 160, kv)
    kw = dict(kv_dt="i8", antiq_scale=host["a_scale"], antiq_offset=host["a_off"]) if kv == "int8" else {}
    want, wpos, wlen, wmax = cport.fusion(host["q"], host["k"], host["v"], host["table"], host["l1"], host["ids"], host["seq"],
                                          H, H, D ** -0.5, bs, topk, 256, **kw)
    results = {}
    for geometry in (0, 1, 2):
        for batched in (1, 0):
            for tail in (1, 0):
                _abi.check(_abi.spfaSetWorkgroupGeometry(geometry), "geometry")
                _abi.check(_abi.spfaSetBatchedSelection(batched), "batched selection")
                _abi.check(_abi.spfaSetTailSharing(tail), "tail sharing")
                try:
                    plan, pos, plen, pmax = _plan(dev, B, H, D, bs, topk)
                finally:
                    _abi.spfaSetWorkgroupGeometry(0)
                    _abi.spfaSetBatchedSelection(1)
                    _abi.spfaSetTailSharing(1)
                d = plan.describe()
                assert d["form"] == "valu"
                if batched and want_j:
                    # (a launch of exactly J items per workgroup takes the static form of the same idea instead)
                    assert d["super_items"] >= 2 or d["scored_ahead"] >= 2, d
                    if geometry == 1 and d["super_items"]:
                        assert d["super_items"] == want_j and d["geometry"] == "wide", d
                else:
                    assert d["super_items"] == 0, d
                for rep in range(2):
                    out = plan.run()
                    torch.cuda.synchronize()
                    assert np.array_equal(pos.cpu().numpy(), wpos), (geometry, batched, tail, rep)
                    assert np.array_equal(plen.cpu().numpy(), wlen) and np.array_equal(pmax.cpu().numpy(), wmax)
                    err = float(parity.rel_err(out.cpu().numpy(), want).max())
                    assert err <= 1e-3, (geometry, batched, tail, rep, err)
                    for t in (pos, plen, pmax):
                        t.fill_(-7)
                results[(geometry, batched, tail)] = out.clone()
                plan.close()
    ref_out = results[(1, 0, 0)]
    for key, out in results.items():
        assert float((out - ref_out).abs().max()) <= 4e-6 * float(ref_out.abs().max()), key
        if key[2] == 0 and key[0] == 1:       # no sharing of last items: the summation order is fixed by the item alone
            assert torch.equal(out, ref_out), key


def test_super_items_with_the_selection_of_the_reference_kernel_and_long_tables():
    """Sink / recent pages and the pageLen / 8 cap inside batched selection, tables of several scan steps (600 columns),
    slot overflow (P = 32)."""
    import custom_ops
    import gpu_helpers as Hh
    B, H, D, C, bs, max_page, topk = 120, 32, 128, 256, 16, 600, 16
    host, dev = _case(7777, B, H, D, C, bs, max_page, 700)
    want, wpos, wlen, wmax = cport.fusion(host["q"], host["k"], host["v"], host["table"], host["l1"], host["ids"], host["seq"],
                                          H, H, D ** -0.5, bs, topk, 32, sink_recent=True, max_pages_cap=True)
    pos = torch.full((B, H, 32), -7, dtype=torch.int32, device=Hh.DEV)
    plen = torch.full((B, H, 8), -7, dtype=torch.int32, device=Hh.DEV)
    pmax = torch.full((B, 8), -7, dtype=torch.int64, device=Hh.DEV)
    plan = custom_ops.PreparedFusion(dev["q"], dev["k"], dev["v"], dev["table"], dev["l1"], dev["ids"], dev["seq"], pos, plen,
                                     pmax, H, D ** -0.5, H, bs, k=topk, out_dtype=torch.float32, sink_recent=True,
                                     max_pages_cap=True)
    assert plan.describe()["super_items"] >= 2
    out = plan.run()
    torch.cuda.synchronize()
    assert np.array_equal(pos.cpu().numpy(), wpos) and np.array_equal(plen.cpu().numpy(), wlen)
    assert np.array_equal(pmax.cpu().numpy(), wmax)
    assert float(parity.rel_err(out.cpu().numpy(), want).max()) <= 1e-3
    plan.close()
