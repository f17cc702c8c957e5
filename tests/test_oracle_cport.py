"""The C restatement (oracle/oracle.c) against the numpy oracle and the golden vectors (CPU only)."""
import numpy as np
import pytest

import cases
import datagen
import parity
from oracle import cport, ref


@pytest.mark.parametrize("name", list(cases.CENT_SELECT_CASES))
def test_c_cent_select_golden(name):
    spec = cases.CENT_SELECT_CASES[name]
    q, l1, ids, table, seq = cases.cent_select_inputs(spec)
    g = cases.load(name)
    for k in spec["ks"]:
        want = g[f"page_position_k{k}"]
        P = max(256, want.shape[2])
        pos, len8, max8, idx = cport.cent_select(q, l1, ids, table, seq, k, spec["block_size"], P)
        parity.assert_positions_match(pos, want)
        assert np.array_equal(len8[:, :, 0], g[f"page_position_length_k{k}"])
        assert np.array_equal(max8[:, 0], g[f"max_page_position_length_k{k}"])
        assert np.array_equal(idx, ref.compute_cent(q[:len(spec["seq_lens"])], l1, k))


def test_c_variants_match_numpy():
    spec = cases.CENT_SELECT_CASES["cs_ragged"]
    q, l1, ids, table, seq = cases.cent_select_inputs(spec)
    for kw in (dict(sink_recent=True), dict(max_pages_cap=True)):
        a = ref.cent_select(q, l1, ids, table, seq, 8, 128, 300, **kw)
        b = cport.cent_select(q, l1, ids, table, seq, 8, 128, 300, **kw)
        for x, y in zip(a[:3], b[:3]):
            assert np.array_equal(x, y)


@pytest.mark.parametrize("kv", ["f16", "i8"])
def test_c_fusion_matches_numpy(kv):
    Hq, Hkv, D, C, k, bs = 8, 2, 128, 256, 8, 128
    seq_lens = [4000, 20 * 128, 300]
    q, l1 = datagen.cent_case(5, 3, Hq, Hkv, D, C, True)
    ids, table, seq = datagen.page_case(5, Hkv, C, 96, 4, 80, seq_lens)
    kc, vc = datagen.kv_case(5, 96, bs, Hkv, D, "BSH", dist="randn", dtype=kv)
    a_s = a_o = None
    if kv == "i8":
        a_s, a_o = datagen.antiquant_case(5, Hkv, D, True)
    want, wpos, wlen, wmax = ref.sparse_paged_fusion_attention(
        q, kc, vc, table, l1, ids, seq, Hq, Hkv, D ** -0.5, bs, k, 256,
        antiquant_scale=None if a_s is None else a_s.astype(np.float32),
        antiquant_offset=None if a_o is None else a_o.astype(np.float32))
    got, pos, len8, max8 = cport.fusion(q, kc, vc, table, l1, ids, seq, Hq, Hkv, D ** -0.5, bs, k, 256,
                                        kv_dt=kv, antiq_scale=a_s, antiq_offset=a_o)
    assert np.array_equal(pos, wpos) and np.array_equal(len8, wlen) and np.array_equal(max8, wmax)
    assert float(parity.rel_err(got, want).max()) <= 1e-4
