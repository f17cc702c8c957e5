"""GPU parity (through the C-ABI) of compute_cent / select_position / cent_select against the golden
vectors of the reference's CPU path and against the oracle.  Integer stages: bit-exact."""
import numpy as np
import pytest
import torch

import cases
import datagen
import parity
from oracle import ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import custom_ops
    return custom_ops


@pytest.mark.parametrize("name", list(cases.CENT_CASES))
def test_compute_cent_golden(ops, name):
    from gpu_helpers import dev
    spec = cases.CENT_CASES[name]
    q, l1 = cases.cent_inputs(spec)
    g = cases.load(name)
    want = g["indices"].astype(np.int32)
    sc = parity.exact_scores(q, l1)
    got64 = ops.compute_cent(dev(q), dev(l1)).cpu().numpy()
    assert got64.dtype == np.int32 and got64.shape == want.shape
    if spec.get("grid", True):
        for k in (8, 16, 32, 64):
            parity.assert_indices_match(got64, want, k, datagen.has_tie_in_top(sc, k), sc)
        # the kernel's own tie rule: descending score, then ascending id == the oracle exactly
        assert np.array_equal(got64, ref.compute_cent(q, l1, 64))
        for k in (8, 33):
            gk = ops.compute_cent(dev(q), dev(l1), k=k).cpu().numpy()
            assert np.array_equal(gk, got64[:, :, :k])
    else:
        # reference distributions: fp32 summation order decides near-ties; every pick must be
        # within a few ulps of the k-th score and the score sequence must be sorted
        got_s = np.take_along_axis(sc, got64.astype(np.int64), axis=-1)
        assert (np.diff(got_s, axis=-1) <= 1e-5 * np.abs(got_s[..., :-1]) + 1e-6).all()
        kth = np.sort(sc, axis=-1)[..., ::-1][..., 63:64]
        assert (got_s >= kth - 1e-4 * np.maximum(1.0, np.abs(kth))).all()


def test_compute_cent_bf16_and_large_k(ops):
    from gpu_helpers import dev
    rng = datagen.rng_for(77)
    # 8-bit grid values are exact in bf16 too -> exact scores -> bit-exact indices
    q = datagen.grid_f16(rng, (3, 8, 128))
    l1 = datagen.grid_f16(rng, (4, 700, 128))          # C not a multiple of 64
    want = ref.compute_cent(q, l1, 300)
    got = ops.compute_cent(dev(q, torch.bfloat16), dev(l1, torch.bfloat16), k=300).cpu().numpy()
    sc = parity.exact_scores(q, l1)
    parity.assert_indices_match(got, want, 300, datagen.has_tie_in_top(sc, 300), sc)
    assert np.array_equal(got, want)
    got16 = ops.compute_cent(dev(q), dev(l1), k=300).cpu().numpy()
    assert np.array_equal(got16, want)


@pytest.mark.parametrize("name", list(cases.SELECT_CASES))
def test_select_position_golden(ops, name):
    from gpu_helpers import dev
    spec = cases.SELECT_CASES[name]
    ids, table, seq, idx = cases.select_inputs(spec)
    g = cases.load(name)
    pos, length, gather = ops.select_position(dev(ids), dev(table), dev(seq), dev(idx),
                                              page_size=spec["block_size"])
    pos, length, gather = pos.cpu().numpy(), length.cpu().numpy(), gather.cpu().numpy()
    assert pos.shape == (spec["B"], spec["Hq"], 256) and length.shape == (spec["B"], spec["Hq"], 8)
    parity.assert_positions_match(pos, g["page_position"])
    assert (length == length[:, :, :1]).all()
    assert np.array_equal(length[:, :, 0], g["page_position_length"])
    w = g["block_table_gather"].shape[2]
    assert np.array_equal(gather[:, :, :w], g["block_table_gather"])      # tests/select_position.py:64


@pytest.mark.parametrize("name", list(cases.CENT_SELECT_CASES))
def test_cent_select_golden(ops, name):
    from gpu_helpers import dev
    spec = cases.CENT_SELECT_CASES[name]
    q, l1, ids, table, seq = cases.cent_select_inputs(spec)
    g = cases.load(name)
    for k in spec["ks"]:
        want = g[f"page_position_k{k}"]
        P = max(256, want.shape[2])
        pos, length, mx = ops.cent_select(dev(q), dev(l1), dev(ids), dev(table), dev(seq), k=k, max_page_num=P)
        pos, length, mx = pos.cpu().numpy(), length.cpu().numpy(), mx.cpu().numpy()
        assert pos.shape == (len(spec["seq_lens"]), spec["Hq"], P)       # batch from seq_len (D10)
        parity.assert_positions_match(pos, want)
        assert (length == length[:, :, :1]).all()
        assert np.array_equal(length[:, :, 0], g[f"page_position_length_k{k}"])
        assert mx.dtype == np.int64 and (mx == mx[:, :1]).all()
        assert np.array_equal(mx[:, 0], g[f"max_page_position_length_k{k}"])


def test_cent_select_variants_vs_oracle(ops):
    """sink/recent pages, pageLen/8 cap, truncation to P slots: against the oracle."""
    from gpu_helpers import dev
    spec = cases.CENT_SELECT_CASES["cs_ragged"]
    q, l1, ids, table, seq = cases.cent_select_inputs(spec)
    for kw in (dict(sink_recent=True), dict(max_pages_cap=True), dict(sink_recent=True, max_pages_cap=True)):
        for k, P in ((8, 600), (64, 600), (64, 40)):
            wpos, wlen, wmax, _ = ref.cent_select(q, l1, ids, table, seq, k, 128, P, **kw)
            pos, length, mx = ops.cent_select(dev(q), dev(l1), dev(ids), dev(table), dev(seq), k=k,
                                              max_page_num=P, **kw)
            assert np.array_equal(pos.cpu().numpy(), wpos), (kw, k, P)
            assert np.array_equal(length.cpu().numpy(), wlen), (kw, k, P)
            assert np.array_equal(mx.cpu().numpy(), wmax), (kw, k, P)


def test_abi_errors(ops):
    from gpu_helpers import dev
    from custom_ops._abi import AclError
    q = dev(np.zeros((2, 8, 64), np.float16))
    l1 = dev(np.zeros((2, 32, 64), np.float16))
    with pytest.raises(AclError, match="k = 64"):
        ops.compute_cent(q, l1)                      # k > C
    with pytest.raises(AclError, match="dtype"):
        ops.compute_cent(q, l1.to(torch.bfloat16), k=8)
    with pytest.raises(AclError, match="must be 32, 64, 128, 256 or 512"):
        ops.compute_cent(dev(np.zeros((2, 8, 48), np.float16)), dev(np.zeros((2, 32, 48), np.float16)), k=8)
