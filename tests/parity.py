"""Comparison helpers for the parity tests (integer stages: bit-exact)."""
import numpy as np

SENTINEL = 0x7FFFFFFF


def exact_scores(q, l1):
    q = np.asarray(q)
    l1 = np.asarray(l1)
    B, Hq, D = q.shape
    Hkv, C, _ = l1.shape
    G = Hq // Hkv
    s = np.einsum("bngd,ncd->bngc", q.astype(np.float64).reshape(B, Hkv, G, D),
                  l1.astype(np.float64))
    return s.reshape(B, Hq, C)


def assert_indices_match(got, want, k, tie_rows=None, scores=None, what="indices"):
    """got/want: [B,Hq,>=k] top-k ids sorted descending.

    Rows without an exact tie among their k+1 best scores must match bit for bit.
    Rows with a tie (order among equal scores is unspecified in torch.topk) must pick
    the same score sequence and, unless the tie sits on the k / k+1 boundary, the same set.
    """
    got = np.asarray(got)[..., :k].astype(np.int64)
    want = np.asarray(want)[..., :k].astype(np.int64)
    assert got.shape == want.shape, (got.shape, want.shape)
    if tie_rows is None:
        tie_rows = np.zeros(got.shape[:-1], dtype=bool)
    clean = ~tie_rows
    bad = (got[clean] != want[clean]).any(axis=-1)
    assert not bad.any(), f"{what}: {int(bad.sum())} tie-free rows differ"
    if tie_rows.any():
        assert scores is not None
        sg = np.take_along_axis(scores, got, axis=-1)[tie_rows]
        sw = np.take_along_axis(scores, want, axis=-1)[tie_rows]
        assert np.array_equal(sg, sw), f"{what}: tied rows pick different score sequences"


def assert_positions_match(got, want, what="page_position"):
    """Bit-exact on the common width; whatever lies beyond it must be all sentinel
    (the device pads to a fixed max_page_num, the reference oracle to the batch max,
    tests/utils.py:45-48 vs add_custom.cpp:958)."""
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape[:2] == want.shape[:2], (got.shape, want.shape)
    w = min(got.shape[2], want.shape[2])
    assert np.array_equal(got[:, :, :w], want[:, :, :w]), f"{what}: mismatch in first {w} slots"
    assert (got[:, :, w:] == SENTINEL).all(), f"{what}: got has entries past the common width"
    assert (want[:, :, w:] == SENTINEL).all(), f"{what}: want has entries past the common width"


def rel_err(got, want, floor=None):
    """Element-wise |got-want| / max(|want|, floor); floor defaults to 1e-2 * max|want|."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    if floor is None:
        floor = 1e-2 * max(float(np.abs(want).max()), 1e-30)
    return np.abs(got - want) / np.maximum(np.abs(want), floor)
