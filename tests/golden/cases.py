"""Golden-case specifications shared by gen_golden.py (producer) and the tests (consumers).

Each case names its seed and shapes; inputs are regenerated from the seed with
tests/datagen.py, outputs of the reference's CPU path (tests/utils.py of the
reference) are stored in <name>.npz next to this file together with a sha256 of
the regenerated inputs.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import datagen  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# compute_cent cases: reference k is hard-coded to 64 (tests/utils.py:101); its sorted
# output pins every k <= 64 by prefix.
CENT_CASES = {
    # BASELINE config 1: B=2 N=8 D=64 C=256 (k=8 checked by prefix)
    "cc_c1_mha": dict(seed=11, B=2, Hq=8, Hkv=8, D=64, C=256, grid=True),
    "cc_c1_gqa": dict(seed=12, B=2, Hq=8, Hkv=2, D=64, C=256, grid=True),
    # shapes of the reference's tests/compute_cent.py:16-22
    "cc_ref": dict(seed=13, B=8, Hq=32, Hkv=8, D=128, C=512, grid=True),
    # reference distributions (not tie/rounding safe: compared tolerantly)
    "cc_rand": dict(seed=14, B=4, Hq=32, Hkv=8, D=128, C=512, grid=False),
}

# select_position cases (tests/select_position.py:16-28)
SELECT_CASES = {
    "sp_ref": dict(seed=21, B=8, Hq=32, Hkv=8, C=512, k=64, kv_page_len=1280, max_batch=256,
                   max_page=1024, block_size=128, seq_lens=[32 * 1024] * 8),
    # ragged lengths incl. 1 token, partial pages, one page, D10 (B < max_batch)
    "sp_ragged": dict(seed=22, B=6, Hq=8, Hkv=2, C=64, k=8, kv_page_len=300, max_batch=16,
                      max_page=256, block_size=128,
                      seq_lens=[1, 127, 128, 129, 5000, 32 * 1024]),
    "sp_bs16": dict(seed=23, B=3, Hq=4, Hkv=4, C=32, k=4, kv_page_len=512, max_batch=4,
                    max_page=512, block_size=16, seq_lens=[16 * 200, 777, 4096]),
}

# cent_select = compute_cent -> select_position (tests/cent_select.py:74-98)
CENT_SELECT_CASES = {
    # BASELINE config 2: B=16 H=32 D=128 C=1024 k=16 block_size=128
    "cs_c2": dict(seed=31, B=16, Hq=32, Hkv=8, D=128, C=1024, ks=[16, 64], kv_page_len=1280,
                  max_batch=256, max_page=1024, block_size=128, seq_lens=[32 * 1024] * 16),
    # BASELINE config 3 shape, reduced batch: H=32 (MHA) D=128 S=32k C=2048 k=32
    "cs_c3s": dict(seed=32, B=4, Hq=32, Hkv=32, D=128, C=2048, ks=[32, 64], kv_page_len=1280,
                   max_batch=8, max_page=256, block_size=128, seq_lens=[32 * 1024] * 4),
    # ragged, seq_len shorter than q batch (D10: batch comes from seq_len)
    "cs_ragged": dict(seed=33, B=3, Hq=8, Hkv=2, D=64, C=256, ks=[8, 64], kv_page_len=700,
                      max_batch=8, max_page=600, block_size=128,
                      seq_lens=[70000, 300, 32 * 1024 + 5], q_batch=5),
}


def cent_inputs(spec):
    return datagen.cent_case(spec["seed"], spec["B"], spec["Hq"], spec["Hkv"], spec["D"],
                             spec["C"], spec.get("grid", True))


def select_inputs(spec):
    ids, table, seq = datagen.page_case(spec["seed"], spec["Hkv"], spec["C"], spec["kv_page_len"],
                                        spec["max_batch"], spec["max_page"], spec["seq_lens"])
    idx = datagen.indices_case(spec["seed"], spec["B"], spec["Hq"], spec["k"], spec["C"])
    return ids, table, seq, idx


def cent_select_inputs(spec):
    qb = spec.get("q_batch", spec["B"])
    q, l1 = datagen.cent_case(spec["seed"], qb, spec["Hq"], spec["Hkv"], spec["D"], spec["C"], True)
    ids, table, seq = datagen.page_case(spec["seed"], spec["Hkv"], spec["C"], spec["kv_page_len"],
                                        spec["max_batch"], spec["max_page"], spec["seq_lens"])
    return q, l1, ids, table, seq


def load(name):
    return np.load(os.path.join(HERE, name + ".npz"), allow_pickle=False)
