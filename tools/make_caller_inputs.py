#!/usr/bin/env python3
"""Inputs for tests/cpp/ref_caller (the compiled caller with the reference binding's call sequence) outside pytest, so
that the program can be profiled: tools/make_caller_inputs.py <dir>; then tests/cpp/ref_caller <dir>."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402

out = sys.argv[1]
os.makedirs(out, exist_ok=True)
B, Hq, Hkv, D, C, k, bs, max_page, kv_page_len = 5, 8, 2, 128, 256, 8, 128, 24, 96
seq_lens = [24 * 128, 3000, 129, 1, 2000]
q, l1 = datagen.cent_case(41, B, Hq, Hkv, D, C, True)
ids, table, seq = datagen.page_case(41, Hkv, C, kv_page_len, B, max_page, seq_lens)
kc, vc = datagen.kv_case(41, kv_page_len, bs, Hkv, D, "BSH", dist="randn")
for name, a in (("q", q), ("l1", l1), ("key", kc), ("val", vc), ("table", table), ("ids", ids), ("seq", seq)):
    np.ascontiguousarray(a).tofile(os.path.join(out, f"{name}.bin"))
meta = dict(B=B, Hq=Hq, Hkv=Hkv, D=D, C=C, k=k, bs=bs, max_page=max_page, n_blocks=kv_page_len, kv_page_len=kv_page_len, P=256)
with open(os.path.join(out, "meta.txt"), "w") as f:
    f.write("".join(f"{key} {val}\n" for key, val in meta.items()))
