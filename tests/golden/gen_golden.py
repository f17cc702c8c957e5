#!/usr/bin/env python3
"""Generate golden vectors by running the reference's own CPU path.

Runs ONLY in the build container, where /root/reference exists: it imports the
reference's tests/utils.py (torch only) with importlib, feeds it the seeded
inputs of tests/golden/cases.py and stores the outputs as small .npz files.
The reference itself never travels; the .npz files hold data only.

    python tests/golden/gen_golden.py            # writes tests/golden/*.npz
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import cases  # noqa: E402
import datagen  # noqa: E402

REF_UTILS = "/root/reference/tests/utils.py"


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_utils", REF_UTILS)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def exact_scores(q, l1):
    B, Hq, D = q.shape
    Hkv, C, _ = l1.shape
    G = Hq // Hkv
    s = np.einsum("bngd,ncd->bngc", q.astype(np.float64).reshape(B, Hkv, G, D), l1.astype(np.float64))
    return s.reshape(B, Hq, C)


def main():
    ref = load_reference()
    torch.manual_seed(0)

    for name, spec in cases.CENT_CASES.items():
        q, l1 = cases.cent_inputs(spec)
        idx = ref.torch_compute_cent(t(q), t(l1)).numpy()            # int64 [B,Hq,64]
        sc = exact_scores(q, l1)
        tie_rows = datagen.has_tie_in_top(sc, 64)
        gaps = {k: datagen.boundary_gap(sc, k) for k in (8, 16, 32, 64)}
        if spec.get("grid", True):
            assert all(g > 0 for g in gaps.values()), f"{name}: boundary tie, change the seed {gaps}"
        np.savez_compressed(os.path.join(HERE, name + ".npz"),
                            indices=idx.astype(np.int16), tie_rows=tie_rows,
                            input_sha=np.array(datagen.sha256_of(q, l1)))
        print(name, "indices", idx.shape, "rows with a tie inside top-65:", int(tie_rows.sum()),
              "min boundary gaps", gaps)

    for name, spec in cases.SELECT_CASES.items():
        ids, table, seq, idx = cases.select_inputs(spec)
        pos, length, c_full = ref.torch_select_position(t(ids), t(table), t(seq), t(idx),
                                                        block_size=spec["block_size"])
        np.savez_compressed(os.path.join(HERE, name + ".npz"),
                            page_position=pos.numpy(), page_position_length=length.numpy(),
                            block_table_gather=c_full.numpy(),
                            input_sha=np.array(datagen.sha256_of(ids, table, seq, idx)))
        print(name, "pos", tuple(pos.shape), "mean len", float(length.float().mean()))

    for name, spec in cases.CENT_SELECT_CASES.items():
        q, l1, ids, table, seq = cases.cent_select_inputs(spec)
        B = len(spec["seq_lens"])
        idx64 = ref.torch_compute_cent(t(q), t(l1))                  # [qB,Hq,64]
        sc = exact_scores(q[:B], l1)
        out = {"indices64": idx64.numpy().astype(np.int16),
               "input_sha": np.array(datagen.sha256_of(q, l1, ids, table, seq))}
        for k in spec["ks"]:
            gap = datagen.boundary_gap(sc, k)
            assert gap > 0, f"{name}: k={k} boundary tie, change the seed"
            # the reference oracle's sorted top-64 prefix IS its top-k for k <= 64
            pos, length, _ = ref.torch_select_position(t(ids), t(table), t(seq),
                                                       idx64[:B, :, :k].to(torch.int32),
                                                       block_size=spec["block_size"])
            out[f"page_position_k{k}"] = pos.numpy()
            out[f"page_position_length_k{k}"] = length.numpy()
            # tests/cent_select.py:96-98: max_h(len) * block_size
            out[f"max_page_position_length_k{k}"] = (length.max(dim=1).values.to(torch.int64)
                                                     * spec["block_size"]).numpy()
            print(name, f"k={k}", "pos", tuple(pos.shape), "mean len", float(length.float().mean()),
                  "boundary gap", gap)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


if __name__ == "__main__":
    main()
