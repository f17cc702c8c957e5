#!/usr/bin/env python3
"""Time the reference's own CPU path (tests/utils.py of the reference: torch_compute_cent + torch_select_position)
on the golden cases that carry BASELINE configs[0] and configs[1].  Runs ONLY in the build container, where
/root/reference exists; prints one line per case (median of 5)."""
import os
import statistics
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import cases  # noqa: E402
from gen_golden import load_reference, t  # noqa: E402


def med(fn, n=5):
    xs = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        xs.append(time.perf_counter() - t0)
    return statistics.median(xs)


def main():
    ref = load_reference()
    print(f"host threads: torch {torch.get_num_threads()}, cpus {os.cpu_count()}")
    spec = cases.CENT_CASES["cc_c1_mha"]
    q, l1 = cases.cent_inputs(spec)
    s = med(lambda: ref.torch_compute_cent(t(q), t(l1)))
    print(f"configs[0] compute_cent B=2 N=8 D=64 C=256: {s * 1e3:.2f} ms  ({spec['B'] / s:.0f} sequences/s)")
    spec = cases.CENT_SELECT_CASES["cs_c2"]
    q, l1, ids, table, seq = cases.cent_select_inputs(spec)

    def cent_select():
        idx = ref.torch_compute_cent(t(q), t(l1))[:, :, :16].to(torch.int32)
        return ref.torch_select_position(t(ids), t(table), t(seq), idx)
    s = med(cent_select, 3)
    print(f"configs[1] cent_select B=16 H=32 D=128 C=1024 k=16 S=32k: {s * 1e3:.1f} ms  ({spec['B'] / s:.1f} sequences/s)")


if __name__ == "__main__":
    main()
