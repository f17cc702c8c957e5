#!/usr/bin/env python3
"""GPU time of BASELINE configs[0] (compute_cent) and configs[1] (cent_select) on the golden-case inputs."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("cann-ops-adv-vllm_amd", "tests", os.path.join("tests", "golden")):
    sys.path.insert(0, os.path.join(ROOT, p))
import cases  # noqa: E402
import custom_ops  # noqa: E402


def timeit(fn, iters=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
spec = cases.CENT_CASES["cc_c1_mha"]
q, l1 = (dev(x) for x in cases.cent_inputs(spec))
ms = timeit(lambda: custom_ops.compute_cent(q, l1, k=8))
print(f"configs[0] compute_cent B=2 N=8 D=64 C=256 k=8: {ms * 1e3:.1f} us per call (launch included)")
spec = cases.CENT_SELECT_CASES["cs_c2"]
q, l1, ids, table, seq = (dev(x) for x in cases.cent_select_inputs(spec))
ms = timeit(lambda: custom_ops.cent_select(q, l1, ids, table, seq, k=16))
print(f"configs[1] cent_select B=16 H=32/8 D=128 C=1024 k=16 S=32k: {ms * 1e3:.1f} us per call (launch included)")
