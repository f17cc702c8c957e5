#!/usr/bin/env python3
"""Run the standalone cent_select / compute_cent ops on the bench shapes (profiling target for the Q.Cent^T MFMA)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cann-ops-adv-vllm_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import custom_ops  # noqa: E402

a = bench.parse()
dev = torch.device("cuda", 0)
wl = bench.build_workload(a, dev, a.batch, a.seed, custom_ops)
for _ in range(a.steps):
    custom_ops.cent_select(wl["q"], wl["l1"], wl["block_ids"], wl["table"], wl["seq"], k=a.topk)
    custom_ops.compute_cent(wl["q"], wl["l1"], k=a.topk)
torch.cuda.synchronize()
for name, fn in (("cent_select", lambda: custom_ops.cent_select(wl["q"], wl["l1"], wl["block_ids"], wl["table"], wl["seq"], k=a.topk)),
                 ("compute_cent", lambda: custom_ops.compute_cent(wl["q"], wl["l1"], k=a.topk))):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call (launch included, back to back)")
print("done")
