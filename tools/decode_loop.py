#!/usr/bin/env python3
"""A stack of fused decode launches (one per layer, each with its own query / outputs / workspace, all reading the same
KV pool) driven eagerly and from one HIP graph: what a serving loop pays per layer on top of the kernel time.
With --assign the sequences also grow by one token per step and, whenever a page fills up (every block_size steps), every
layer assigns the new page its cluster id (custom_ops.assign_block_ids) before the step's launches: decode tokens/s with
and without that maintenance step.
Usage: tools/decode_loop.py [bench.py shape flags] --layers 8 [--assign]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cann-ops-adv-vllm_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import custom_ops  # noqa: E402

layers = 8
if "--layers" in sys.argv:
    i = sys.argv.index("--layers")
    layers = int(sys.argv[i + 1])
    del sys.argv[i:i + 2]
with_assign = "--assign" in sys.argv
if with_assign:
    sys.argv.remove("--assign")
a = bench.parse()
dev = torch.device("cuda", 0)
wl = bench.build_workload(a, dev, a.batch, a.seed, custom_ops)
B, Hq, Hkv, D, bs = a.batch, a.heads, a.kv_heads, a.head_dim, a.block_size
plans = []
for layer in range(layers):
    q = wl["q"].clone()
    pos = torch.empty((B, Hq, 256), dtype=torch.int32, device=dev)
    plen = torch.empty((B, Hq, 8), dtype=torch.int32, device=dev)
    pmax = torch.empty((B, 8), dtype=torch.int64, device=dev)
    plans.append(custom_ops.PreparedFusion(q, wl["key"], wl["val"], wl["table"], wl["l1"], wl["block_ids"], wl["seq"],
                                           pos, plen, pmax, Hq, D ** -0.5, Hkv, bs, k=a.topk,
                                           antiquant_scale=wl["a_scale"], antiquant_offset=wl["a_off"]))


def step():
    for p in plans:
        p.run()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


eager = timed(step)
side = torch.cuda.Stream()
graph = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    step()
    side.synchronize()
    with torch.cuda.graph(graph, stream=side):
        step()
replay = timed(graph.replay)
print(f"{layers} layers, batch {B}: eager {eager:.3f} ms/step ({eager / layers:.4f} per layer, "
      f"{B / eager * 1e3:.0f} tokens/s), graph {replay:.3f} ms/step ({replay / layers:.4f} per layer, "
      f"{B / replay * 1e3:.0f} tokens/s)")

if with_assign:
    seq = wl["seq"]                       # the plans read it by pointer: grow the sequences in place
    start = a.seqlen - 2 * bs
    first_new = torch.empty_like(seq)
    n_steps = 2 * bs

    def grow(maintain):
        seq.fill_(start)
        for s in range(n_steps):
            seq.add_(1)
            if maintain and (start + s + 1) % bs == 0:
                first_new.copy_(torch.div(seq, bs, rounding_mode="floor") - 1)      # the page that just filled up
                for _ in range(layers):
                    custom_ops.assign_block_ids(wl["key"], wl["table"], seq, wl["l1"], wl["block_ids"], first_new,
                                                block_size=bs, max_new_pages=1, antiquant_scale=wl["a_scale"],
                                                antiquant_offset=wl["a_off"])
            step()

    def timed_loop(maintain):
        grow(maintain)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        grow(maintain)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    plain, maintained = timed_loop(False), timed_loop(True)
    one = timed(lambda: custom_ops.assign_block_ids(wl["key"], wl["table"], seq, wl["l1"], wl["block_ids"], first_new,
                                                    block_size=bs, max_new_pages=1, antiquant_scale=wl["a_scale"],
                                                    antiquant_offset=wl["a_off"]))
    print(f"growing sequences, {n_steps} steps x {layers} layers: {B * n_steps / plain * 1e3:.0f} tokens/s without, "
          f"{B * n_steps / maintained * 1e3:.0f} tokens/s with a page -> cluster assignment per layer every {bs} steps "
          f"({B} new pages x {Hkv} kv heads per call, {one * 1e3:.1f} us per call)")
