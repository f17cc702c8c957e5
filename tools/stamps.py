#!/usr/bin/env python3
"""Diagnostic: timeline of the item pipeline from in-kernel stamps (needs a -DSPFA_STAMPS library build).  The stamps are
s_memrealtime values: ticks of a constant 100 MHz clock, i.e. 10 ns each."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cann-ops-adv-vllm_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import custom_ops  # noqa: E402
from custom_ops import _abi  # noqa: E402

a = bench.parse()
if a.force_tail_sharing:
    _abi.spfaSetTailSharing(2)
dev = torch.device("cuda", 0)
wl = bench.build_workload(a, dev, a.batch, a.seed, custom_ops)
B, Hq, Hkv, D, bs = a.batch, a.heads, a.kv_heads, a.head_dim, a.block_size
pos = torch.empty((B, Hq, 256), dtype=torch.int32, device=dev)
plen = torch.empty((B, Hq, 8), dtype=torch.int32, device=dev)
pmax = torch.empty((B, 8), dtype=torch.int64, device=dev)
dbg = torch.zeros((1024, 16, 16), dtype=torch.int64, device=dev)
_abi.lib.spfaSetDebugBuffer.argtypes = [ctypes.c_void_p]
_abi.lib.spfaSetDebugBuffer(dbg.data_ptr())
plan = custom_ops.PreparedFusion(wl["q"], wl["key"], wl["val"], wl["table"], wl["l1"], wl["block_ids"], wl["seq"],
                                 pos, plen, pmax, Hq, D ** -0.5, Hkv, bs, k=a.topk, antiquant_scale=wl["a_scale"],
                                 antiquant_offset=wl["a_off"])
import numpy as np
import time


def report(tag, fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    d = dbg.cpu().numpy().astype("int64")
    span = d.max() - d[d > 0].min()
    print(f"--- {tag}: {us:.1f} us, stamp span {span} ticks ({span / us:.0f} ticks/us)")
    return d


e = torch.empty(0, dtype=wl["q"].dtype, device=dev)
ei = torch.empty(0, dtype=torch.int64, device=dev)
plan.run()
torch.cuda.synchronize()
act = pmax[:, 0].contiguous()
if Hkv != Hq:   # the union of a group's lists, one list per kv head (what the fused op streams)
    G = Hq // Hkv
    SENT = 0x7FFFFFFF
    srt = pos.view(B, Hkv, G * 256).to(torch.int64).sort(dim=2).values
    dup = torch.zeros_like(srt, dtype=torch.bool)
    dup[:, :, 1:] = srt[:, :, 1:] == srt[:, :, :-1]
    srt[dup] = SENT
    srt = srt.sort(dim=2).values
    n = int((srt != SENT).sum(dim=2).max().item())
    pos_attn = srt[:, :, :max(n, 1)].to(torch.int32).contiguous()
    act = torch.full((B,), pos_attn.shape[2] * bs, dtype=torch.int64, device=dev)
else:
    pos_attn = pos
try:
    d_attn = report("attention only", lambda: custom_ops.sparse_paged_attention(
        wl["q"], wl["key"], wl["val"], e, e, act, e, e, e, e, e, wl["a_scale"] if wl["a_scale"] is not None else e,
        wl["a_off"] if wl["a_off"] is not None else e, wl["table"], ei, pos_attn, Hq, D ** -0.5, "BNSD", Hkv, bs, 1))
except RuntimeError:
    d_attn = np.zeros((1, 16, 16), dtype="int64")
_ok = d_attn[:, 1, 8] > 0
if _ok.any():
    _idx = np.nonzero(_ok)[0]
    _dur = d_attn[:, :, 9].max(axis=1)[_ok] - d_attn[_ok, 1, 8]
    print("attention only, duration by blockIdx % 8 (mean): " + " ".join(f"{c}:{_dur[_idx % 8 == c].mean():.0f}" for c in range(8)))
for it in range(5):
    row = d_attn[:, it, :]
    ms = (row[:, 9] > 0) & (row[:, 8] > 0)
    if ms.any():
        print(f"item#{it}: n={int(ms.sum())} stream={np.median(row[ms, 9] - row[ms, 8]):.0f} "
              f"first_start={np.median(row[ms, 8]) - d_attn[d_attn > 0].min():.0f}")
_t15 = d_attn[:, 15, :]
_okA = d_attn[:, 1, 8] > 0
if _okA.any():
    _st = d_attn[_okA, 1, 8]
    _wend = d_attn[:, :15, 9].max(axis=1)[_okA]
    _kend = np.where(_t15[_okA, 2] > 0, _t15[_okA, 2], _wend)
    _n = (d_attn[:, 1:15, 8] > 0).sum(axis=1)[_okA]
    print("attention only: items per workgroup " + " ".join(f"{k}:{int((_n == k).sum())}" for k in sorted(set(_n.tolist()))))
    print("attention only: wave 0 done / loop exit / kernel end after first stream start (p0 p50 p100): " +
          " ".join(f"{np.percentile(_wend - _st, q):.0f}" for q in (0, 50, 100)) + " / " +
          (" ".join(f"{np.percentile(_t15[_okA, 0] - _st, q):.0f}" for q in (0, 50, 100)) if (_t15[_okA, 0] > 0).any() else "-") + " / " +
          " ".join(f"{np.percentile(_kend - _st, q):.0f}" for q in (0, 50, 100)))
d = report("fused", plan.run)
row = d[:, 0, :]
ok = row[:, 11] > 0
t_first = row[ok, 11].min()
_last = np.where(d[:, 15, 2] > 0, d[:, 15, 2], d[:, :15, 9].max(axis=1))   # kernel end (tail protocol done) of wave 0
ends = _last[ok] - row[ok, 11]   # s_memtime bases differ between XCDs: only own-start deltas
print(f"prologue: start-spread(p50/max)={np.median(row[ok, 11] - t_first):.0f}/{(row[ok, 11] - t_first).max():.0f} "
      f"score={np.median(row[ok, 12] - row[ok, 11]):.0f} "
      f"rows={np.median((row[ok, 13] - row[ok, 12])[row[ok, 13] > 0]) if (row[ok, 13] > 0).any() else 0:.0f} "
      f"end-start={np.median(row[ok, 14] - row[ok, 11]):.0f}")
print("workgroup durations (ticks from its own start): " +
      " ".join(f"p{q}={np.percentile(ends, q):.0f}" for q in (0, 10, 50, 90, 100)))
idx = np.nonzero(ok)[0]
print("duration by blockIdx % 8 (mean, min, max): " + " | ".join(
    f"{c}: {ends[idx % 8 == c].mean():.0f} {ends[idx % 8 == c].min():.0f} {ends[idx % 8 == c].max():.0f}" for c in range(8)))
half = len(idx) // 2
print(f"first half of the grid mean {ends[idx < half].mean():.0f}, second half {ends[idx >= half].mean():.0f}")
per_item = d[:, 1:10, 9] - d[:, 1:10, 8]
print("stream ticks by item position (median): " + " ".join(f"{np.median(per_item[ok, j][per_item[ok, j] > 0]):.0f}" for j in range(9) if (per_item[ok, j] > 0).any()))
gaps = d[:, 2:10, 8] - d[:, 1:9, 9]
print("wave 0 gap between items (median): " + " ".join(f"{np.median(gaps[ok, j][(d[ok, 2 + j, 8] > 0)]):.0f}" for j in range(8) if (d[ok, 2 + j, 8] > 0).any()))
t15 = d[:, 15, :]
if (t15[ok, 0] > 0).any():
    print(f"tail: loop exit -> own arrival {np.median(t15[ok, 1] - t15[ok, 0]):.0f}, helping {np.median(t15[ok, 2] - t15[ok, 1]):.0f} "
          f"(max {np.max(t15[ok, 2] - t15[ok, 1]):.0f}) ticks; passes p50/max {np.median(t15[ok, 3]):.0f}/{t15[ok, 3].max()}; "
          f"chunks helped total {t15[ok, 4].sum()}, per wg max {t15[ok, 4].max()}")
    print("tail: helping time by phase, median over workgroups (ticks): scan %.0f, copy %.0f, stream %.0f, state + arrival %.0f; "
          "productive visits %.1f" % tuple(np.median(t15[ok, 6 + i]) for i in range(5)))
    print(f"tail: loop exit relative to own start p0/p50/p100: " + " ".join(f"{np.percentile(t15[ok, 0] - row[ok, 11], q):.0f}" for q in (0, 50, 100)))
    print(f"tail: kernel end relative to own start p0/p50/p100: " + " ".join(f"{np.percentile(t15[ok, 2] - row[ok, 11], q):.0f}" for q in (0, 50, 100)))
n_items = (d[:, 1:15, 8] > 0).sum(axis=1)[ok]
print("items per workgroup: " + " ".join(f"{k}:{int((n_items == k).sum())}" for k in sorted(set(n_items.tolist()))))
names = {(0, 1): "fetch", (1, 2): "score", (2, 3): "topk", (3, 4): "select", (4, 5): "union", (5, 6): "qstage"}
for it in range(5):
    row = d[:, it, :]
    ok = row[:, 0] > 0
    if not ok.any():
        continue
    parts = []
    for (i, j), n in names.items():
        m = ok & (row[:, j] > 0) & (row[:, i] > 0)
        if m.any():
            parts.append(f"{n}={np.median(row[m, j] - row[m, i]):.0f}")
    ms = ok & (row[:, 9] > 0) & (row[:, 8] > 0)
    if ms.any():
        parts.append(f"stream={np.median(row[ms, 9] - row[ms, 8]):.0f} start={np.median(row[ms, 8]) - d[d > 0].min():.0f}")
    print(f"item#{it}: n={int(ok.sum())} " + " ".join(parts))
