"""Batch sharding of the sparse paged attention chain across the GPUs of one node.

Every (batch, head) row of every stage is independent (reference: cent_select.cpp:109-121, attention loop
sparse_paged_fusion_attention_split_Bbn2s2_Us2.h:4444-4446); the only cross-row value is the per-batch
max_page_position_length.  So rank r simply owns a contiguous slice of the batch with its q rows,
total_seq_len entries, block_table rows and the KV pages they reference; l1_cent is replicated.  No KV
or activation crosses xGMI.  The one collective is a MAX all-reduce of a single int64 (the global
max_page_position_length), issued on a side stream so it never sits between two decode launches.
The reference itself has no multi-device code.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, end) slice of the batch owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(batch, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_batch(rank: int, world: int, query: torch.Tensor, total_seq_len: torch.Tensor,
                block_table: torch.Tensor):
    """Slices of the per-sequence operands for `rank` (views, no copies).  Page ids inside block_table
    keep addressing the rank's own KV pool."""
    s, e = shard_range(total_seq_len.shape[0], rank, world)
    return query[s:e], total_seq_len[s:e], block_table[s:e]


class GlobalMaxLength:
    """All-reduce(MAX) of max_page_position_length over the ranks, overlapped with the next launch.

    submit(pmax) snapshots the rank-local maximum on a side stream and starts the 8-byte all-reduce
    there (RCCL when the tensors are on a GPU, gloo on the CPU); before_launch() makes the compute
    stream wait only for that snapshot, so the next launch may overwrite pmax while the collective
    is still in flight.
    """

    def __init__(self, device, group: Optional[dist.ProcessGroup] = None):
        self.device = torch.device(device)
        self.group = group
        self.cuda = self.device.type == "cuda"
        self.stream = torch.cuda.Stream(self.device) if self.cuda else None
        self.buf = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.snapshot_done = None
        self.work = None

    def before_launch(self):
        if self.cuda and self.snapshot_done is not None:
            torch.cuda.current_stream(self.device).wait_event(self.snapshot_done)

    def submit(self, max_page_position_length: torch.Tensor):
        if not self.cuda:
            self.buf.copy_(max_page_position_length[:, 0].max().reshape(1))
            dist.all_reduce(self.buf, op=dist.ReduceOp.MAX, group=self.group)
            return
        produced = torch.cuda.current_stream(self.device).record_event()
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(produced)
            if self.work is not None:
                self.work.wait()          # previous all-reduce still owns buf
            self.buf.copy_(max_page_position_length[:, 0].max().reshape(1))
            self.snapshot_done = self.stream.record_event()
            self.work = dist.all_reduce(self.buf, op=dist.ReduceOp.MAX, group=self.group, async_op=True)

    def wait(self) -> torch.Tensor:
        if self.cuda:
            with torch.cuda.stream(self.stream):
                if self.work is not None:
                    self.work.wait()
                    self.work = None
            self.stream.synchronize()
        return self.buf
