"""Batch sharding of the sparse paged attention chain across the GPUs of one node.

Every (batch, head) row of every stage is independent (reference: cent_select.cpp:109-121, attention loop
sparse_paged_fusion_attention_split_Bbn2s2_Us2.h:4444-4446); the only cross-row value is the per-batch
max_page_position_length.  So rank r simply owns a contiguous slice of the batch with its q rows,
total_seq_len entries, block_table rows and the KV pages they reference; l1_cent is replicated.  No KV
or activation crosses xGMI.  The one collective is a MAX all-reduce of a single int64 (the global
max_page_position_length), issued on side streams so it never sits between two decode launches.
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
    """All-reduce(MAX) of max_page_position_length over the ranks, off the path of the decode launches.

    With RCCL: submit(pmax) takes a snapshot of the rank-local maximum on a side stream (one small kernel) and runs the
    8-byte all-reduce on a second side stream; nothing on the host ever waits.  The only thing the compute stream has to
    respect is that a launch must not overwrite a pmax tensor whose snapshot is still to be taken:
    before_launch(pmax) makes it wait for the last snapshot that read THAT tensor.  A decode loop that alternates two
    sets of output tensors therefore never waits for the snapshot of the step just launched — that wait, on a single
    set of outputs, is a bubble of two dependent tiny kernels between every two launches (measured: 35 us per step).
    Snapshots rotate through `depth` buffers: a snapshot waits for the collective that last used its buffer, the one
    `depth` submissions back.  `every` = N submits on every N-th call only (0: never): a serving loop that needs the
    global maximum once per several steps, or an A/B of what the collective costs.

    With a backend that moves GPU tensors through the host (gloo: rehearsals of the ranks on one device) the value is
    fetched to the host (which waits for the launch) and reduced there: correct, not overlapped.

    collective_ms() is the mean duration of the timed all-reduces (events on the collective's stream; RCCL only).
    """

    def __init__(self, device, group: Optional[dist.ProcessGroup] = None, every: int = 1, depth: int = 4):
        self.device = torch.device(device)
        self.group = group
        self.every = int(every)
        self.depth = max(1, int(depth))
        self.cuda = self.device.type == "cuda"
        self.calls = 0
        self.submitted = 0
        self.bufs = [torch.zeros(1, dtype=torch.int64, device=self.device) for _ in range(self.depth)]
        self.buf = self.bufs[0]                      # the buffer of the most recent submission
        self.snapshots = {}                          # data_ptr of a pmax tensor -> event of the last snapshot that read it
        self.timed = []                              # (start, end) events of the all-reduces
        self.stream_ordered = self.cuda and dist.get_backend(group) == "nccl"
        if self.stream_ordered:
            self.snap_stream = torch.cuda.Stream(self.device)
            self.coll_stream = torch.cuda.Stream(self.device)
            self.coll_done = [None] * self.depth

    def before_launch(self, max_page_position_length: Optional[torch.Tensor] = None):
        """Call before a launch that writes `max_page_position_length` (None: any tensor ever submitted)."""
        if not self.stream_ordered:
            return
        cur = torch.cuda.current_stream(self.device)
        if max_page_position_length is None:
            for ev in self.snapshots.values():
                cur.wait_event(ev)
        else:
            ev = self.snapshots.get(max_page_position_length.data_ptr())
            if ev is not None:
                cur.wait_event(ev)

    def submit(self, max_page_position_length: torch.Tensor):
        self.calls += 1
        if self.every <= 0 or self.calls % self.every != 0:
            return
        slot = self.submitted % self.depth
        self.submitted += 1
        buf = self.bufs[slot]
        self.buf = buf
        if not self.stream_ordered:
            local = max_page_position_length[:, 0].max().reshape(1)
            if self.cuda:                            # through the host: waits for the launch that produced the value
                host = local.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.MAX, group=self.group)
                buf.copy_(host)
            else:
                buf.copy_(local)
                dist.all_reduce(buf, op=dist.ReduceOp.MAX, group=self.group)
            return
        produced = torch.cuda.current_stream(self.device).record_event()
        with torch.cuda.stream(self.snap_stream):
            self.snap_stream.wait_event(produced)
            if self.coll_done[slot] is not None:     # the collective that last used this buffer
                self.snap_stream.wait_event(self.coll_done[slot])
            torch.amax(max_page_position_length[:, 0], dim=0, keepdim=True, out=buf)   # one kernel, straight into buf
            snapshot_done = self.snap_stream.record_event()
            self.snapshots[max_page_position_length.data_ptr()] = snapshot_done
        with torch.cuda.stream(self.coll_stream):
            # RCCL: the call only enqueues (the collective runs in stream order behind the snapshot), so it can be
            # bracketed with events and nothing on the host ever waits for it
            self.coll_stream.wait_event(snapshot_done)
            timed = len(self.timed) < 256
            if timed:
                start = torch.cuda.Event(enable_timing=True)
                start.record(self.coll_stream)
            dist.all_reduce(buf, op=dist.ReduceOp.MAX, group=self.group)
            done = torch.cuda.Event(enable_timing=timed)
            done.record(self.coll_stream)
            self.coll_done[slot] = done
            if timed:
                self.timed.append((start, done))

    def wait(self) -> torch.Tensor:
        """Block until every submitted collective is complete; returns the most recent global maximum."""
        if self.stream_ordered:
            self.snap_stream.synchronize()
            self.coll_stream.synchronize()
        return self.buf

    def collective_ms(self) -> Optional[float]:
        """Mean duration of the timed all-reduces in ms (None before any has run).  Call after wait()."""
        if not self.timed:
            return None
        return sum(a.elapsed_time(b) for a, b in self.timed) / len(self.timed)
