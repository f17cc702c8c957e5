"""Batch sharding over ranks (world_size 2, gloo, CPU): shards reproduce the unsharded page selection and
the MAX all-reduce yields the global max_page_position_length.  The per-rank compute here is the CPU
oracle (no GPU in this test); the collective and the shard arithmetic are the code under test."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
from oracle import ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_shard():
    # import custom_ops.shard without importing the package __init__ (which needs the HIP library + GPU ops)
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "spfa_shard", os.path.join(ROOT, "cann-ops-adv-vllm_amd", "custom_ops", "shard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = _load_shard()
    spec = cases.CENT_SELECT_CASES["cs_ragged"]
    q, l1, ids, table, seq = cases.cent_select_inputs(spec)
    B = len(spec["seq_lens"])
    qs, ss, ts = shard.shard_batch(rank, world, torch.from_numpy(q[:B]), torch.from_numpy(seq),
                                   torch.from_numpy(table[:B]))
    pos, len8, max8, _ = ref.cent_select(qs.numpy(), l1, ids, ts.numpy(), ss.numpy(), 8, 128, 600)
    red = shard.GlobalMaxLength("cpu")
    red.before_launch()
    if ss.numel() > 0:
        red.submit(torch.from_numpy(max8))
    else:
        red.submit(torch.zeros((1, 8), dtype=torch.int64))
    gmax = int(red.wait().item())
    # every = 2: only every second call reaches the collective (and with it every rank, or none); every = 0: none does
    sparse, never = shard.GlobalMaxLength("cpu", every=2), shard.GlobalMaxLength("cpu", every=0)
    for step in range(5):
        local = torch.full((1, 8), 100 * (step + 1) + rank, dtype=torch.int64)
        sparse.submit(local)
        never.submit(local)
    assert sparse.submitted == 2 and int(sparse.wait().item()) == 100 * 4 + world - 1
    assert never.submitted == 0 and int(never.wait().item()) == 0 and never.collective_ms() is None
    ret[rank] = (shard.shard_range(B, rank, world), pos, int(gmax))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_shard_matches_unsharded():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    spec = cases.CENT_SELECT_CASES["cs_ragged"]
    q, l1, ids, table, seq = cases.cent_select_inputs(spec)
    full_pos, full_len, full_max, _ = ref.cent_select(q, l1, ids, table, seq, 8, 128, 600)
    covered = []
    for r in range(world):
        (s, e), pos, gmax = ret[r]
        covered += list(range(s, e))
        assert np.array_equal(pos, full_pos[s:e])
        assert gmax == int(full_max[:, 0].max())
    assert covered == list(range(len(spec["seq_lens"])))


def test_shard_range_partitions():
    shard = _load_shard()
    for B in (0, 1, 7, 64, 513):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_range(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1
