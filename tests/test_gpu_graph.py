"""The exec phase only enqueues (a memset of the workspace counters and one kernel): a prepared launch can be
captured into a HIP graph and replayed, which is how a serving loop would drive it.  The reference's binding
re-plans on every call (extension/csrc/add_custom.cpp:836-895); its executor can be marked repeatable (:847)."""
import numpy as np
import pytest
import torch

import datagen

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("Hq,Hkv,B,persistent", [(8, 8, 3, True), (8, 2, 3, True), (32, 32, 40, True), (8, 2, 3, False),
                                                 (32, 32, 40, False)])
def test_prepared_launch_in_a_hip_graph(Hq, Hkv, B, persistent):
    """persistent: the plan's workspace keeps its zeroed counters between launches (one kernel per launch after the
    first); otherwise every launch, captured ones included, starts with the zeroing kernel."""
    import custom_ops
    import gpu_helpers as H
    D, C, topk, bs, max_page, kv_page_len = 128, 256, 8, 128, 40, 120
    seq_lens = [int(x) for x in datagen.rng_for(1).integers(1, max_page * bs, size=B)]
    q, l1 = datagen.cent_case(11, B, Hq, Hkv, D, C, grid=True)
    ids, table, seq = datagen.page_case(11, Hkv, C, kv_page_len, B, max_page, seq_lens)
    k, v = datagen.kv_case(11, kv_page_len, bs, Hkv, D, "BSH", dist="randn")
    qd, kd, vd, l1d, td, idsd, seqd = (H.dev(x) for x in (q, k, v, l1, table, ids, seq))
    pos = torch.empty((B, Hq, 256), dtype=torch.int32, device=H.DEV)
    plen = torch.empty((B, Hq, 8), dtype=torch.int32, device=H.DEV)
    pmax = torch.empty((B, 8), dtype=torch.int64, device=H.DEV)
    plan = custom_ops.PreparedFusion(qd, kd, vd, td, l1d, idsd, seqd, pos, plen, pmax, Hq, D ** -0.5, Hkv, bs, k=topk,
                                     out_dtype=torch.float32, persistent_workspace=persistent)
    plan.run()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        plan.run()                                   # warm-up on the capture stream
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            plan.run()
    for step in range(3):
        q2, _ = datagen.cent_case(100 + step, B, Hq, Hkv, D, C, grid=True)
        qd.copy_(H.dev(q2))                          # tensors are captured by pointer: update in place
        seqd.copy_(H.dev(np.maximum(1, seq - step * 17).astype(np.int32)))
        plan.out.zero_(); pos.zero_(); plen.zero_(); pmax.zero_()
        graph.replay()
        torch.cuda.synchronize()
        got = [t.clone() for t in (plan.out, pos, plen, pmax)]
        plan.run()                                   # the same launch, eagerly
        torch.cuda.synchronize()
        assert torch.equal(got[1], pos) and torch.equal(got[2], plen) and torch.equal(got[3], pmax)
        scale = float(plan.out.abs().max())
        assert float((got[0] - plan.out).abs().max()) <= 2e-6 * scale
        assert bool(torch.isfinite(got[0]).all()) and float(got[0].abs().max()) > 0
    plan.close()


def test_one_shot_launch_with_host_lengths_can_be_captured():
    """Per-batch actual_seq_lengths from the host travel as kernel arguments, so even a one-shot executor (consumed by
    phase 2, which is all the capture sees) leaves nothing behind that a replay could find freed."""
    import ctypes

    import gpu_helpers as H
    from custom_ops import _abi
    B, Hq, Hkv, D, bs, P, n_blocks = 3, 8, 2, 128, 128, 12, 40
    rng = datagen.rng_for(21)
    q = H.dev(datagen.ref_rand_f16(rng, (B, Hq, D)))
    k_np, v_np = datagen.kv_case(21, n_blocks, bs, Hkv, D, "BSH", dist="randn")
    key, val = H.dev(k_np), H.dev(v_np)
    table = H.dev(rng.integers(0, n_blocks, size=(B, 24)).astype(np.int32))
    pos = H.dev(np.stack([np.stack([np.sort(rng.permutation(24)[:P]) for _ in range(Hkv)]) for _ in range(B)]).astype(np.int32))
    lengths = [3 * bs + 5, P * bs, bs - 1]                       # not uniform: they go through the workspace
    want = H.call_attention(q, key, val, table, pos, lengths, Hq, Hkv, D ** -0.5, bs, out_dtype=torch.float32)
    torch.cuda.synchronize()

    out = torch.zeros((B, Hq, D), dtype=torch.float32, device=H.DEV)
    tens = [_abi.to_acl(t) for t in (q, table, pos, out)]
    kl, vl = _abi.to_acl_list([key]), _abi.to_acl_list([val])
    arr = _abi.to_int_array(lengths)
    ws_size, ex = ctypes.c_uint64(0), _abi.T()
    args = [tens[0], kl, vl, None, None, arr] + [None] * 7 + [tens[1], None, tens[2], Hq, D ** -0.5, b"BNSD", Hkv, bs, 1,
                                                               tens[3]]
    _abi.check(_abi.aclnnSparsePagedAttentionGetWorkspaceSize(*args, ctypes.byref(ws_size), ctypes.byref(ex)), "phase 1")
    for d in tens:                                               # descriptors may die right after phase 1
        _abi.aclDestroyTensor(d)
    _abi.aclDestroyTensorList(kl); _abi.aclDestroyTensorList(vl); _abi.aclDestroyIntArray(arr)
    ws = torch.empty(int(ws_size.value), dtype=torch.uint8, device=H.DEV)
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            _abi.check(_abi.aclnnSparsePagedAttention(ws.data_ptr(), int(ws_size.value), ex, _abi.current_stream(q.device)),
                       "phase 2 under capture")                 # consumes the executor
    for _ in range(2):
        out.zero_()
        ws.fill_(0xAB)                                           # the workspace is scratch between launches
        graph.replay()
        torch.cuda.synchronize()
        assert float((out - want).abs().max()) <= 2e-6 * float(want.abs().max())


@pytest.mark.parametrize("Hq,Hkv,B", [(8, 2, 3), (32, 32, 40)])
def test_first_launch_of_a_plan_may_be_a_captured_one(Hq, Hkv, B):
    """A plan whose very first launch is recorded into a graph (nothing has run on its workspace yet), followed by an
    eager launch BEFORE any replay, then replays: the eager launch must not take the recording for a launch that left
    the workspace clean."""
    import custom_ops
    import gpu_helpers as H
    D, C, topk, bs, max_page, kv_page_len = 128, 256, 8, 128, 40, 120
    seq_lens = [int(x) for x in datagen.rng_for(2).integers(1, max_page * bs, size=B)]
    q, l1 = datagen.cent_case(12, B, Hq, Hkv, D, C, grid=True)
    ids, table, seq = datagen.page_case(12, Hkv, C, kv_page_len, B, max_page, seq_lens)
    k, v = datagen.kv_case(12, kv_page_len, bs, Hkv, D, "BSH", dist="randn")
    qd, kd, vd, l1d, td, idsd, seqd = (H.dev(x) for x in (q, k, v, l1, table, ids, seq))
    outs = []
    for captured_first in (False, True):
        pos = torch.empty((B, Hq, 256), dtype=torch.int32, device=H.DEV)
        plen = torch.empty((B, Hq, 8), dtype=torch.int32, device=H.DEV)
        pmax = torch.empty((B, 8), dtype=torch.int64, device=H.DEV)
        plan = custom_ops.PreparedFusion(qd, kd, vd, td, l1d, idsd, seqd, pos, plen, pmax, Hq, D ** -0.5, Hkv, bs, k=topk,
                                         out_dtype=torch.float32)
        plan._ws.fill_(0xAB)                             # as a workspace from an allocator would be
        if captured_first:
            side = torch.cuda.Stream()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    plan.run()
            plan.run()                                   # eager, before any replay
            torch.cuda.synchronize()
            first = [t.clone() for t in (plan.out, pos, plen, pmax)]
            graph.replay()
            graph.replay()
            torch.cuda.synchronize()
            second = [t.clone() for t in (plan.out, pos, plen, pmax)]
            for a, b in zip(first[1:], second[1:]):
                assert torch.equal(a, b)
            assert float((first[0] - second[0]).abs().max()) <= 2e-6 * float(first[0].abs().max())
            outs.append(first)
        else:
            plan.run()
            torch.cuda.synchronize()
            outs.append([t.clone() for t in (plan.out, pos, plen, pmax)])
        plan.close()
    for a, b in zip(outs[0][1:], outs[1][1:]):
        assert torch.equal(a, b)
    assert float((outs[0][0] - outs[1][0]).abs().max()) <= 2e-6 * float(outs[0][0].abs().max())
