"""aclnnSparsePagedFusionAttentionNoAlloc: phase 2 of the fused op on the library's built-in scratch instead of a caller
workspace.  It must give what aclnnSparsePagedFusionAttention gives on the same plan (integer outputs bit for bit,
attention_out to the bar of the other parity tests, bit for bit where the launch is deterministic), on two calls in a
row of which the second is smaller, and allocate nothing."""
import ctypes

import numpy as np
import pytest
import torch

import datagen

pytestmark = pytest.mark.gpu


class Plan:
    """Phase 1 by hand: a repeatable executor of the fused op bound to its own output tensors."""

    def __init__(self, case, k):
        import gpu_helpers as H
        from custom_ops import _abi
        self._abi = _abi
        B, Hq, Hkv, D, bs = case["B"], case["Hq"], case["Hkv"], case["D"], case["bs"]
        self.pos = torch.full((B, Hq, 256), -7, dtype=torch.int32, device=H.DEV)
        self.plen = torch.full((B, Hq, 8), -7, dtype=torch.int32, device=H.DEV)
        self.pmax = torch.full((B, 8), -7, dtype=torch.int64, device=H.DEV)
        self.out = torch.full((B, Hq, D), float("nan"), dtype=torch.float32, device=H.DEV)
        q, key, val, table, l1, ids, seq = case["dev"]
        tens = [_abi.to_acl(t) for t in (q, table, l1, ids, seq, self.pos, self.plen, self.pmax, self.out)]
        kl, vl = _abi.to_acl_list([key]), _abi.to_acl_list([val])
        self.ws_size, self.ex = ctypes.c_uint64(0), _abi.T()
        args = [tens[0], kl, vl] + [None] * 10 + [tens[1], None, tens[2], tens[3], tens[4], Hq, D ** -0.5, b"BNSD", Hkv, bs,
                                                  1, tens[5], tens[6], tens[7], tens[8]]
        _abi.check(_abi.spfaSetTopK(k), "spfaSetTopK")
        _abi.check(_abi.aclnnSparsePagedFusionAttentionGetWorkspaceSize(*args, ctypes.byref(self.ws_size),
                                                                        ctypes.byref(self.ex)), "phase 1")
        for d in tens:
            _abi.aclDestroyTensor(d)
        _abi.aclDestroyTensorList(kl)
        _abi.aclDestroyTensorList(vl)
        _abi.check(_abi.aclSetAclOpExecutorRepeatable(self.ex), "repeatable")
        self.stream = _abi.current_stream(q.device)

    def run_with_workspace(self):
        ws = torch.empty(max(1, int(self.ws_size.value)), dtype=torch.uint8, device=self.out.device)
        self._abi.check(self._abi.aclnnSparsePagedFusionAttention(ws.data_ptr(), int(self.ws_size.value), self.ex,
                                                                  self.stream), "phase 2 (workspace)")
        return ws

    def run_no_alloc(self):
        self._abi.check(self._abi.aclnnSparsePagedFusionAttentionNoAlloc(self.ex, self.stream), "phase 2 (no alloc)")

    def results(self):
        return [t.clone() for t in (self.pos, self.plen, self.pmax, self.out)]

    def close(self):
        self._abi.aclDestroyAclOpExecutor(self.ex)


def make_case(seed, B, Hq, Hkv, max_page, kv_page_len, D=128, C=256, bs=128):
    import gpu_helpers as H
    seq_lens = [int(x) for x in datagen.rng_for(seed).integers(1, max_page * bs, size=B)]
    q, l1 = datagen.cent_case(seed, B, Hq, Hkv, D, C, grid=True)
    ids, table, seq = datagen.page_case(seed, Hkv, C, kv_page_len, B, max_page, seq_lens)
    k, v = datagen.kv_case(seed, kv_page_len, bs, Hkv, D, "BSH", dist="randn")
    return dict(B=B, Hq=Hq, Hkv=Hkv, D=D, bs=bs, dev=tuple(H.dev(x) for x in (q, k, v, table, l1, ids, seq)))


@pytest.mark.parametrize("tail", [2, 0])
def test_no_alloc_entry_point_equals_the_workspace_entry_point(tail):
    """tail = 2: the big launch shares every workgroup's last item (boards and data areas live in the scratch;
    attention_out agrees to fp32 rounding); tail = 0: deterministic launches, everything bit for bit."""
    from custom_ops import _abi
    big = make_case(31, 40, 32, 32, 40, 160)        # 1280 items: more than the resident workgroups, items are claimed
    small = make_case(32, 3, 8, 2, 24, 96)          # 6 items: dealt to several workgroups each (page-list split)
    _abi.check(_abi.spfaSetTailSharing(tail), "spfaSetTailSharing")
    try:
        plans = [(Plan(big, 8), Plan(big, 8)), (Plan(small, 16), Plan(small, 16))]
    finally:
        _abi.spfaSetTailSharing(1)
    keep = [a.run_with_workspace() for a, _ in plans]
    torch.cuda.synchronize()
    # "allocates nothing", seen from the driver: the device's free memory (hipMemGetInfo) before and after the calls,
    # which also counts what a library would allocate behind torch's back; torch's own allocator count beside it
    allocated = torch.cuda.memory_stats()["allocation.all.allocated"]
    free_before, _ = torch.cuda.mem_get_info()
    for _ in range(2):                               # big then small, twice, back to back on one stream
        for _, b in plans:
            b.run_no_alloc()
    torch.cuda.synchronize()
    free_after, _ = torch.cuda.mem_get_info()
    assert torch.cuda.memory_stats()["allocation.all.allocated"] == allocated
    assert free_after >= free_before
    for a, b in plans:
        want, got = a.results(), b.results()
        for w, g in zip(want[:3], got[:3]):
            assert torch.equal(w, g)
        assert bool(torch.isfinite(got[3]).all())
        if tail == 0:
            assert torch.equal(want[3], got[3])
        else:
            assert float((want[3] - got[3]).abs().max()) <= 2e-6 * float(want[3].abs().max())
    # and against the oracle, so that both entry points being wrong together can not pass
    from oracle import cport
    q, key, val, table, l1, ids, seq = (t.cpu().numpy() for t in small["dev"])
    as16 = lambda a: a.view(np.uint16)
    out, pos, len8, max8 = cport.fusion(as16(q), as16(key), as16(val), table, as16(l1), ids, seq, small["Hq"],
                                        small["Hkv"], small["D"] ** -0.5, small["bs"], 16, 256)
    got = plans[1][1].results()
    assert np.array_equal(got[0].cpu().numpy(), pos) and np.array_equal(got[2].cpu().numpy(), max8)
    err = np.abs(got[3].cpu().numpy() - out) / np.maximum(np.abs(out), 1e-2 * np.abs(out).max())
    assert err.max() <= 1e-3                          # relative, elements below 1 % of the largest measured against that
    del keep
    for a, b in plans:
        a.close()
        b.close()


def test_no_alloc_entry_point_rejects_a_plan_beyond_its_scratch():
    """Sharing of last items for 16-head group passes needs one ~300 KB data area per resident workgroup: more than the
    32 MiB the library holds.  Phase 1 plans it (nothing is launched), the entry point refuses it and says why."""
    import gpu_helpers as H
    from custom_ops import _abi
    B, Hq, Hkv, D, C, bs, max_page = 48, 128, 8, 256, 256, 128, 256
    q = torch.zeros((B, Hq, D), dtype=torch.float16, device=H.DEV)
    kv = torch.zeros((8, bs, Hkv, D), dtype=torch.float16, device=H.DEV)
    l1 = torch.zeros((Hkv, C, D), dtype=torch.float16, device=H.DEV)
    ids = torch.zeros((Hkv, 8), dtype=torch.int32, device=H.DEV)
    table = torch.zeros((B, max_page), dtype=torch.int32, device=H.DEV)
    seq = torch.ones((B,), dtype=torch.int32, device=H.DEV)
    case = dict(B=B, Hq=Hq, Hkv=Hkv, D=D, bs=bs, dev=(q, kv, kv, table, l1, ids, seq))
    _abi.check(_abi.spfaSetTailSharing(2), "spfaSetTailSharing")
    try:
        plan = Plan(case, 8)
    finally:
        _abi.spfaSetTailSharing(1)
    assert plan.ws_size.value > 32 << 20
    with pytest.raises(_abi.AclError, match="built-in scratch holds"):
        plan.run_no_alloc()
    plan.close()


def test_captured_no_alloc_launch_replays_after_other_launches():
    """The scratch is shared by every launch on the GPU, so a captured launch must not depend on what the host believed
    about the scratch when it was recorded: warm-up of A, capture of A, an eager launch of a smaller plan B (other
    counters, other data areas), replay of A, eager B again, replay of A — every result equals the workspace entry
    point's.  The calls inside the capture allocate nothing by construction (allocation is illegal there)."""
    from custom_ops import _abi
    big = make_case(41, 40, 32, 32, 40, 160)          # shared last items: boards + data areas in the scratch
    small = make_case(42, 3, 8, 2, 24, 96)            # page-list split: arrival counters + partial states
    mid = make_case(43, 24, 16, 16, 30, 120)          # a plan of its own size in between
    _abi.check(_abi.spfaSetTailSharing(2), "spfaSetTailSharing")
    try:
        ref_a, a = Plan(big, 8), Plan(big, 8)
        ref_b, b = Plan(small, 16), Plan(small, 16)
        ref_c, c = Plan(mid, 8), Plan(mid, 8)
    finally:
        _abi.spfaSetTailSharing(1)
    keep = [pl.run_with_workspace() for pl in (ref_a, ref_b, ref_c)]
    torch.cuda.synchronize()
    want = {id(pl): pl.results() for pl in (ref_a, ref_b, ref_c)}

    def check(pl, ref_pl):
        torch.cuda.synchronize()
        got, exp = pl.results(), want[id(ref_pl)]
        for w, g in zip(exp[:3], got[:3]):
            assert torch.equal(w, g)
        assert bool(torch.isfinite(got[3]).all())
        assert float((exp[3] - got[3]).abs().max()) <= 2e-6 * float(exp[3].abs().max())
        for t in (pl.pos, pl.plen, pl.pmax):
            t.fill_(-7)
        pl.out.fill_(float("nan"))

    side = torch.cuda.Stream()
    graph_a, graph_c = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        for pl in (a, c):
            pl.stream = _abi.current_stream(pl.out.device)
        # C is captured before it has ever run; A after an eager warm-up
        with torch.cuda.graph(graph_c, stream=side):
            c.run_no_alloc()
        a.run_no_alloc()
        side.synchronize()
        with torch.cuda.graph(graph_a, stream=side):
            a.run_no_alloc()
    check(a, ref_a)
    b.run_no_alloc(); check(b, ref_b)
    graph_a.replay(); check(a, ref_a)
    graph_c.replay(); check(c, ref_c)
    b.run_no_alloc(); check(b, ref_b)
    graph_a.replay(); check(a, ref_a)
    graph_a.replay(); graph_c.replay(); check(a, ref_a); check(c, ref_c)
    b.run_no_alloc(); check(b, ref_b)
    del keep
    for pl in (ref_a, a, ref_b, b, ref_c, c):
        pl.close()
