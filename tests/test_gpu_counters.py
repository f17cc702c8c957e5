"""Library-owned counters: a launch of the two attention ops keeps the part of its workspace that must start zeroed in a
per-(device, stream) slot of the library's own device memory, so a call with ANY caller workspace is one kernel.  What
has to hold: the same results as with the counters in the caller's workspace behind the zeroing kernel
(spfaSetLibraryCounters(0)); a workspace full of garbage is fine; launches that interleave on two streams do not
disturb each other; more streams than slots fall back without anybody noticing."""
import ctypes

import numpy as np
import pytest
import torch

import datagen

pytestmark = pytest.mark.gpu


def _case(seed, B, Hq, Hkv, max_page, kv_page_len, D=128, C=256, bs=128):
    import gpu_helpers as H
    seq_lens = [int(x) for x in datagen.rng_for(seed).integers(1, max_page * bs, size=B)]
    q, l1 = datagen.cent_case(seed, B, Hq, Hkv, D, C, grid=True)
    ids, table, seq = datagen.page_case(seed, Hkv, C, kv_page_len, B, max_page, seq_lens)
    k, v = datagen.kv_case(seed, kv_page_len, bs, Hkv, D, "BSH", dist="randn")
    return dict(B=B, Hq=Hq, Hkv=Hkv, D=D, bs=bs, dev=tuple(H.dev(x) for x in (q, k, v, table, l1, ids, seq)))


class Exec:
    """A repeatable executor of the fused op with outputs of its own; launched with whatever workspace is handed in."""

    def __init__(self, case, k, tail):
        import gpu_helpers as H
        from custom_ops import _abi
        self._abi = _abi
        B, Hq, Hkv, D, bs = case["B"], case["Hq"], case["Hkv"], case["D"], case["bs"]
        self.outs = [torch.full((B, Hq, 256), -7, dtype=torch.int32, device=H.DEV),
                     torch.full((B, Hq, 8), -7, dtype=torch.int32, device=H.DEV),
                     torch.full((B, 8), -7, dtype=torch.int64, device=H.DEV),
                     torch.full((B, Hq, D), float("nan"), dtype=torch.float32, device=H.DEV)]
        q, key, val, table, l1, ids, seq = case["dev"]
        tens = [_abi.to_acl(t) for t in (q, table, l1, ids, seq, *self.outs)]
        kl, vl = _abi.to_acl_list([key]), _abi.to_acl_list([val])
        self.ws_size, self.ex = ctypes.c_uint64(0), _abi.T()
        args = [tens[0], kl, vl] + [None] * 10 + [tens[1], None, tens[2], tens[3], tens[4], Hq, D ** -0.5, b"BNSD", Hkv, bs,
                                                  1, tens[5], tens[6], tens[7], tens[8]]
        _abi.check(_abi.spfaSetTailSharing(tail), "spfaSetTailSharing")
        try:
            _abi.check(_abi.spfaSetTopK(k), "spfaSetTopK")
            _abi.check(_abi.aclnnSparsePagedFusionAttentionGetWorkspaceSize(*args, ctypes.byref(self.ws_size),
                                                                            ctypes.byref(self.ex)), "phase 1")
        finally:
            _abi.spfaSetTailSharing(1)
        for d in tens:
            _abi.aclDestroyTensor(d)
        _abi.aclDestroyTensorList(kl)
        _abi.aclDestroyTensorList(vl)
        _abi.check(_abi.aclSetAclOpExecutorRepeatable(self.ex), "repeatable")
        self.ws = torch.empty(max(1, int(self.ws_size.value)), dtype=torch.uint8, device=H.DEV)

    def launch(self, stream=None):
        st = self._abi.current_stream(self.ws.device) if stream is None else ctypes.c_void_p(stream.cuda_stream)
        self._abi.check(self._abi.aclnnSparsePagedFusionAttention(self.ws.data_ptr(), int(self.ws_size.value), self.ex, st),
                        "phase 2")

    def results(self):
        return [t.clone() for t in self.outs]

    def reset(self):
        for t in self.outs[:3]:
            t.fill_(-7)
        self.outs[3].fill_(float("nan"))

    def close(self):
        self._abi.aclDestroyAclOpExecutor(self.ex)


def _same(want, got):
    for w, g in zip(want[:3], got[:3]):
        assert torch.equal(w, g)
    assert bool(torch.isfinite(got[3]).all())
    assert float((want[3] - got[3]).abs().max()) <= 2e-6 * float(want[3].abs().max())


@pytest.fixture()
def reference_results():
    """(cases, results with the counters in the caller's workspace behind the zeroing kernel)."""
    from custom_ops import _abi
    cases = {"queue": (_case(51, 40, 32, 32, 40, 160), 8, 2),     # items claimed from the queue, shared last items
             "split": (_case(52, 3, 8, 2, 24, 96), 16, 1),        # page-list split: arrival counters, shared scoring
             "group": (_case(53, 24, 16, 4, 30, 120), 8, 1)}      # group passes
    _abi.check(_abi.spfaSetLibraryCounters(0), "spfaSetLibraryCounters")
    want = {}
    try:
        for name, (case, k, tail) in cases.items():
            e = Exec(case, k, tail)
            e.launch()
            torch.cuda.synchronize()
            want[name] = e.results()
            e.close()
    finally:
        _abi.spfaSetLibraryCounters(1)
    return cases, want


def test_garbage_workspace_and_repeated_launches(reference_results):
    cases, want = reference_results
    execs = {name: Exec(case, k, tail) for name, (case, k, tail) in cases.items()}
    for rnd in range(3):
        for name, e in execs.items():           # different plans take turns on one stream: they share its slot
            e.ws.fill_(0xAB if rnd % 2 == 0 else 0xFF)
            e.launch()
        torch.cuda.synchronize()
        for name, e in execs.items():
            _same(want[name], e.results())
            e.reset()
    for e in execs.values():
        e.close()


def test_launches_on_two_streams_interleave(reference_results):
    cases, want = reference_results
    names = ["queue", "group"]
    a = {n: Exec(*cases[n]) for n in names}
    b = {n: Exec(*cases[n]) for n in names}
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for rnd in range(4):
        for n in names:
            a[n].launch(s1)
            b[n].launch(s2)
    torch.cuda.synchronize()
    for n in names:
        _same(want[n], a[n].results())
        _same(want[n], b[n].results())
    for e in list(a.values()) + list(b.values()):
        e.close()


def test_more_streams_than_slots(reference_results):
    cases, want = reference_results
    streams = [torch.cuda.Stream() for _ in range(24)]
    execs = [Exec(*cases["split"]) for _ in streams]
    torch.cuda.synchronize()
    for e, s in zip(execs, streams):
        e.ws.fill_(0xCD)
    torch.cuda.synchronize()
    for rnd in range(2):
        for e, s in zip(execs, streams):
            e.launch(s)
    torch.cuda.synchronize()
    for e in execs:
        _same(want["split"], e.results())
        e.close()


def test_python_surface_is_unchanged_by_the_setting():
    """custom_ops.sparse_paged_fusion_attention / sparse_paged_attention (a new executor and a new workspace per call)
    with and without library-owned counters, against the oracle."""
    import gpu_helpers as H
    from custom_ops import _abi
    from oracle import ref
    Hq, Hkv, D, C, k, bs = 8, 2, 128, 256, 8, 128
    seq_lens = [4000, 32 * 128, 129, 1]
    B = len(seq_lens)
    q, l1 = datagen.cent_case(3, B, Hq, Hkv, D, C, True)
    ids, table, seq = datagen.page_case(3, Hkv, C, 96, B, 80, seq_lens)
    kc, vc = datagen.kv_case(3, 96, bs, Hkv, D, "BSH", dist="randn")
    want, wpos, wlen, wmax = ref.sparse_paged_fusion_attention(q, kc, vc, table, l1, ids, seq, Hq, Hkv, D ** -0.5, bs, k, 256)
    dv = [H.dev(x) for x in (q, kc, vc, table, l1, ids, seq)]
    for setting in (1, 0, 1):
        _abi.check(_abi.spfaSetLibraryCounters(setting), "spfaSetLibraryCounters")
        try:
            for _ in range(2):
                out, pos, plen, pmax = H.call_fusion(*dv, Hq, Hkv, D ** -0.5, bs, k, out_dtype=torch.float32)
                torch.cuda.synchronize()
                assert np.array_equal(pos.cpu().numpy(), wpos) and np.array_equal(plen.cpu().numpy(), wlen)
                assert np.array_equal(pmax.cpu().numpy(), wmax)
                got = out.cpu().numpy()
                err = np.abs(got - want) / np.maximum(np.abs(want), 1e-2 * np.abs(want).max())
                assert err.max() <= 1e-3
                sep = H.call_attention(dv[0], dv[1], dv[2], dv[3], pos, pmax[:, 0].contiguous(), Hq, Hkv, D ** -0.5, bs,
                                       out_dtype=torch.float32)
                torch.cuda.synchronize()
                assert float((sep - out).abs().max()) <= 1e-3 * float(out.abs().max())   # other kernel form (one q head per pass)
        finally:
            _abi.spfaSetLibraryCounters(1)


def test_cent_select_is_one_kernel_with_library_counters():
    """cent_select accumulates the per-sequence maximum over rows that different workgroups select: with the stream's
    slot of library-owned counters the launch is one kernel and leaves the slot clean; without (setting off, or a launch
    that is being captured) a kernel zeroes the output first.  Same results either way, against the oracle."""
    import custom_ops
    import gpu_helpers as H
    from custom_ops import _abi
    from oracle import ref
    B, Hq, Hkv, D, C, k, bs, max_page = 37, 16, 4, 128, 512, 16, 128, 300
    seq_lens = [int(x) for x in datagen.rng_for(5).integers(1, max_page * bs, size=B)]
    q, l1 = datagen.cent_case(61, B, Hq, Hkv, D, C, grid=True)
    ids, table, seq = datagen.page_case(61, Hkv, C, 400, B, max_page, seq_lens)
    wpos, wlen, wmax, _ = ref.cent_select(q, l1, ids, table, seq, k, bs, 256)
    dv = [H.dev(x) for x in (q, l1, ids, table, seq)]

    def run(stream=None):
        with torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream()):
            n0 = int(_abi.spfaKernelLaunches())
            pos, plen, pmax = custom_ops.cent_select(*dv, k=k)
            n = int(_abi.spfaKernelLaunches()) - n0
        torch.cuda.synchronize()
        assert np.array_equal(pos.cpu().numpy(), wpos) and np.array_equal(plen.cpu().numpy(), wlen)
        assert np.array_equal(pmax.cpu().numpy(), wmax)
        return n

    assert run() <= 2                      # (the first use of a stream's slot zeroes it)
    assert [run() for _ in range(3)] == [1, 1, 1]
    side = torch.cuda.Stream()
    assert run(side) <= 2 and run(side) == 1 and run() == 1
    _abi.check(_abi.spfaSetLibraryCounters(0), "spfaSetLibraryCounters")
    try:
        assert run() == 2
    finally:
        _abi.spfaSetLibraryCounters(1)
    assert run() == 1
    # a sequence count beyond what a slot holds falls back by itself
    Bbig = 6000
    qb = torch.zeros((Bbig, 4, 64), dtype=torch.float16, device=H.DEV)
    l1b = torch.zeros((4, 128, 64), dtype=torch.float16, device=H.DEV)
    idsb = torch.zeros((4, 8), dtype=torch.int32, device=H.DEV)
    tableb = torch.zeros((Bbig, 2), dtype=torch.int32, device=H.DEV)
    seqb = torch.full((Bbig,), 200, dtype=torch.int32, device=H.DEV)
    n0 = int(_abi.spfaKernelLaunches())
    pos, plen, pmax = custom_ops.cent_select(qb, l1b, idsb, tableb, seqb, k=8)
    torch.cuda.synchronize()
    assert int(_abi.spfaKernelLaunches()) - n0 == 2
    # all scores tie: the 8 lowest cluster ids are taken, every page has cluster id 0 -> both pages selected
    assert bool((plen == 2).all()) and bool((pmax == 2 * 128).all())
