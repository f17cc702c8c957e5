"""Phase 1 of the aclnn ABI is synchronous host code: validation and launch planning can be exercised without a GPU.
Descriptors carry fake device addresses (never dereferenced, nothing is launched).  The error cases are the ones the
reference's host checks name (sparse_paged_fusion_attention_tiling_check.cc:28-48,171-222,497-525)."""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int64, c_uint64, c_void_p

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "cann-ops-adv-vllm_amd", "lib", "libcust_opapi.so")
F32, F16, I8, I32, I64, BF16 = 0, 1, 2, 3, 9, 27
PARAM_NULL, PARAM_INVALID = 161001, 161002


@pytest.fixture(scope="module")
def lib():
    L = ctypes.CDLL(LIB)
    L.aclCreateTensor.restype = c_void_p
    L.aclCreateTensor.argtypes = [POINTER(c_int64), c_uint64, c_int, POINTER(c_int64), c_int64, c_int,
                                  POINTER(c_int64), c_uint64, c_void_p]
    L.aclCreateTensorList.restype = c_void_p
    L.aclCreateTensorList.argtypes = [POINTER(c_void_p), c_uint64]
    L.aclCreateIntArray.restype = c_void_p
    L.aclCreateIntArray.argtypes = [POINTER(c_int64), c_uint64]
    L.aclGetRecentErrMsg.restype = c_char_p
    L.aclDestroyAclOpExecutor.argtypes = [c_void_p]
    T = c_void_p
    attrs = [c_int64, c_double, c_char_p, c_int64, c_int64, c_int64]
    L.aclnnSparsePagedAttentionGetWorkspaceSize.argtypes = [T] * 16 + attrs + [T, POINTER(c_uint64), POINTER(T)]
    L.aclnnSparsePagedFusionAttentionGetWorkspaceSize.argtypes = [T] * 18 + attrs + [T] * 4 + [POINTER(c_uint64), POINTER(T)]
    L.aclnnCentSelectGetWorkspaceSize.argtypes = [T] * 9 + [POINTER(c_uint64), POINTER(T)]
    L.aclnnAssignBlockIdsGetWorkspaceSize.argtypes = [T] * 7 + [c_int64, c_int64, T, POINTER(c_uint64), POINTER(T)]
    L.spfaSetTopK.argtypes = [c_int64]
    L.spfaSetActualSeqLengthsDevice.argtypes = [c_void_p, c_int64]
    return L


def tensor(lib, shape, dtype, addr=0x10000):
    n = len(shape)
    dims = (c_int64 * n)(*shape)
    strides = [1] * n
    for i in range(n - 2, -1, -1):
        strides[i] = strides[i + 1] * shape[i + 1]
    return lib.aclCreateTensor(dims, n, dtype, (c_int64 * n)(*strides), 0, 2, dims, n, c_void_p(addr))


def tlist(lib, t):
    return lib.aclCreateTensorList((c_void_p * 1)(t), 1)


def attention(lib, *, Hq=8, Hkv=2, D=128, bs=128, B=3, q_dtype=F16, kv_dtype=F16, kv_shape=None, layout=b"BNSD",
              pos_rows=None, antiq=None, out_dtype=None, pse=None, mask=None, qs2=None, qo2=None):
    q = tensor(lib, (B, Hq, D), q_dtype)
    kv_shape = kv_shape or (40, bs, Hkv, D)
    k, v = tlist(lib, tensor(lib, kv_shape, kv_dtype)), tlist(lib, tensor(lib, kv_shape, kv_dtype))
    table = tensor(lib, (B, 24), I32)
    pos = tensor(lib, (B, pos_rows or Hkv, 12), I32)
    out = tensor(lib, (B, Hq, D), out_dtype if out_dtype is not None else q_dtype)
    ws, ex = c_uint64(0), c_void_p()
    st = lib.aclnnSparsePagedAttentionGetWorkspaceSize(
        q, k, v, pse, mask, None, None, None, None, qs2, qo2, antiq, None, table, None, pos,
        Hq, 0.088, layout, Hkv, bs, 1, out, ctypes.byref(ws), ctypes.byref(ex))
    if st == 0:
        lib.aclDestroyAclOpExecutor(ex)
    return st, ws.value, (lib.aclGetRecentErrMsg() or b"").decode()


def test_attention_plans_without_a_device(lib):
    st, ws, _ = attention(lib)
    assert st == 0 and ws >= 256            # the work counters; no split / sharing areas without a device to size them
    st, ws2, _ = attention(lib, pos_rows=8)  # one list per q head
    assert st == 0
    st, _, _ = attention(lib, kv_shape=(40, 2, 128, 128))          # [blocks, Hkv, bs, D]
    assert st == 0
    st, _, _ = attention(lib, q_dtype=BF16, kv_dtype=BF16, out_dtype=F32)
    assert st == 0


@pytest.mark.parametrize("kw,fragment", [
    (dict(bs=100), "blockSize"),                     # not a multiple of 16
    (dict(bs=1024, kv_shape=(40, 1024, 2, 128)), "blockSize"),      # > 512
    (dict(bs=48, kv_dtype=I8, kv_shape=(40, 48, 2, 128)), "multiple of 32"),
    (dict(Hq=130, Hkv=2), "group size"),             # G = 65 > 64
    (dict(Hq=8, Hkv=3), "not divisible"),
    (dict(kv_dtype=BF16), "KV cache dtype"),         # f16 query with bf16 cache
    (dict(kv_dtype=I8), "int8 KV cache needs"),      # no antiquantScale
    (dict(pos_rows=3), "blockposition"),
    (dict(kv_shape=(40, 128, 2, 64)), "KV cache must be"),
    (dict(out_dtype=I32), "attentionOut dtype"),
    (dict(D=80, kv_shape=(40, 128, 2, 80)), "head dim"),
    (dict(B=65537), "batch size"),
    (dict(layout=b"SBH"), "Only support input_layout"),
    (dict(layout=b"BSND"), "query layout must be BNSD"),          # 4-dim cache
    (dict(D=512, kv_dtype=I8, kv_shape=(40, 128, 2, 512)), "int8 KV cache needs"),   # the shape itself is accepted
    (dict(D=1024, kv_shape=(40, 128, 2, 1024)), "head dim"),
])
def test_attention_rejects(lib, kw, fragment):
    st, _, msg = attention(lib, **kw)
    assert st == PARAM_INVALID, msg
    assert fragment in msg, msg


def test_score_bias_mask_and_output_quantisation_operands(lib):
    """pseShift / attenMask / quantScale2 (…_tiling_check.cc:737-844,412-464, …_tiling.cc:569-650)."""
    S = 24 * 128                                           # sMax = blocktable columns * blockSize
    BOOL, U8 = 12, 4
    assert attention(lib, pse=tensor(lib, (3, 8, 1, S), F16))[0] == 0
    assert attention(lib, pse=tensor(lib, (1, 8, 1, S + 5), F16), mask=tensor(lib, (3, 1, 1, S), BOOL))[0] == 0
    assert attention(lib, mask=tensor(lib, (3, S), U8))[0] == 0
    for kw, fragment in [
        (dict(pse=tensor(lib, (3, 8, 1, 12), F16)), "less than sMax"),
        (dict(pse=tensor(lib, (2, 8, 1, S), F16)), "does not match (B, N, 1, S)"),
        (dict(pse=tensor(lib, (3, 4, 1, S), F16)), "does not match (B, N, 1, S)"),
        (dict(pse=tensor(lib, (3, 8, 1, S), BF16)), "pseShift dtype"),
        (dict(pse=tensor(lib, (3, 8, S), F16)), "pseShift must be"),
        (dict(mask=tensor(lib, (2, S), BOOL)), "attenMask must hold"),
        (dict(mask=tensor(lib, (3, S - 1), BOOL)), "less than sMax"),
        (dict(mask=tensor(lib, (3, S), F16)), "attenMask must be bool"),
        (dict(out_dtype=I8), "quantScale2 is absent"),
        (dict(qs2=tensor(lib, (1,), F32)), "not int8, but quantScale2"),
        (dict(out_dtype=I8, qs2=tensor(lib, (1,), F16)), "float32 or bfloat16"),
        (dict(out_dtype=I8, qs2=tensor(lib, (1,), BF16)), "needs a bfloat16 query"),
        (dict(out_dtype=I8, qs2=tensor(lib, (8, 64), F32)), "quantScale2 must hold"),
        (dict(out_dtype=I8, qs2=tensor(lib, (8, 128), F32), qo2=tensor(lib, (1,), F32)), "quantOffset2 must have"),
        (dict(out_dtype=I8, qs2=tensor(lib, (1,), F32), qo2=tensor(lib, (1,), BF16)), "same dtype"),
    ]:
        st, _, msg = attention(lib, **kw)
        assert st == PARAM_INVALID and fragment in msg, (kw.keys(), msg)
    assert attention(lib, out_dtype=I8, qs2=tensor(lib, (1,), F32))[0] == 0
    assert attention(lib, out_dtype=I8, qs2=tensor(lib, (1, 8, 1, 128), F32), qo2=tensor(lib, (1, 8, 1, 128), F32))[0] == 0
    assert attention(lib, q_dtype=BF16, kv_dtype=BF16, out_dtype=I8, qs2=tensor(lib, (8 * 128,), BF16))[0] == 0
    # antiquant parameters: per channel [2, Hkv, D] (any spelling with that many values) or per tensor [2]
    kv8 = dict(kv_dtype=I8)
    assert attention(lib, antiq=tensor(lib, (2,), F16), **kv8)[0] == 0
    assert attention(lib, antiq=tensor(lib, (2, 2 * 128), F16), **kv8)[0] == 0
    st, _, msg = attention(lib, antiq=tensor(lib, (2, 1), F16), **kv8)
    assert st == PARAM_INVALID and "antiquantScale" in msg


def test_kv_padding_is_ignored_with_a_block_table(lib):
    """…_tiling_check.cc:590-594: left padding is dropped whenever page attention is on."""
    q = tensor(lib, (3, 8, 128), F16)
    k, v = tlist(lib, tensor(lib, (40, 128, 2, 128), F16)), tlist(lib, tensor(lib, (40, 128, 2, 128), F16))
    ws, ex = c_uint64(0), c_void_p()
    st = lib.aclnnSparsePagedAttentionGetWorkspaceSize(
        q, k, v, None, None, None, None, None, None, None, None, None, None, tensor(lib, (3, 24), I32),
        tensor(lib, (1,), I64), tensor(lib, (3, 2, 12), I32), 8, 0.088, b"BNSD", 2, 128, 1, tensor(lib, (3, 8, 128), F16),
        ctypes.byref(ws), ctypes.byref(ex))
    assert st == 0, lib.aclGetRecentErrMsg()
    lib.aclDestroyAclOpExecutor(ex)
    st = lib.aclnnSparsePagedAttentionGetWorkspaceSize(
        q, k, v, None, None, None, tensor(lib, (1,), F32), None, None, None, None, None, None, tensor(lib, (3, 24), I32),
        None, tensor(lib, (3, 2, 12), I32), 8, 0.088, b"BNSD", 2, 128, 1, tensor(lib, (3, 8, 128), F16),
        ctypes.byref(ws), ctypes.byref(ex))
    assert st == PARAM_INVALID and b"dequantScale1" in lib.aclGetRecentErrMsg()


def test_unsupported_operands_and_null_pointers(lib):
    st, _, msg = attention(lib, antiq=tensor(lib, (2, 2, 128), F16))
    assert st == PARAM_INVALID and "antiquant" in msg
    ws, ex = c_uint64(0), c_void_p()
    q = tensor(lib, (3, 8, 128), F16)
    st = lib.aclnnCentSelectGetWorkspaceSize(q, None, None, None, None, None, None, None, None, ctypes.byref(ws),
                                             ctypes.byref(ex))
    assert st == PARAM_NULL and b"l1_cent" in lib.aclGetRecentErrMsg()


def test_fusion_plan_and_mismatches(lib):
    B, Hq, Hkv, D, C, bs = 4, 8, 2, 128, 512, 128
    def call(l1_shape=(Hkv, C, D), ids_rows=Hkv, seq_dtype=I32, pos_shape=(B, Hq, 256)):
        q = tensor(lib, (B, Hq, D), F16)
        k = tlist(lib, tensor(lib, (64, bs, Hkv, D), F16))
        v = tlist(lib, tensor(lib, (64, bs, Hkv, D), F16))
        args = [q, k, v] + [None] * 10 + [tensor(lib, (B, 16), I32), None, tensor(lib, l1_shape, F16),
                                          tensor(lib, (ids_rows, 64), I32), tensor(lib, (B,), seq_dtype)]
        outs = [tensor(lib, pos_shape, I32), tensor(lib, (B, Hq, 8), I32), tensor(lib, (B, 8), I64),
                tensor(lib, (B, Hq, D), F16)]
        ws, ex = c_uint64(0), c_void_p()
        st = lib.aclnnSparsePagedFusionAttentionGetWorkspaceSize(*args, Hq, 0.088, b"BNSD", Hkv, bs, 1, *outs,
                                                                 ctypes.byref(ws), ctypes.byref(ex))
        if st == 0:
            lib.aclDestroyAclOpExecutor(ex)
        return st, (lib.aclGetRecentErrMsg() or b"").decode()
    assert call()[0] == 0
    st, msg = call(l1_shape=(Hkv, 32, D))            # default top-k 64 > 32 clusters
    assert st == PARAM_INVALID and "top-k" in msg
    # the selection knobs are one-shot: a value serves the next phase 1 only, also when that call is rejected
    assert lib.spfaSetTopK(16) == 0
    assert call(l1_shape=(Hkv, 32, D))[0] == 0
    st, msg = call(l1_shape=(Hkv, 32, D))
    assert st == PARAM_INVALID and "top-k" in msg    # back to 64
    assert lib.spfaSetTopK(16) == 0
    assert call(ids_rows=3)[0] == PARAM_INVALID      # consumed by a call that fails for another reason
    st, msg = call(l1_shape=(Hkv, 32, D))
    assert st == PARAM_INVALID and "top-k" in msg
    st, msg = call(ids_rows=3)
    assert st == PARAM_INVALID and "block_ids" in msg
    st, msg = call(seq_dtype=I64)                    # total_seq_len is int32 in the reference (…Us2.h:1664)
    assert st == PARAM_INVALID and "seq_len" in msg
    st, msg = call(pos_shape=(B, Hkv, 256))          # the fused op writes one row per q head
    assert st == PARAM_INVALID and "block_position" in msg
    st, msg = call(l1_shape=(4, C, D))
    assert st == PARAM_INVALID


def test_device_lengths_are_disarmed_by_a_rejected_call(lib):
    """spfaSetActualSeqLengthsDevice is consumed by the next attention phase 1 even if that call fails early."""
    assert lib.spfaSetActualSeqLengthsDevice(c_void_p(0x5000), 1) == 0     # too few values for batch 3 ...
    st, _, msg = attention(lib, bs=100)                                     # ... but the call fails before that check
    assert st == PARAM_INVALID and "blockSize" in msg
    st, _, msg = attention(lib)                                             # the next call reads its own (absent) host array
    assert st == 0, msg


def test_assign_block_ids_plan_and_mismatches(lib):
    B, Hkv, D, C, bs = 4, 2, 128, 512, 128

    def call(kv_shape=(64, bs, Hkv, D), kv_dtype=F16, first_dtype=I32, first_n=B, antiq=None, ids_rows=Hkv, l1_dtype=F16,
             block_size=bs):
        ws, ex = c_uint64(0), c_void_p()
        st = lib.aclnnAssignBlockIdsGetWorkspaceSize(
            tlist(lib, tensor(lib, kv_shape, kv_dtype)), tensor(lib, (B, 16), I32), tensor(lib, (B,), I32),
            tensor(lib, (first_n,), first_dtype), tensor(lib, (Hkv, C, D), l1_dtype), antiq, None, block_size, 0,
            tensor(lib, (ids_rows, 64), I32), ctypes.byref(ws), ctypes.byref(ex))
        if st == 0:
            lib.aclDestroyAclOpExecutor(ex)
        return st, ws.value, (lib.aclGetRecentErrMsg() or b"").decode()

    st, ws, msg = call()
    assert st == 0 and ws == 0, msg
    assert call(kv_shape=(64, Hkv, bs, D))[0] == 0                     # [blocks, Hkv, bs, D]
    assert call(kv_shape=(64, bs, Hkv * D))[0] == 0
    assert call(kv_dtype=I8, antiq=tensor(lib, (2, Hkv, D), F16))[0] == 0
    for kw, fragment in [
        (dict(kv_dtype=I8), "needs antiquantScale"),
        (dict(antiq=tensor(lib, (2, Hkv, D), F16)), "not int8"),
        (dict(kv_dtype=BF16), "key cache dtype"),
        (dict(kv_shape=(64, bs, Hkv, 64)), "key cache must be"),
        (dict(first_dtype=I64), "first_new_page"),
        (dict(first_n=B - 1), "first_new_page has"),
        (dict(ids_rows=3), "block_ids"),
        (dict(block_size=1024), "blockSize"),
        (dict(l1_dtype=F32), "l1_cent must be"),
    ]:
        st, _, msg = call(**kw)
        assert st == PARAM_INVALID and fragment in msg, (kw.keys(), msg)


def test_one_shot_extras_are_validated_and_consumed(lib):
    """The operands that arrive through one-shot settings (softmax LSE output, shared prefix, antiquant mode): checked in
    phase 1, and taken by the very next phase 1 whether it succeeds or fails."""
    lib.spfaSetSoftmaxLseOut.argtypes = [c_void_p, c_int64]
    lib.spfaSetSharedPrefix.argtypes = [c_void_p, c_void_p, c_int64, c_int64]
    lib.spfaSetAntiquantMode.argtypes = [c_int64]
    lib.spfaSetWorkgroupGeometry.argtypes = [c_int]
    B, Hq = 3, 8
    # LSE: too few values -> rejected; the next call (nothing armed) plans fine
    assert lib.spfaSetSoftmaxLseOut(c_void_p(0x4000), B * Hq - 1) == 0
    st, _, msg = attention(lib)
    assert st == PARAM_INVALID and "softmax LSE output" in msg
    assert attention(lib)[0] == 0
    assert lib.spfaSetSoftmaxLseOut(c_void_p(0x4000), B * Hq) == 0
    assert attention(lib)[0] == 0
    # shared prefix: argument checks in the setter, plan checks in phase 1
    assert lib.spfaSetSharedPrefix(c_void_p(0x5000), None, 64, 64) == PARAM_INVALID
    assert lib.spfaSetSharedPrefix(c_void_p(0x5000), c_void_p(0x6000), 64, 65) == PARAM_INVALID
    assert lib.spfaSetSharedPrefix(c_void_p(0x5008), c_void_p(0x6000), 64, 64) == 0
    st, _, msg = attention(lib)
    assert st == PARAM_INVALID and "16-byte aligned" in msg
    assert lib.spfaSetSharedPrefix(c_void_p(0x5000), c_void_p(0x6000), 300, 200) == 0
    st, ws_prefix, _ = attention(lib)
    assert st == 0
    assert attention(lib)[0] == 0                                   # consumed: no prefix any more
    # a pseShift that covers sMax but not prefix + sMax
    assert lib.spfaSetSharedPrefix(c_void_p(0x5000), c_void_p(0x6000), 300, 200) == 0
    st, _, msg = attention(lib, pse=tensor(lib, (1, Hq, 1, 24 * 128), F16))
    assert st == PARAM_INVALID and "prefix + sMax" in msg
    # antiquant mode 1: float32 [2, B, S] parameters and an int8 cache
    assert lib.spfaSetAntiquantMode(4) == PARAM_INVALID      # parameters in page-attention layout: not supported
    assert lib.spfaSetAntiquantMode(-1) == PARAM_INVALID
    assert lib.spfaSetAntiquantMode(1) == 0
    st, _, msg = attention(lib)
    assert st == PARAM_INVALID and "int8 KV cache" in msg
    assert lib.spfaSetAntiquantMode(1) == 0
    st, _, msg = attention(lib, kv_dtype=I8, antiq=tensor(lib, (2, 2, 128), F16))
    assert st == PARAM_INVALID and "float32 [2, B, S]" in msg
    assert lib.spfaSetAntiquantMode(1) == 0
    st, _, msg = attention(lib, kv_dtype=I8, antiq=tensor(lib, (2, B, 24 * 128 - 1), F32))
    assert st == PARAM_INVALID and "less than sMax" in msg
    assert lib.spfaSetAntiquantMode(1) == 0
    assert attention(lib, kv_dtype=I8, antiq=tensor(lib, (2, B, 24 * 128), F32))[0] == 0
    assert attention(lib, kv_dtype=I8, antiq=tensor(lib, (2, 2, 128), F16))[0] == 0     # the mode was one-shot
    # mode 2: [2, Hkv] in the query dtype; mode 3: float32 [2, B, Hkv, S]  (the helper's op has 2 kv heads)
    assert lib.spfaSetAntiquantMode(2) == 0
    st, _, msg = attention(lib, kv_dtype=I8, antiq=tensor(lib, (2, 2, 128), F16))
    assert st == PARAM_INVALID and "[2, Hkv = 2]" in msg
    assert lib.spfaSetAntiquantMode(2) == 0
    assert attention(lib, kv_dtype=I8, antiq=tensor(lib, (2, 2), F16))[0] == 0
    assert lib.spfaSetAntiquantMode(3) == 0
    st, _, msg = attention(lib, kv_dtype=I8, antiq=tensor(lib, (2, B, 24 * 128), F32))
    assert st == PARAM_INVALID and "[2, B, Hkv, S]" in msg
    assert lib.spfaSetAntiquantMode(3) == 0
    st, _, msg = attention(lib, kv_dtype=I8, antiq=tensor(lib, (2, B, 3, 24 * 128), F32))
    assert st == PARAM_INVALID and "holds 3 heads" in msg
    assert lib.spfaSetAntiquantMode(3) == 0
    assert attention(lib, kv_dtype=I8, antiq=tensor(lib, (2, B, 2, 24 * 128), F32))[0] == 0
    # the geometry knob stays until changed and accepts 0, 1, 2 only
    assert lib.spfaSetWorkgroupGeometry(3) == PARAM_INVALID
    for g in (1, 2, 0):
        assert lib.spfaSetWorkgroupGeometry(g) == 0
        assert attention(lib)[0] == 0


def test_token_lists_are_planned_in_windows(lib, monkeypatch):
    """Token-level positions of any length: the plan holds the whole list in LDS up to 4096 entries, a window of 4096
    beyond that, and a smaller window where a 16-head pass leaves less room."""
    lib.spfaSetTokenPositions.argtypes = [c_int]
    lib.spfaDescribeExecutor.argtypes = [c_void_p, c_char_p, c_uint64]

    def window(P, Hq=8, Hkv=2, bs=128, kv_dtype=F16, antiq=None):
        B, D = 2, 128
        q = tensor(lib, (B, Hq, D), F16)
        kv = (40, bs, Hkv, D)
        k, v = tlist(lib, tensor(lib, kv, kv_dtype)), tlist(lib, tensor(lib, kv, kv_dtype))
        table, pos = tensor(lib, (B, 400), I32), tensor(lib, (B, Hkv, P), I32)
        out = tensor(lib, (B, Hq, D), F16)
        ws, ex = c_uint64(0), c_void_p()
        assert lib.spfaSetTokenPositions(1) == 0
        st = lib.aclnnSparsePagedAttentionGetWorkspaceSize(q, k, v, None, None, None, None, None, None, None, None, antiq,
                                                           None, table, None, pos, Hq, 0.088, b"BNSD", Hkv, bs, 1, out,
                                                           ctypes.byref(ws), ctypes.byref(ex))
        assert st == 0, lib.aclGetRecentErrMsg()
        buf = ctypes.create_string_buffer(512)
        assert lib.spfaDescribeExecutor(ex, buf, 512) > 0
        lib.aclDestroyAclOpExecutor(ex)
        plan = dict(part.split("=") for part in buf.value.decode().split())
        return int(plan["token_window"]), int(plan["rows_per_pass"]), int(plan["lds_bytes"])

    assert window(700)[0] == 768                      # whole list, padded to virtual pages of 128 entries
    assert window(4096)[0] == 4096
    assert window(4097)[0] == 4096                    # two windows
    assert window(50000)[0] == 4096
    assert window(50000, bs=48)[0] == 4080            # whole virtual pages
    w, rows, lds = window(50000, Hq=32, Hkv=2)        # 16 q heads per pass: wider state areas, smaller window, same rows
    assert rows == 16 and w < 4096 and w % 128 == 0 and lds <= 160 * 1024
    w, rows, lds = window(20000, Hq=16, Hkv=4, kv_dtype=I8, antiq=tensor(lib, (2, 4, 128), F16))
    assert rows == 4 and w == 4096 and lds <= 160 * 1024
    monkeypatch.setenv("SPFA_TOKEN_WINDOW", "512")
    assert window(50000)[0] == 512 and window(300)[0] == 384
